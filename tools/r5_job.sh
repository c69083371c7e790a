set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
for scene in sponza sanmiguel; do
  echo "== $scene"
  SWEEP_SCENE=$scene timeout -k 10 300 python tools/sweep_env.py "" 2>/dev/null
  for v in w7s288 w7s304; do
    ADYPT_PATH_VERBOSE=1 ADYPT_LIB=$PWD/adypt_amd/libadypt_$v.so SWEEP_SCENE=$scene timeout -k 10 300 python tools/sweep_env.py "ADYPT_PATH_BLOCKS_PER_CU=7" 2>&1 | grep -E "^\{|k_path:" | sed "s/^/$v /"
  done
  SWEEP_SCENE=$scene timeout -k 10 300 python tools/sweep_env.py "" 2>/dev/null
done
