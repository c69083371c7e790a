set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
for scene in sponza sanmiguel; do
  echo "== $scene"
  SWEEP_SCENE=$scene timeout -k 10 600 python tools/sweep_env.py "" "ADYPT_REFILL_MIN=12" "ADYPT_REFILL_MIN=14" "" "ADYPT_REFILL_MIN=12" "ADYPT_REFILL_MIN=14" "" "ADYPT_REFILL_MIN=12" "ADYPT_REFILL_MIN=14" 2>/dev/null
done
