set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5/gputest.log; tail -3 gpurun_out/r5/gputest.log
ADYPT_LIB=$PWD/adypt_amd/libadypt_cap64.so timeout -k 10 300 python tools/path_floor.py > gpurun_out/r5/path_floor_cap64.json 2> gpurun_out/r5/path_floor_cap64.err; echo "floor cap64 rc $?"
timeout -k 10 300 python tools/path_floor.py > gpurun_out/r5/path_floor_product.json 2> gpurun_out/r5/path_floor_product.err; echo "floor product rc $?"
timeout -k 10 300 python tools/sweep_env.py "" "ADYPT_REF_TRIANGLES_MAX_MB=0" "" "ADYPT_REF_TRIANGLES_MAX_MB=0" > gpurun_out/r5/refcopy_sponza.jsonl 2> gpurun_out/r5/refcopy_sponza.err; echo "sweep sponza rc $?"
SWEEP_SCENE=sanmiguel timeout -k 10 600 python tools/sweep_env.py "" "ADYPT_REF_TRIANGLES_MAX_MB=0" "" "ADYPT_REF_TRIANGLES_MAX_MB=0" > gpurun_out/r5/refcopy_sanmiguel.jsonl 2> gpurun_out/r5/refcopy_sanmiguel.err; echo "sweep sanmiguel rc $?"
cat gpurun_out/r5/refcopy_*.jsonl
