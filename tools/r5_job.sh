set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
ADYPT_LIB=$PWD/adypt_amd/libadypt_rcp.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fused_bounces.py -x -q > gpurun_out/r5/gputest_rcp.log 2>&1; echo "pytest(rcp) rc $?"; tail -2 gpurun_out/r5/gputest_rcp.log
timeout -k 10 600 python tools/ab.py default libadypt_rcp.so > gpurun_out/r5/ab_rcp.jsonl 2>&1; cat gpurun_out/r5/ab_rcp.jsonl
