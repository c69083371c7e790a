set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fused_bounces.py tests/test_gpu_frames_in_flight.py -x -q > gpurun_out/r5/gputest_quick.log 2>&1; rc=$?; echo "pytest(quick) rc $rc"; tail -2 gpurun_out/r5/gputest_quick.log
[ $rc -eq 0 ] || exit 1
ADYPT_LIB=$PWD/adypt_amd/libadypt_blockcnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/block_counts.json 2> gpurun_out/r5/block_counts.err; echo "blockcnt rc $?"
ADYPT_BLOCKS_SET=shade ADYPT_LIB=$PWD/adypt_amd/libadypt_shadecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/shade_block_counts.json 2> gpurun_out/r5/shade_block_counts.err; echo "shadecnt rc $?"
ADYPT_BLOCKS_SET=rare ADYPT_LIB=$PWD/adypt_amd/libadypt_rarecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/rare_block_counts.json 2> gpurun_out/r5/rare_block_counts.err; echo "rarecnt rc $?"; cat gpurun_out/r5/rare_block_counts.json
timeout -k 10 900 bash tools/collect_profiles.sh > gpurun_out/r5/collect_bench.log 2>&1; echo "collect bench rc $?"; grep -E "valu_insts_per_ray|traffic_bytes_per_ray\"|source_hash|effective_clock" gpurun_out/profiles_bench/pmc_profile.json
