set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r5/gputest.log; tail -3 gpurun_out/r5/gputest.log
[ $rc -eq 0 ] || exit 1
ADYPT_LIB=$PWD/adypt_amd/libadypt_blockcnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/block_counts.json 2> gpurun_out/r5/block_counts.err; echo "blockcnt rc $?"
ADYPT_BLOCKS_SET=shade ADYPT_LIB=$PWD/adypt_amd/libadypt_shadecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/shade_block_counts.json 2> gpurun_out/r5/shade_block_counts.err; echo "shadecnt rc $?"
ADYPT_BLOCKS_SET=rare ADYPT_LIB=$PWD/adypt_amd/libadypt_rarecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/rare_block_counts.json 2> gpurun_out/r5/rare_block_counts.err; echo "rarecnt rc $?"
timeout -k 10 900 bash tools/collect_profiles.sh > gpurun_out/r5/collect_bench.log 2>&1; echo "collect bench rc $?"; grep -E "valu_insts_per_ray|source_hash" gpurun_out/profiles_bench/pmc_profile.json
