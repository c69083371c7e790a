set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
ADYPT_LIB=$PWD/adypt_amd/libadypt_blockcnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/block_counts.json 2> gpurun_out/r5/block_counts.err; echo "blockcnt rc $?"
ADYPT_BLOCKS_SET=shade ADYPT_LIB=$PWD/adypt_amd/libadypt_shadecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/shade_block_counts.json 2> gpurun_out/r5/shade_block_counts.err; echo "shadecnt rc $?"; cat gpurun_out/r5/shade_block_counts.json
timeout -k 10 900 bash tools/collect_profiles.sh > gpurun_out/r5/collect_bench.log 2>&1; echo "collect bench rc $?"; tail -40 gpurun_out/r5/collect_bench.log | cut -c1-220
