set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
ADYPT_LIB=$PWD/adypt_amd/libadypt_blockcnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/block_counts.json 2> gpurun_out/r5/block_counts.err; echo "blockcnt rc $?"
ADYPT_BLOCKS_SET=shade ADYPT_LIB=$PWD/adypt_amd/libadypt_shadecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/shade_block_counts.json 2> gpurun_out/r5/shade_block_counts.err; echo "shadecnt rc $?"
ADYPT_BLOCKS_SET=rare ADYPT_LIB=$PWD/adypt_amd/libadypt_rarecnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/rare_block_counts.json 2> gpurun_out/r5/rare_block_counts.err; echo "rarecnt rc $?"
timeout -k 10 1000 bash tools/collect_profiles.sh sanmiguel > gpurun_out/r5/collect_sanmiguel.log 2>&1; echo "collect sanmiguel rc $?"; grep -E "valu_insts_per_ray|source_hash" gpurun_out/profiles_sanmiguel/pmc_profile.json
