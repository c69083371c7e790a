set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 1000 bash tools/collect_profiles.sh sanmiguel > gpurun_out/r5/collect_sanmiguel.log 2>&1; echo "collect sanmiguel rc $?"; grep -E "valu_insts_per_ray|source_hash" gpurun_out/profiles_sanmiguel/pmc_profile.json
