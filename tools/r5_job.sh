set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
O=gpurun_out/r5/final; mkdir -p $O
timeout -k 10 1000 bash tools/collect_profiles.sh sanmiguel > gpurun_out/r5/collect_sanmiguel.log 2>&1; echo "collect sanmiguel rc $?"; grep -E "valu_insts_per_ray|source_hash|traffic_bytes_per_ray\"" gpurun_out/profiles_sanmiguel/pmc_profile.json
