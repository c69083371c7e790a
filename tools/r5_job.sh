set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
O=gpurun_out/r5/final; mkdir -p $O
timeout -k 10 1000 bash tools/collect_profiles.sh sanmiguel > gpurun_out/r5/collect_sanmiguel.log 2>&1; echo "collect sanmiguel rc $?"; grep -E "valu_insts_per_ray|source_hash" gpurun_out/profiles_sanmiguel/pmc_profile.json
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2> $O/driver_command.err; echo "driver rc $?"
