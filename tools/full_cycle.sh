#!/bin/bash
# GPU box: everything profiles/r6_* is made of, for the tree as it stands, in one gpurun call (~8 minutes).  Needs the six counting variants
# (see tools/path_block_counts.py; built here on the CPU side: adypt_amd/libadypt_{blockcnt,shadecnt,rarecnt,lanes_trip,lanes_shade,lanes_rare,lanes_wait}.so; tools/build_counting_variants.sh builds them).
#   gpurun --timeout 1150 -- 'bash tools/full_cycle.sh'        then, here:  bash tools/finalize_counter_profiles.sh      (counter profiles of this tree into profiles/)
#   gpurun --timeout 1150 -- 'bash tools/full_cycle.sh lines'  then, here:  python tools/finalize_bench_profiles.py        (the bench lines, now with those profiles: pmc_stale false)
# Stops at the first failing step (a failed GPU test means no profile of this tree is wanted).
set -u
export ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache} TMPDIR=/tmp
C=gpurun_out/r6/counts; O=gpurun_out/r6/final; rm -rf $O; mkdir -p $O
if [ "${1:-all}" != "lines" ]; then
rm -rf $C; mkdir -p $C
echo "[cycle] pytest -m gpu"; timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r6/pytest_gpu.txt 2>&1 || { tail -5 gpurun_out/r6/pytest_gpu.txt; exit 1; }
tail -1 gpurun_out/r6/pytest_gpu.txt
echo "[cycle] block and lane counts"
ADYPT_LIB=adypt_amd/libadypt_blockcnt.so timeout -k 10 100 python tools/path_block_counts.py > $C/trip.json 2> $C/err.txt || exit 1
ADYPT_BLOCKS_SET=shade ADYPT_LIB=adypt_amd/libadypt_shadecnt.so timeout -k 10 100 python tools/path_block_counts.py > $C/shade.json 2>> $C/err.txt || exit 1
ADYPT_BLOCKS_SET=rare ADYPT_LIB=adypt_amd/libadypt_rarecnt.so timeout -k 10 100 python tools/path_block_counts.py > $C/rare.json 2>> $C/err.txt || exit 1
for s in trip shade rare wait; do ADYPT_BLOCKS_LANES=1 ADYPT_BLOCKS_SET=$s ADYPT_LIB=adypt_amd/libadypt_lanes_$s.so timeout -k 10 100 python tools/path_block_counts.py > $C/lanes_$s.json 2>> $C/err.txt || exit 1; done
echo "[cycle] counter profiles, bench scene"; bash tools/collect_profiles.sh > gpurun_out/collect_bench.log 2>&1 || exit 1
echo "[cycle] counter profiles, 10 M-triangle scene"; bash tools/collect_profiles.sh sanmiguel > gpurun_out/collect_sanmiguel.log 2>&1 || exit 1
echo "[cycle] counter profiles, primary rays only (config 2)"; bash tools/collect_profiles.sh primary > gpurun_out/collect_primary.log 2>&1 || exit 1
fi
echo "[cycle] bench lines"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2> $O/driver_command.err || exit 1
python bench.py > $O/default.json 2> $O/default.err || exit 1
ADYPT_FUSED_BOUNCES=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-hbm-block --no-cpu-baseline > $O/launch_per_bounce.json 2> /dev/null || exit 1
ADYPT_RARE_MIN=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-hbm-block --no-cpu-baseline --no-single-frame --no-extra-blocks > $O/no_deferral.json 2> /dev/null || exit 1
python bench.py --scene salle --width 4096 --height 4096 --steps 8 --warmup 2 --no-hbm-block --no-cpu-baseline > $O/salle.json 2> /dev/null || exit 1
for n in 1 2 4 8; do SWEEP_NRANKS=$n python tools/path_sweep.py 1 "-:" "-:ADYPT_FUSED_BOUNCES=0"; done > $O/shard.log 2> $O/shard.err || exit 1
python tools/primary_rate.py 200 > $O/primary.log 2> /dev/null || exit 1
python tools/shard_breakdown.py > $O/shard_breakdown.log 2> $O/shard_breakdown.err || exit 1
python - <<'PY'
import json
for f in ("driver_command", "default", "no_deferral", "launch_per_bounce"):
    d = json.loads(open("gpurun_out/r6/final/%s.json" % f).read().strip().splitlines()[-1])
    print("[cycle]", f, d["value"], d["ms_per_step"], d["roofline"].get("kernel_Mrays_s"), "pmc_stale", d["roofline"].get("pmc_stale"))
PY
