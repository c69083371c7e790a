set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 600 python tools/sweep_env.py "" "ADYPT_REFILL_MIN=12" "ADYPT_REFILL_MIN=20" "ADYPT_REFILL_MIN=24" "ADYPT_SHADE_MIN=48" "ADYPT_SHADE_MIN=56" "" "ADYPT_PATH_LDS_DEPTH=3" "ADYPT_REFILL_MIN=8" > gpurun_out/r5/sweep_sponza.jsonl 2>/dev/null; cat gpurun_out/r5/sweep_sponza.jsonl
timeout -k 10 600 python tools/ab.py default libadypt_s320.so libadypt_s384.so libadypt_prio1.so libadypt_prio3.so > gpurun_out/r5/ab_slots.jsonl 2>/dev/null; cat gpurun_out/r5/ab_slots.jsonl
