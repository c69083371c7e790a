set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
O=gpurun_out/r5/final; mkdir -p $O
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2> $O/driver_command.err; echo "driver rc $?"
timeout -k 10 600 python bench.py > $O/default.json 2> $O/default.err; echo "default rc $?"
ADYPT_FUSED_BOUNCES=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-hbm-block --no-cpu-baseline > $O/launch_per_bounce.json 2>/dev/null; echo "lpb rc $?"
timeout -k 10 400 python bench.py --scene salle --width 4096 --height 4096 --steps 8 --warmup 2 --no-hbm-block --no-cpu-baseline > $O/salle.json 2>/dev/null; echo "salle rc $?"
for n in 1 2 4 8; do SWEEP_NRANKS=$n timeout -k 10 300 python tools/path_sweep.py 1 "-:" "-:ADYPT_FUSED_BOUNCES=0"; done > $O/shard.log 2> $O/shard.err; echo "shard rc $?"
timeout -k 10 300 python tools/primary_rate.py 200 > $O/primary.log 2>&1; echo "primary rc $?"
timeout -k 10 600 python tools/shard_breakdown.py > $O/shard_breakdown.log 2> $O/shard_breakdown.err; echo "breakdown rc $?"
python - <<'PY'
import json
for n in ("driver_command","default","launch_per_bounce","salle"):
    d=json.loads(open('gpurun_out/r5/final/%s.json'%n).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
    print(n, d["value"], d["value_min"], d["value_max"], "k", r.get("kernel_Mrays_s"), "stale", r.get("pmc_stale"), "valu", (r.get("valu_issue") or {}).get("frac_range"), "hbm", (d.get("roofline_hbm_resident") or {}).get("kernel_Mrays_s"), (d.get("roofline_hbm_resident") or {}).get("frac_of_gather_roof"), "single", {k:v for k,v in ((d.get("single_frame") or {}).get("one_frame_per_pass") or {}).items() if k!="note"})
PY
