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
