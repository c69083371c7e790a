set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
O=gpurun_out/r5/final; mkdir -p $O
for n in 1 2 4 8; do SWEEP_NRANKS=$n timeout -k 10 300 python tools/path_sweep.py 1 "-:" "-:ADYPT_FUSED_BOUNCES=0"; done > $O/shard.log 2> $O/shard.err; echo "shard rc $?"
timeout -k 10 300 python tools/primary_rate.py 200 > $O/primary.log 2>&1; echo "primary rc $?"
timeout -k 10 600 python tools/shard_breakdown.py > $O/shard_breakdown.log 2> $O/shard_breakdown.err; echo "breakdown rc $?"; tail -3 $O/shard_breakdown.log | cut -c1-300
