set -u
export ADYPT_CACHE=/tmp/adypt_cache
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_driver.json 2> gpurun_out/r3_bench_driver.err && tail -c 300 gpurun_out/r3_bench_driver.json && \
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err && \
python bench.py --scene salle --width 4096 --height 4096 --no-hbm-block --no-cpu-baseline --no-single-frame > gpurun_out/r3_bench_salle4096.json 2> gpurun_out/r3_bench_salle.err && \
python bench.py --tmp-lifetime 1 --no-hbm-block --no-cpu-baseline --no-single-frame > gpurun_out/r3_bench_life1.json 2>/dev/null && \
python tools/primary_rate.py > gpurun_out/r3_primary_rate.txt 2>&1 && \
python tools/shard_emulate.py 20 > gpurun_out/r3_shard20.txt 2>&1 && python tools/shard_emulate.py 64 > gpurun_out/r3_shard64.txt 2>&1 && python tools/shard_emulate.py 128 > gpurun_out/r3_shard128.txt 2>&1
echo done
