set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 300 python -m pytest tests/test_gpu_frames_in_flight.py -x -q -k "single_frames or bit_invariant" > gpurun_out/r5/gputest_roll.log 2>&1; rc=$?; echo "pytest(rolling) rc $rc"; tail -4 gpurun_out/r5/gputest_roll.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r5/gputest.log; tail -3 gpurun_out/r5/gputest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench_driver.json 2> gpurun_out/r5/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_driver.json').read().strip().splitlines()[-1])
print("value",d["value"],d["value_min"],d["value_max"],"k_path",d["roofline"]["kernel_Mrays_s"],"hbm",d["roofline_hbm_resident"]["kernel_Mrays_s"],d["roofline_hbm_resident"]["whole_frame_Mrays_s"],"primary",d["primary_only"]["Mrays_s_per_call"],"single",d["single_frame"]["Mrays_s"],d["single_frame"]["one_frame_per_pass"],"life1",d["tmp_lifetime_1"]["Mrays_s"])
PY
ADYPT_LIB=$PWD/adypt_amd/libadypt_blockcnt.so timeout -k 10 300 python tools/path_block_counts.py > gpurun_out/r5/block_counts.json 2> gpurun_out/r5/block_counts.err; echo "blockcnt rc $?"; cat gpurun_out/r5/block_counts.json
PMC_QUICK_COPY=gpurun_out/r5/pmc_quick.txt timeout -k 10 600 bash tools/pmc_quick.sh > gpurun_out/r5/pmc_quick.log 2>&1; echo "pmc rc $?"; grep -A12 "^k_path<false" gpurun_out/r5/pmc_quick.txt | head -30
