#!/bin/bash
# GPU box: rocprofv3 evidence for the bench.py workload (same command, fewer steps).  Outputs under gpurun_out/profiles/.
# Kernel-trace/stats and every --pmc group are separate runs (gpurun refuses --pmc combined with trace domains other than kernel-trace).
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
OUT=gpurun_out/profiles; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline"   # --warmup 0: every launch the profiler sees is a timed one
python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline > /dev/null 2>&1   # builds the scene cache outside the profiled runs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats_bench.json 2> $OUT/stats.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- $CMD > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python3 tools/pmc_traffic.py $OUT $OUT/stats_bench.json > $OUT/pmc_traffic.json
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -8 $OUT/kernel_stats.csv | cut -c1-170; grep -A12 "^k_trace<false" $OUT/pmc_summary.txt; cat $OUT/pmc_traffic.json
