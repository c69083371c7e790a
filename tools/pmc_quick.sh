#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (SQ issue/wait counters) of the bench workload; prints the k_trace<false,false> rows.
#   bash tools/pmc_quick.sh [extra env assignments are inherited]
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
OUT=gpurun_out/pmc_quick; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-hbm-block --no-single-frame"
python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-hbm-block --no-single-frame > /dev/null 2>&1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- $CMD > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*counter_collection.csv" -delete
cp $OUT/pmc_summary.txt ${PMC_QUICK_COPY:-$OUT/pmc_summary_copy.txt}; grep -A14 "^k_trace<false\|^k_path<false\|^k_shade(" $OUT/pmc_summary.txt
