#!/bin/bash
# GPU box: rocprofv3 --pmc passes over the bench workload for the CU's vector-memory pipeline (texture addresser TA, texture data TD, L1
# = TCP) and the LDS: is the pipeline that profiles/r2_ablations_k_trace.txt calls the co-limiter of k_trace busy by its own counters?
# One group per pass (a block has few counters per pass); a group the profiler rejects leaves its .err behind and is skipped.
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
OUT=gpurun_out/pmc_vmem; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-hbm-block --no-single-frame"
python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-hbm-block --no-single-frame > /dev/null 2>&1
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_BUSY_avr TA_BUSY_max TA_BUSY_min" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$i -- $CMD > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "group $i rejected: $grp"
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT -name "*counter_collection.csv" -delete
grep -A40 "^k_path<false\|^k_trace<false" $OUT/pmc_summary.txt | sed -n 1,90p
grep -A30 "^k_shade" $OUT/pmc_summary.txt | sed -n 1,32p
