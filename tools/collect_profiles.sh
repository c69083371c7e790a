#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py.  Outputs under gpurun_out/profiles_<tag>/ (copy what is to be judged into profiles/).
#   bash tools/collect_profiles.sh            the bench scene at the driver's command line (--steps 20 --warmup 5)
#   bash tools/collect_profiles.sh sanmiguel  the 10 M-triangle stand-in (roofline_hbm_resident)
# Kernel-trace/stats and every --pmc group are separate runs (gpurun refuses --pmc combined with trace domains other than
# kernel-trace).  The kernel-trace pass runs the driver's exact command; the PMC passes add the flags that leave only ONE repeat of the
# warm-up and the timed frames of ONE scene in the process, so that every launch of the dominant kernel the counters see belongs to them.
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
WHAT=${1:-bench}
if [ "$WHAT" = "sanmiguel" ]; then
  TAG=sanmiguel; FULL="python3 bench.py --scene sanmiguel --steps 32 --warmup 16 --repeats 1 --no-cpu-baseline --no-single-frame --no-extra-blocks"
elif [ "$WHAT" = "primary" ]; then
  # BASELINE config 2: the counters of k_trace_camera<false, true> — a kernel name only bench.py's primary_only block launches (8 warm-up + 128 timed calls);
  # four counter groups, the extra blocks left ON (tools/pmc_profile.py reads that block: PMC_BLOCK=primary_only)
  TAG=primary; OUT=gpurun_out/profiles_$TAG; rm -rf $OUT; mkdir -p $OUT
  CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-block --no-single-frame --repeats 1"
  $CMD > $OUT/plain_run.json 2> /dev/null
  for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- $CMD > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  done
  python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
  PMC_BLOCK=primary_only python3 tools/pmc_profile.py $OUT $OUT/pmc_FETCH_SIZE.json "rocprofv3 --pmc <group> -- $CMD" > $OUT/pmc_profile.json
  find $OUT -name "*counter_collection.csv" -delete
  grep -A14 "^k_trace_camera<false, true" $OUT/pmc_summary.txt; cat $OUT/pmc_profile.json
  exit 0
else
  TAG=bench; FULL="python3 bench.py --gpus 1 --steps 20 --warmup 5"
fi
CMD="$FULL --no-cpu-baseline --no-hbm-block --no-single-frame --no-extra-blocks --repeats 1"
OUT=gpurun_out/profiles_$TAG; rm -rf $OUT; mkdir -p $OUT
$CMD > $OUT/plain_run.json 2> /dev/null   # builds the scene cache outside the profiled runs; also the un-profiled reference line
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $FULL > $OUT/stats_bench.json 2> $OUT/stats.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 tools/kernel_trace_phases.py $OUT/stats $OUT/stats_bench.json > $OUT/kernel_trace_phases.json
# (the translation and fabric-latency groups: what a table beyond the caches costs — VERDICT r4 task 1; names as rocprofv3 -L lists them on this release)
for grp in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- $CMD > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python3 tools/pmc_profile.py $OUT $OUT/pmc_FETCH_SIZE.json "rocprofv3 --pmc <group> -- $CMD" $OUT/kernel_trace_phases.json > $OUT/pmc_profile.json
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
head -8 $OUT/kernel_stats.csv | cut -c1-170; grep -A14 "^k_path<false\|^k_trace<false" $OUT/pmc_summary.txt; cat $OUT/pmc_profile.json
