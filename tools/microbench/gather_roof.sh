#!/bin/bash
# GPU box: the gather roof of k_path's access pattern (tools/microbench/gather_roof.hip) + the counters that explain it.
#   bash tools/microbench/gather_roof.sh        -> gpurun_out/gather_roof/{plain.jsonl, pmc_*.jsonl, gather_roof.json, counters_list.txt}
# One plain pass (timing: best of 3 launches per configuration), then one rocprofv3 --pmc pass per counter group with ONE launch per
# configuration (gpurun refuses --pmc combined with trace domains; every group is its own process).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/gather_roof; rm -rf $OUT; mkdir -p $OUT
BIN=tools/microbench/gather_roof.bin
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/microbench/gather_roof.hip || exit 1
timeout 300 $BIN --wgs 6,8 --chains 1,2,4 > $OUT/plain.jsonl 2> $OUT/plain.err || { echo "plain pass failed"; cat $OUT/plain.err; exit 1; }
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# every translation / memory-side counter this rocprofv3 knows (names differ between ROCm releases: probe, do not assume)
grep -o -E "\b(TCP_UTCL1_[A-Z0-9_]+|TCP_UTCL2_[A-Z0-9_]+|UTCL2_[A-Z0-9_]+|TCC_EA0_RDREQ_LEVEL[A-Z0-9_]*|TCC_EA0_RD_UNCACHED[A-Z0-9_]*|TCC_TAG_STALL[A-Z0-9_]*|TCP_PENDING_STALL_CYCLES[A-Z0-9_]*|TCP_TCC_READ_REQ_LATENCY[A-Z0-9_]*|TCP_TCC_READ_REQ[A-Z0-9_]*|TCP_TA_TCP_STATE_READ[A-Z0-9_]*|TCC_EA0_RDREQ_DRAM[A-Z0-9_]*|TCC_EA0_RDREQ_32B[A-Z0-9_]*|TCC_BUBBLE[A-Z0-9_]*)\b" $OUT/counters_list.txt | sort -u > $OUT/counters_of_interest.txt
PMCARGS="--pmc --wgs 6 --chains 1,2"
run_group() { # tag, counters...
  local tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$tag -- $BIN $PMCARGS > $OUT/pmc_$tag.jsonl 2> $OUT/pmc_$tag.err || echo "group $tag failed: $(tail -2 $OUT/pmc_$tag.err)"
}
run_group fetch FETCH_SIZE
run_group tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run_group busy GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
# translation and memory-side counters (the _sum forms rocprofv3 -L lists for this release; a group that does not exist fails on its own)
run_group utcl1a TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum
run_group utcl1b TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run_group utcl1c TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
run_group utcl1d TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
run_group ealevel TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
run_group eadram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum
run_group tcplat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run_group tcpstall TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum
python3 tools/microbench/gather_roof_join.py $OUT > $OUT/gather_roof.json
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/gather_roof.json | head -c 6000
