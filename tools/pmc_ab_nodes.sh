#!/bin/bash
# GPU box: the 10 M-triangle stand-in (BASELINE config 4) with 80-byte nodes packed (product) and at a 128-byte stride (adypt_amd/libadypt_node128.so, measurement
# variant adypt_amd/csrc/measure/k_node_stride128.py): k_path's rate and the memory-side counters per node visit — does the gather microbenchmark's 1.49 x for
# line-aligned records (profiles/r5_gather_roof.json) transfer to the kernel, and if not, where does it go?    bash tools/pmc_ab_nodes.sh > gpurun_out/r6_node_stride.txt
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
OUT=gpurun_out/pmc_ab_nodes; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --scene sanmiguel --steps 32 --warmup 16 --repeats 1 --no-cpu-baseline --no-single-frame --no-extra-blocks --no-hbm-block"
$CMD > /dev/null 2>&1
for lib in default libadypt_node128.so; do
  if [ "$lib" != "default" ]; then export ADYPT_LIB=$PWD/adypt_amd/$lib; else unset ADYPT_LIB; fi
  D=$OUT/$lib; mkdir -p $D
  $CMD > $D/plain.json 2> /dev/null
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $D/pmc_$tag -- $CMD > $D/pmc_$tag.json 2> $D/pmc_$tag.err
  done
  python3 tools/pmc_summary.py $D > $D/summary.txt
  find $D -name "*counter_collection.csv" -delete
  echo "== $lib"; grep -A16 "^k_path<false" $D/summary.txt | sed -n 1,17p
  python3 - $D/plain.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("   line", d["value"], "kernel", r["kernel_Mrays_s"], "nodes_per_ray", r["nodes_per_ray"], "tris_per_ray", r["tris_per_ray"], "kernel_rays", r["kernel_rays"], "+warmup", r["kernel_rays_warmup"])
PY
done
