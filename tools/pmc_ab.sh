#!/bin/bash
# GPU box: the SQ issue / wait / LDS counters of k_path for two or more builds of the library, side by side (developer tool for A/B work on the trip).
#   bash tools/pmc_ab.sh libadypt_base.so default        -> gpurun_out/pmc_ab/<lib>.txt (k_path<false> rows), one rocprofv3 --pmc pass per counter group and build
set -u
export TMPDIR=/tmp ADYPT_CACHE=${ADYPT_CACHE:-/tmp/adypt_cache}
OUT=gpurun_out/pmc_ab; rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-block --no-single-frame --no-extra-blocks --repeats 1"
$CMD > /dev/null 2>&1
for lib in "$@"; do
  if [ "$lib" != "default" ]; then export ADYPT_LIB=$PWD/adypt_amd/$lib; else unset ADYPT_LIB; fi
  D=$OUT/$lib; mkdir -p $D
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $D/pmc_$tag -- $CMD > $D/pmc_$tag.json 2> $D/pmc_$tag.err
  done
  python3 tools/pmc_summary.py $D > $D/summary.txt
  find $D -name "*counter_collection.csv" -delete
  echo "== $lib"; grep -A30 "^k_path<false" $D/summary.txt | sed -n 1,28p; grep -h '"value"' $D/pmc_SQ_WAVES.json | python3 -c "import sys, json; [print('   line', json.loads(l)['value'], json.loads(l)['roofline'].get('kernel_Mrays_s')) for l in sys.stdin]"
done
