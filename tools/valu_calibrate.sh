#!/bin/bash
# GPU box: what the vector-ALU issue counters read on kernels of ONE instruction each (tools/microbench/valu_throughput*.hip, 5 waves per
# SIMD, every SIMD busy) — true cycles per wave-instruction per SIMD from GRBM_GUI_ACTIVE (no assumed clock), and what SQ_ACTIVE_INST_VALU
# counts per instruction.  Output: gpurun_out/valu_calibration.json (copy into profiles/).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/valu_calib; rm -rf $OUT; mkdir -p $OUT
for b in valu_throughput valu_throughput2; do
  [ -x tools/microbench/$b.bin ] || hipcc --offload-arch=gfx950 -O3 -o tools/microbench/$b.bin tools/microbench/$b.hip || exit 1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$b -- tools/microbench/$b.bin > $OUT/$b.txt 2> $OUT/$b.err || exit 1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq_$b -- tools/microbench/$b.bin > /dev/null 2> $OUT/pmc_sq_$b.err || exit 1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm_$b -- tools/microbench/$b.bin > /dev/null 2> $OUT/pmc_grbm_$b.err || exit 1
done
python3 tools/valu_calibrate.py $OUT > gpurun_out/valu_calibration.json && cat gpurun_out/valu_calibration.json
