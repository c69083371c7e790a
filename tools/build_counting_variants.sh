#!/bin/bash
# CPU side: the seven counting variants of the library tools/full_cycle.sh runs on the GPU box (source transform adypt_amd/csrc/measure/k_path_blocks.py):
#   wave entries per block: blockcnt (trip set), shadecnt, rarecnt;  active lanes per block: lanes_trip, lanes_shade, lanes_rare, lanes_wait
set -e
cd "$(dirname "$0")/.."
T=adypt_amd/csrc/measure/k_path_blocks.py
ADYPT_BLOCKS_COUNT=1 tools/build_variant.sh blockcnt --transform $T 2>&1 | tail -1
ADYPT_BLOCKS_COUNT=1 ADYPT_BLOCKS_SET=shade tools/build_variant.sh shadecnt --transform $T 2>&1 | tail -1
ADYPT_BLOCKS_COUNT=1 ADYPT_BLOCKS_SET=rare tools/build_variant.sh rarecnt --transform $T 2>&1 | tail -1
for s in trip shade rare wait; do ADYPT_BLOCKS_COUNT=1 ADYPT_BLOCKS_LANES=1 ADYPT_BLOCKS_SET=$s tools/build_variant.sh lanes_$s --transform $T 2>&1 | tail -1; done
