#!/bin/bash
# Developer tool: copy what tools/collect_profiles.sh [sanmiguel] and tools/path_profile.py left under gpurun_out/ into profiles/r4_* and
# regenerate the instruction mix and the issue model from the current sources (CPU only).   bash tools/finalize_counter_profiles.sh
set -e
cd "$(dirname "$0")/.."
for t in bench sanmiguel; do
  d=gpurun_out/profiles_$t
  cp $d/pmc_profile.json profiles/r4_pmc_$t.json; cp $d/pmc_summary.txt profiles/r4_pmc_summary_$t.txt; cp $d/kernel_stats.csv profiles/r4_kernel_stats_$t.csv
  cp $d/kernel_trace_phases.json profiles/r4_kernel_trace_phases_$t.json; cp $d/stats_bench.json profiles/r4_bench_json_under_kernel_trace_$t.json
done
cp gpurun_out/r4/path_profile.json profiles/r4_k_path_wave_profile.json
python3 tools/instruction_mix.py k_path > profiles/r4_k_path_instruction_mix.json
python3 tools/valu_issue_model.py > profiles/r4_valu_issue_model.json
echo "sources $(python3 tools/source_hash.py); profiles: $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r4_pmc_bench.json profiles/r4_pmc_sanmiguel.json | tr '\n' ' ')"
