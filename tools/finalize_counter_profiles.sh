#!/bin/bash
# Developer tool: copy what tools/collect_profiles.sh [sanmiguel] and tools/path_block_counts.py left under gpurun_out/ into profiles/r6_* and
# regenerate the trip budget and the issue model from the current sources (CPU only).   bash tools/finalize_counter_profiles.sh [dir of the counts]
# (dir of the counts, default gpurun_out/r6/counts: trip.json, shade.json, rare.json = tools/path_block_counts.py with the three counting variants)
set -e
cd "$(dirname "$0")/.."
for t in bench sanmiguel; do
  d=gpurun_out/profiles_$t
  cp $d/pmc_profile.json profiles/r6_pmc_$t.json; cp $d/pmc_summary.txt profiles/r6_pmc_summary_$t.txt; cp $d/kernel_stats.csv profiles/r6_kernel_stats_$t.csv
  cp $d/kernel_trace_phases.json profiles/r6_kernel_trace_phases_$t.json; cp $d/stats_bench.json profiles/r6_bench_json_under_kernel_trace_$t.json
done
if [ -f gpurun_out/profiles_primary/pmc_profile.json ]; then cp gpurun_out/profiles_primary/pmc_profile.json profiles/r6_pmc_primary.json; cp gpurun_out/profiles_primary/pmc_summary.txt profiles/r6_pmc_summary_primary.txt; fi
C=${1:-gpurun_out/r6/counts}
python3 tools/lane_counts_profile.py $C > profiles/r6_k_path_lane_counts.json
cp $C/trip.json profiles/r6_k_path_block_counts.json; cp $C/shade.json profiles/r6_k_path_shade_block_counts.json; cp $C/rare.json profiles/r6_k_path_rare_block_counts.json
python3 tools/trip_budget.py > profiles/r6_trip_budget.json
python3 tools/valu_issue_model.py > profiles/r6_valu_issue_model.json
echo "sources $(python3 tools/source_hash.py); profiles: $(grep -o '"source_hash": "[0-9a-f]*"' profiles/r6_pmc_bench.json profiles/r6_pmc_sanmiguel.json | tr '\n' ' ')"
