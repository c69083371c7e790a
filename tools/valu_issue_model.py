"""Build profiles/r6_valu_issue_model.json (read by bench.py): the vector-ALU issue model of k_path<false>, weighted by EXECUTED instructions.

  static counts   profiles/r6_trip_budget.json      tools/trip_budget.py: vector instructions per block of the persistent loop, by issue class (no GPU)
  block entries   profiles/r6_k_path_block_counts.json, r6_k_path_shade_block_counts.json, r6_k_path_rare_block_counts.json
                                                     tools/path_block_counts.py: how often a wave enters each block (three counting variants, GPU)
  issue cycles    profiles/r3_valu_calibration.json  one-instruction loops: TRUE cycles a wave-instruction holds its SIMD's issue, per class
  check           profiles/r6_pmc_bench.json         SQ_INSTS_VALU per ray of the product kernel (rocprofv3 --pmc, tools/collect_profiles.sh)

executed instructions per wave-trip = sum over blocks (static x entries / trips); it must reproduce SQ_INSTS_VALU per wave-trip
(= SQ_INSTS_VALU per ray / wave-trips per ray) within 3 % — tests/test_profiles_consistency.py asserts it.  Round 4's model weighted a static
count of the whole loop and was off by 65 %.
    python tools/valu_issue_model.py > profiles/r6_valu_issue_model.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(ROOT, "profiles", n)
cal = {r["kernel"]: r for r in json.load(open(P("r3_valu_calibration.json")))["rows"]}
budget = json.load(open(P("r6_trip_budget.json")))
counts = json.load(open(P("r6_k_path_block_counts.json")))
pmc = json.load(open(P("r6_pmc_bench.json"))) if os.path.exists(P("r6_pmc_bench.json")) else None
old = json.load(open(P("r4_k_path_instruction_mix.json")))

CLASSES = [
    ("normal", "normal rate (v_cvt_f32_ubyte, min/max/min3/max3, v_cmp, v_cndmask, bit-field, shifts, v_perm, SDWA, DPP moves ...)", 4,
     ["k_cvt_ubyte", "k_max_f32", "k_min_f32", "k_max3_f32", "k_med3", "k_cmp", "k_cmp_sgpr", "k_cnd_sgpr", "k_bfe", "k_lshl", "k_lshl_or", "k_perm",
      "k_and_or", "k_bfi", "k_lshl_add", "k_or3", "k_mul_lo", "k_div_fixup", "k_sdwa_cvt"]),
    ("full", "full rate (fp32 add/mul/fma, and/or/xor, integer add, arithmetic shift right, moves)", 2,
     ["k_add_f32", "k_fma_f32", "k_mul_f32", "k_sub_f32", "k_and_b32", "k_or_b32", "k_xor_b32", "k_add_u32", "k_ashr", "k_mov"]),
    ("packed64", "packed fp32 (v_pk_fma_f32) / 64-bit", 4, ["k_pk_fma", "k_mad_u64"]),
    ("trans", "transcendental (v_rcp_f32, v_sqrt_f32)", 8, ["k_rcp", "k_sqrt"]),
]
# entries per trip of every counted block: the trip set directly; the shade and rare sets (other launches of the same frames) through their own `shade` / `trip` counts
entries = counts["wave_entries"]
trips = float(entries["trip"])
w = {k: v / trips for k, v in entries.items()}
for name in ("r6_k_path_shade_block_counts.json", "r6_k_path_rare_block_counts.json"):
    if not os.path.exists(P(name)):
        continue
    e = json.load(open(P(name)))["wave_entries"]
    scale = (1.0 / e["trip"]) if "trip" in e else (w["shade"] / e["shade"])
    for k, v in e.items():
        w.setdefault(k, v * scale)

# the two exclusive branches of the exchange carry marks but no counter: a round follows (= `shade`) or rays are taken (= `exchange` - `shade`)
w.setdefault("X_pick", w["shade"]); w.setdefault("X_take", w["exchange"] - w["shade"])

B = budget["blocks"]


def weight(owner):
    """executions per trip of the block that owns an instruction: its own counter, else the block it sits in"""
    if owner.startswith("trip/"):
        return 1.0
    if owner == "loop control":
        return 1.0 + w["setup"]  # once per iteration: per trip, and per iteration that starts rays instead (a setup follows an exchange)
    if owner == "outside the loop":
        return 0.0
    if owner in w:
        return w[owner] / B[owner].get("instances", 1)  # (equal-sized instances of one inlined block share a counter: static sum x entries / instances)
    parent = B[owner].get("parent")
    return weight(parent) if parent else 1.0


blocks = [(owner, weight(owner), c) for owner, c in B.items() if owner != "outside the loop"]
per_class = {c[0]: 0.0 for c in CLASSES}
rows, total = [], 0.0
for name, weight, c in blocks:
    ex = weight * c["valu"]
    total += ex
    for k, n in c.get("by_class", {}).items():
        per_class[k] += weight * n
    rows.append({"block": name, "static_valu": c["valu"], "executions_per_trip": round(weight, 4), "executed_valu_per_trip": round(ex, 1)})
classes, arch, loops = [], 0.0, 0.0
for key, label, cycles, kernels in CLASSES:
    share = per_class[key] / total
    m = sum(cal[k]["issue_cycles_per_inst_per_simd"] for k in kernels) / len(kernels)
    classes.append({"class": label, "share": round(share, 4), "cycles_architectural": cycles, "cycles_single_class_loop": round(m, 3)})
    arch += share * cycles
    loops += share * m
out = {
    "kernel": "k_path<false, false>",
    "what": __doc__.split("\n    python")[0],
    "blocks": rows,
    "executed_valu_per_wave_trip_model": round(total, 1),
    "wave_trips_per_ray": counts["wave_trips_per_ray"],
    "classes": classes,
    "avg_issue_cycles_per_inst_architectural": round(arch, 3),
    "avg_issue_cycles_per_inst_single_class_loops": round(loops, 3),
    "vmem_cycles_per_load_inst": old["vmem_cycles_per_load_inst"], "vmem_note": old["vmem_note"],
    "counter_note": "SQ_ACTIVE_INST_VALU = 1 per instruction (2 per transcendental) whatever its issue time (profiles/r3_valu_calibration.json): not a busy-cycle count",
}
if pmc:
    measured = pmc["valu_insts_per_ray"] / counts["wave_trips_per_ray"]
    out.update({"executed_valu_per_wave_trip_measured": round(measured, 1), "model_over_measured": round(total / measured, 4),
                "measured_from": "profiles/r6_pmc_bench.json: SQ_INSTS_VALU per ray %.2f / wave-trips per ray %.5f (the counting variant's trips; the product's own scheduling "
                                 "may differ by a fraction of a percent)" % (pmc["valu_insts_per_ray"], counts["wave_trips_per_ray"]),
                "pmc_source_hash": pmc.get("source_hash")})
print(json.dumps(out, indent=1))
