"""Build profiles/r4_valu_issue_model.json (read by bench.py) from the counter calibration (tools/valu_calibrate.sh ->
profiles/r3_valu_calibration.json: TRUE shader cycles per wave-instruction per SIMD of one-instruction kernels) and the instruction
mix of the dominant kernel's persistent loop (profiles/r4_k_path_instruction_mix.json, tools/instruction_mix.py).
    python tools/valu_issue_model.py > profiles/r4_valu_issue_model.json"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = {r["kernel"]: r for r in json.load(open(ROOT + "/profiles/r3_valu_calibration.json"))["rows"]}
mix = json.load(open(ROOT + "/profiles/r4_k_path_instruction_mix.json"))
CLASSES = [
    ("normal rate (v_cvt_f32_ubyte, min/max/min3/max3, v_cmp, v_cndmask, bit-field, shifts, v_perm, ...)", "normal_rate_4_cycles", 4,
     ["k_cvt_ubyte", "k_max_f32", "k_min_f32", "k_max3_f32", "k_med3", "k_cmp", "k_cmp_sgpr", "k_cnd_sgpr", "k_bfe", "k_lshl", "k_lshl_or", "k_perm",
      "k_and_or", "k_bfi", "k_lshl_add", "k_or3", "k_mul_lo", "k_div_fixup"]),
    ("full rate (fp32 add/mul/fma, and/or/xor, integer add, arithmetic shift right, moves)", "full_rate_2_cycles", 2,
     ["k_add_f32", "k_fma_f32", "k_mul_f32", "k_sub_f32", "k_and_b32", "k_or_b32", "k_xor_b32", "k_add_u32", "k_ashr", "k_mov"]),
    ("packed fp32 (v_pk_fma_f32) / 64-bit", "packed_or_64bit_4_cycles", 4, ["k_pk_fma", "k_mad_u64"]),
    ("transcendental (v_rcp_f32, v_sqrt_f32)", "transcendental_8_cycles", 8, ["k_rcp", "k_sqrt"]),
]
classes, ideal, measured = [], 0.0, 0.0
for name, key, architectural, kernels in CLASSES:
    vals = [cal[k]["issue_cycles_per_inst_per_simd"] for k in kernels]
    m = sum(vals) / len(vals)
    share = mix["mix"][key]
    classes.append({"class": name, "share": share, "cycles_architectural": architectural, "cycles_single_class_loop": round(m, 3),
                    "single_class_loops": {k: cal[k]["issue_cycles_per_inst_per_simd"] for k in kernels}})
    ideal += share * architectural
    measured += share * m
print(json.dumps({
    "kernel": mix["kernel"],
    "what": "shader cycles a wave-instruction holds its SIMD's vector-ALU issue, by class.  cycles_single_class_loop: profiles/r3_valu_calibration.json "
            "(GRBM_GUI_ACTIVE / 8 of kernels that issue ONE kind of instruction from 5 waves per SIMD: true cycles, no assumed clock); "
            "cycles_architectural: the 2 / 4 / 4 / 8 those loops approach.  share: the kernel's persistent loop (profiles/r4_k_path_instruction_mix.json; "
            "tools/instruction_mix.py: static count over the persistent loop, the blocks that do not run every trip weighted by the in-kernel profile).",
    "counter_note": "the same calibration shows SQ_ACTIVE_INST_VALU = 1 per instruction (2 per transcendental) whatever its issue time: "
                    "x 4 it is NOT a busy-cycle count (a loop of full-rate instructions reads 1.63 'busy'), so bench.py no longer builds its roof on it",
    "classes": classes,
    "avg_issue_cycles_per_inst_architectural": round(ideal, 3),
    "avg_issue_cycles_per_inst_single_class_loops": round(measured, 3),
    "vmem_cycles_per_load_inst": mix["vmem_cycles_per_load_inst"], "vmem_note": mix["vmem_note"]}, indent=1))
