"""(Round 4's tool; for k_path it is superseded by tools/trip_budget.py + tools/path_block_counts.py + tools/valu_issue_model.py, which weight the mix by
EXECUTED instructions.  Kept for k_trace and because profiles/r6_valu_issue_model.json takes the vector-memory cost per load from its round-4 output.)
Static instruction mix of the dominant kernel's persistent loop, by vector-ALU issue class (profiles/r3_valu_calibration.json).
Compiles device/tracer.hip with the Makefile's flags, takes k_path<false> (default) or k_trace<false,false> (argv[1] = "k_trace"), and counts
the VALU instructions between the persistent loop's header and its last back edge (blocks before / after it run once per wave).  Blocks that
do not run every trip are weighted by how often a trip executes them:
  k_trace  the refill block (ray setup: the block with the IEEE division sequences): 0.28 (profiles/r1_wave_profile_k_trace.json);
  k_path   the blocks between the `; ADYPT_MARK` comments of path.hpp — ray setup, exchange, shading round — by the counts of the in-kernel
           profile (profiles/r4_k_path_wave_profile.json: exchanges / trips, rounds / trips; a setup block runs with an exchange).
Instructions not in a measured class count as 4 cycles.
    python tools/instruction_mix.py [k_path|k_trace] > profiles/r4_<kernel>_instruction_mix.json"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
which = sys.argv[1] if len(sys.argv) > 1 else "k_path"
flags = re.search(r"HIPFLAGS\s*:=\s*(.*?)\n\n", open(os.path.join(CSRC, "Makefile")).read(), re.S).group(1).replace("\\\n", " ").replace("$(ARCH)", "gfx950").split()
flags = [f for f in flags if f not in ("-fPIC",)]
with tempfile.NamedTemporaryFile(suffix=".s") as t:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DADYPT_BUILD", "--cuda-device-only", "-S", os.path.join(CSRC, "device/tracer.hip"), "-o", t.name], stderr=subprocess.DEVNULL)
    text = open(t.name).read()
name = {"k_trace": "_ZN5adypt7k_traceILb0ELb0EEEvNS_9TraceArgsE", "k_path": "_ZN5adypt6k_pathILb0ELb0EEEvNS_12PathKernArgsE"}[which]
body = text[text.index("\n" + name + ":"):text.index(".amdhsa_kernel " + name)].splitlines()

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_ashrrev_i32", "v_mov_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_add_i32", "v_sub_i32"}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def klass(m):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if m.endswith("_dpp"):
        return "normal"  # DPP moves: counted at the normal rate (round 2)
    if base.startswith(TRANS):
        return "trans"
    if base.startswith("v_pk_") or base.endswith("_f64") or base.endswith("_b64") or base.endswith("_u64") or "u64" in base:
        return "packed64"
    return "full" if base in FULL else "normal"


# basic blocks: label lines end with ':'; find back edges (branch to an earlier label) -> the outermost loop = earliest target .. last back edge
labels, lines, marks = {}, [], {}
for ln in body:
    s = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        labels[s.split(":")[0]] = len(lines)
    m = re.match(r"^; ADYPT_MARK (\w+)", s)
    if m:
        marks[m.group(1)] = len(lines)
    lines.append(s)
back = []
for i, s in enumerate(lines):
    m = re.match(r"^s_cbranch\w*\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", s)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            back.append((labels[tgt], i))
lo = min(b[0] for b in back); hi = max(b[1] for b in back)
if which == "k_trace":
    # the refill block: the contiguous region of the loop around the v_div_scale / v_div_fmas sequences of the ray setup
    div = [i for i in range(lo, hi) if lines[i].startswith(("v_div_scale", "v_div_fmas", "v_div_fixup"))]
    clusters, cur = [], [div[0]]   # Woop's division is in the triangle test (one sequence); the ray setup has three in a row: take the largest cluster
    for i in div[1:]:
        if i - cur[-1] < 120: cur.append(i)
        else: clusters.append(cur); cur = [i]
    clusters.append(cur)
    refill = max(clusters, key=len)
    r_lo = max(v for v in labels.values() if v <= refill[0]); r_hi = min([v for v in labels.values() if v > refill[-1]] + [hi])
    regions = [("refill (ray setup + result writes)", r_lo, r_hi, 0.28)]
    weights_from = "profiles/r1_wave_profile_k_trace.json"
else:
    prof = json.load(open(os.path.join(ROOT, "profiles", "r4_k_path_wave_profile.json")))
    w_ex, w_sh = prof["exchanges"] / prof["trips"], prof["shading_rounds"] / prof["trips"]
    regions = [("ray setup", marks["setup_begin"], marks["setup_end"], w_ex), ("exchange (deposit / take, lists under the workgroup lock)", marks["exchange_begin"], marks["shade_begin"], w_ex),
               ("shading round (FetchInfo, illum switch, replacement paths, publish)", marks["shade_begin"], marks["shade_end"], w_sh), ("exchange, end", marks["shade_end"], marks["exchange_end"], w_ex)]
    weights_from = "profiles/r4_k_path_wave_profile.json (exchanges / trips = %.3f, shading rounds / trips = %.4f)" % (w_ex, w_sh)
counts = {"full": 0.0, "normal": 0.0, "packed64": 0.0, "trans": 0.0}
per_region = {r[0]: 0 for r in regions}
n_loop = 0
for i in range(lo, hi + 1):
    m = lines[i].split()[0] if lines[i] else ""
    if not m.startswith("v_") or m.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        continue
    w = 1.0
    for rname, a, b, rw in regions:
        if a <= i < b:
            w = rw
            per_region[rname] += 1
            break
    else:
        n_loop += 1
    counts[klass(m)] += w
tot = sum(counts.values())
mix = {"normal_rate_4_cycles": round(counts["normal"] / tot, 3), "full_rate_2_cycles": round(counts["full"] / tot, 3),
       "packed_or_64bit_4_cycles": round(counts["packed64"] / tot, 3), "transcendental_8_cycles": round(counts["trans"] / tot, 3)}
old = json.load(open(os.path.join(ROOT, "profiles", "r2_k_trace_instruction_mix.json")))
print(json.dumps({"kernel": {"k_trace": "k_trace<false, false>", "k_path": "k_path<false, false>"}[which], "what": __doc__.split("\n    python")[0], "flags": " ".join(flags),
                  "valu_instructions_every_trip": n_loop, "valu_instructions_in_weighted_blocks": per_region, "block_weights": {r[0]: round(r[3], 4) for r in regions}, "weights_from": weights_from,
                  "weighted_valu_instructions_per_trip": round(tot, 1),
                  "mix": mix, "avg_issue_cycles_per_inst": round(4 * mix["normal_rate_4_cycles"] + 2 * mix["full_rate_2_cycles"] + 4 * mix["packed_or_64bit_4_cycles"] + 8 * mix["transcendental_8_cycles"], 3),
                  "vmem_cycles_per_load_inst": old["vmem_cycles_per_load_inst"], "vmem_note": old["vmem_note"]}, indent=1))
