"""Static vector-instruction budget of k_path<false>'s persistent loop, per section and per skippable block, by opcode and issue class
(VERDICT r4 task 3).  No GPU needed: the device sources are copied, marked by adypt_amd/csrc/measure/k_path_blocks.py (assembly comments only),
compiled to gfx950 assembly with the Makefile's flags, and the VALU instructions between the marks are counted.
  sections of the trip (traverse_trip.inc): A choose / pop / push, B loads + triangle hand-over, C Woop test + hit update, D slab test, E finished?
  blocks: the parts behind a wave-level branch (skipped when no lane needs them); `always` = what every trip issues.
The product library has no marks inside the trip; its own totals are printed beside the marked build's (they differ by a few instructions: the marks
are scheduling barriers).  Dynamic weights (how often a block runs) come from tools/path_block_counts.py on the GPU box.
    python tools/trip_budget.py > profiles/r5_trip_budget.json"""
import collections, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
FN = "_ZN5adypt6k_pathILb0EEEvNS_12PathKernArgsE"
FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_ashrrev_i32", "v_mov_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_add_i32", "v_sub_i32"}
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def klass(m):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if m.endswith("_dpp"):
        return "normal"
    if base.startswith(TRANS):
        return "trans"
    if base.startswith("v_pk_") or base.endswith(("_f64", "_b64", "_u64")) or "u64" in base:
        return "packed64"
    return "full" if base in FULL else "normal"


def is_valu(s):
    return s.startswith("v_") and not s.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))


def hipflags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    f = re.search(r"HIPFLAGS\s*:=\s*(.*?)\n\n", mk, re.S).group(1).replace("\\\n", " ").replace("$(ARCH)", "gfx950").split()
    return [x for x in f if x != "-fPIC"]


def body_of(asm_text):
    return [l.strip() for l in asm_text[asm_text.index("\n" + FN + ":"):asm_text.index(".amdhsa_kernel " + FN)].splitlines()]


def loop_bounds(lines, inside):
    """the persistent loop = the depth-1 loop that contains line `inside`: (header index, last back edge)"""
    best = None
    for i, l in enumerate(lines):
        if "Loop Header: Depth=1" in l and i < inside:
            hl = l.split(":")[0]
            br = [j for j, x in enumerate(lines) if re.search(r"s_c?branch\w*\s+" + re.escape(hl) + r"\b", x) and j > inside]
            if br:
                best = (i, br[-1])
    return best


def count(lines):
    ops, cls = collections.Counter(), collections.Counter()
    for s in lines:
        if is_valu(s):
            m = s.split()[0]
            ops[re.sub(r"_(e32|e64)$", "", m)] += 1
            cls[klass(m)] += 1
    return {"valu": sum(ops.values()), "by_class": dict(cls), "by_opcode": dict(sorted(ops.items(), key=lambda kv: -kv[1]))}


def marked_build(flags, block_set):
    with tempfile.TemporaryDirectory() as t:
        # (a sibling directory two levels below: the sources include ../../../include/adypt_hip.h)
        dev = os.path.join(t, "a", "b", "device")
        os.makedirs(os.path.dirname(dev))
        shutil.copytree(os.path.join(CSRC, "device"), dev)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(t, "include"))
        subprocess.check_call([sys.executable, os.path.join(CSRC, "measure", "k_path_blocks.py"), dev], env=dict(os.environ, ADYPT_BLOCKS_COUNT="0", ADYPT_BLOCKS_SET=block_set))
        out = os.path.join(t, "marked.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DADYPT_BUILD", "--cuda-device-only", "-S", os.path.join(dev, "tracer.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return body_of(open(out).read())


def shade_blocks(flags):
    """the blocks INSIDE a shading round (second marked build): what runs every round, and what only when some lane needs it"""
    lines = marked_build(flags, "shade")
    marks = {}
    for i, l in enumerate(lines):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            marks.setdefault(m.group(1), i)
    names = ["S_parked", "S_miss", "S_surface", "S_textured", "S_glossy", "S_diffuse", "S_mirror", "S_dielectric", "S_dead", "S_alive", "S_replace"]
    span = {b: (marks[b + "_begin"], marks[b + "_end"]) for b in names}
    inner = ["S_textured", "S_glossy", "S_diffuse", "S_mirror", "S_dielectric"]  # nested in S_surface
    per = {b: [] for b in names}
    every = []
    for i in range(marks["shade_begin"], marks["shade_end"]):
        owner = next((b for b in inner if span[b][0] <= i < span[b][1]), None) or next((b for b in names if b not in inner and span[b][0] <= i < span[b][1]), None)
        (per[owner] if owner else every).append(lines[i])
    out = {"every round": count(every)}
    out.update({b: count(per[b]) for b in names})
    out["valu_static"] = sum(v["valu"] for v in out.values())
    return out


def main():
    flags = hipflags()
    marked = marked_build(flags, "trip")
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    product = body_of(open(os.path.join(CSRC, "build", "tracer.s")).read())

    marks = {}
    for i, l in enumerate(marked):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            marks.setdefault(m.group(1), i)
    lo, hi = loop_bounds(marked, marks["exchange_end"])
    trip_lo, trip_hi = marks["trip_begin"], marks["trip_end"]
    blocks = ["A_pop", "A_choose", "A_push", "B_tri_load", "B_node_load", "C_woop", "D_slab", "E_flush"]
    span = {b: (marks[b + "_begin"], marks[b + "_end"]) for b in blocks}
    sec_at = sorted((marks["sec_" + s], s) for s in "ABCDE")

    def section_of(i):
        cur = "A"
        for pos, s in sec_at:
            if i >= pos:
                cur = s
        return cur

    in_block = lambda i: next((b for b, (a, e) in span.items() if a <= i < e), None)
    always = {s: [] for s in "ABCDE"}
    per_block = {b: [] for b in blocks}
    for i in range(trip_lo, trip_hi):
        b = in_block(i)
        (per_block[b] if b else always[section_of(i)]).append(marked[i])
    sections = {}
    for s in "ABCDE":
        bl = {b: count(per_block[b]) for b in blocks if b.startswith(s + "_")}
        sections[s] = {"always": count(always[s]), "blocks": bl, "valu_static": count(always[s])["valu"] + sum(v["valu"] for v in bl.values())}
    regions = {"ray setup": count(marked[marks["setup_begin"]:marks["setup_end"]]),
               "exchange (in front of a shading round)": count(marked[marks["exchange_begin"]:marks["shade_begin"]]),
               "shading round": count(marked[marks["shade_begin"]:marks["shade_end"]]),
               "exchange (after a shading round / without one)": count(marked[marks["shade_end"]:marks["exchange_end"]])}
    marked_regions = [(marks["setup_begin"], marks["setup_end"]), (marks["exchange_begin"], marks["exchange_end"]), (trip_lo, trip_hi)]
    rest = [marked[i] for i in range(lo, hi + 1) if not any(a <= i < e for a, e in marked_regions)]
    # the product build: the trip = from the exchange's end to the loop's last back edge (no marks inside)
    pm = {}
    for i, l in enumerate(product):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            pm.setdefault(m.group(1), i)
    plo, phi = loop_bounds(product, pm["exchange_end"])
    print(json.dumps({
        "kernel": "k_path<false>", "what": __doc__.split("\n    python")[0], "flags": " ".join(flags),
        "trip_sections": sections,
        "trip_valu_static_marked_build": sum(v["valu_static"] for v in sections.values()),
        "trip_valu_static_product_build": count(product[pm["exchange_end"]:phi + 1])["valu"],
        "loop_outside_trip": dict(regions, **{"loop control outside every mark (votes, the exchange's condition)": count(rest)}),
        "shading_round_blocks": shade_blocks(flags),
        "note": "valu = static count of vector-ALU instructions (v_readlane / v_writelane / v_readfirstlane excluded, as in SQ_INSTS_VALU's complement of scalar work they "
                "are few).  by_class: full = 2-cycle fp32 / logic / move, normal = 4-cycle, packed64 = v_pk_* and 64-bit (4), trans = 8 (profiles/r3_valu_calibration.json)."}, indent=1))


if __name__ == "__main__":
    main()
