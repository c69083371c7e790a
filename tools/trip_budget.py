"""Static vector-instruction budget of k_path<false>'s persistent loop, per section and per skippable block, by opcode and issue class
(VERDICT r4 task 3).  No GPU needed: the device sources are copied, marked by adypt_amd/csrc/measure/k_path_blocks.py (assembly comments only),
compiled to gfx950 assembly with the Makefile's flags, and the VALU instructions between the marks are counted.
  sections of the trip (traverse_trip.inc): A choose / pop / push, B loads + triangle hand-over, C Woop test + hit update, D slab test, E finished?
  blocks: the parts behind a wave-level branch (skipped when no lane needs them), nested as in the source; `trip/X` = what every trip issues in section X.
The product library has no marks inside the trip; its own totals are printed beside the marked build's (they differ by a few instructions: the marks
are scheduling barriers).  Dynamic weights (how often a block runs) come from tools/path_block_counts.py on the GPU box.
    python tools/trip_budget.py > profiles/r6_trip_budget.json"""
import collections, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
FN = "_ZN5adypt6k_pathILb0ELb0EEEvNS_12PathKernArgsE"
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


def marked_build(flags):
    with tempfile.TemporaryDirectory() as t:
        # (a sibling directory two levels below: the sources include ../../../include/adypt_hip.h)
        dev = os.path.join(t, "a", "b", "device")
        os.makedirs(os.path.dirname(dev))
        shutil.copytree(os.path.join(CSRC, "device"), dev)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(t, "include"))
        subprocess.check_call([sys.executable, os.path.join(CSRC, "measure", "k_path_blocks.py"), dev], env=dict(os.environ, ADYPT_BLOCKS_COUNT="0"))
        out = os.path.join(t, "marked.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DADYPT_BUILD", "--cuda-device-only", "-S", os.path.join(dev, "tracer.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return body_of(open(out).read())


def main():
    flags = hipflags()
    marked = marked_build(flags)
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    product = body_of(open(os.path.join(CSRC, "build", "tracer.s")).read())

    first = {}
    for i, l in enumerate(marked):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            first.setdefault(m.group(1), i)
    lo, hi = loop_bounds(marked, first["exchange_end"])
    # Every vector instruction of the function goes to the innermost open block at its place in the assembly (marks nest like the source's blocks; a block
    # the compiler moved out of line — the unlikely side of a branch — still carries its own pair of marks).  Outside every block: the loop's own control
    # when inside the persistent loop, prologue / epilogue otherwise.
    stack, owner_lines, parents, instances = [], collections.defaultdict(list), {}, collections.Counter()
    sec = None
    for i, l in enumerate(marked):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            name = m.group(1)
            if name.startswith("sec_"):
                sec = name[4:]
            elif name.endswith("_begin"):
                b = name[:-6]
                parents.setdefault(b, stack[-1] if stack else None)
                instances[b] += 1
                stack.append(b)
                if b == "trip":
                    sec = "A"
            elif name.endswith("_end"):
                b = name[:-4]
                if b in stack:
                    while stack and stack.pop() != b:
                        pass
                if b == "trip":
                    sec = None
            continue
        if not is_valu(l):
            continue
        if stack:
            owner = stack[-1] if stack[-1] != "trip" else "trip/" + (sec or "A")
        else:
            owner = "loop control" if lo <= i <= hi else "outside the loop"
        owner_lines[owner].append(l)
    blocks = {}
    for owner, ls in owner_lines.items():
        c = count(ls)
        base = owner.split("/")[0]
        c["parent"] = "trip" if owner.startswith("trip/") else parents.get(base)
        c["instances"] = instances.get(base, 1)  # > 1: an inlined / unrolled block; its one counter counts the entries of all of them
        blocks[owner] = c
    trip_total = sum(c["valu"] for o, c in blocks.items() if o.startswith("trip/") or o[:2] in ("A_", "B_", "C_", "D_", "E_"))
    pm = {}
    for i, l in enumerate(product):
        m = re.search(r"ADYPT_MARK (\w+)", l)
        if m:
            pm.setdefault(m.group(1), i)
    plo, phi = loop_bounds(product, pm["exchange_end"])
    print(json.dumps({
        "kernel": "k_path<false, false>", "what": __doc__.split("\n    python")[0], "flags": " ".join(flags),
        "blocks": dict(sorted(blocks.items())),
        "trip_valu_static_marked_build": trip_total,
        "trip_valu_static_product_build": count(product[pm["exchange_end"]:phi + 1])["valu"],
        "note": "blocks[name].valu = static count of vector-ALU instructions whose innermost enclosing block is `name` (summed over its `instances`) (v_readlane / v_writelane / v_readfirstlane "
                "excluded); 'trip/A' .. 'trip/E' = what every trip issues, by section of traverse_trip.inc (A choose / pop / push, B loads + triangle hand-over, "
                "C Woop test + hit update, D slab test, E finished?); parent = the block it sits in.  by_class: full = 2-cycle fp32 / logic / move, normal = 4-cycle, "
                "packed64 = v_pk_* and 64-bit (4), trans = 8 (profiles/r3_valu_calibration.json).  The product build's trip (no marks inside) is counted from the "
                "exchange's end to the loop's last back edge, rare out-of-line blocks included."}, indent=1))


if __name__ == "__main__":
    main()
