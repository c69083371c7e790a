"""Build check (run by __graft_entry__.build()): the VGPR count of every gfx950 kernel of the library, and one rule for the kernels that
compact their survivors with append_slot (shade.hpp).

The rule: a kernel that contains append_slot must be allocated MORE THAN 16 VGPRs.  Round 3's queue corruption (whole waves claiming one
slot; DESIGN.md, the append_slot fault) was only ever seen in a build whose k_gen_primary came out at exactly 16 — the same instructions with
17+ registers allocated never failed, and the callers that sit on larger granule boundaries (k_shade 72, k_shade_first 80) have run billions of
claims under the slot-claim audit without one error.  Which kernels contain append_slot is read from the assembly, not from a list of names: its
signature is the pair of workgroup barriers followed by v_mbcnt_hi with a returning global atomic between them (k_path uses LDS lists and
is not matched).  python tools/check_vgpr.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
MIN_VGPRS = 17


def kernels(src):
    """The assembly comes from the Makefile's own `asm` target — the compiler, architecture and flags of the shipped objects (ROCM / HIPCC / ARCH
    overrides included), not a copy of the flag list kept here."""
    subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stdout=subprocess.DEVNULL)
    text = open(os.path.join(CSRC, "build", os.path.basename(src).replace(".hip", ".s"))).read()
    counts = dict((n, int(v)) for n, v in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text))
    out = []
    for name, n in counts.items():
        m = re.search(r"\n" + re.escape(name) + r":.*?\n\s+s_endpgm", text, re.S)
        body = m.group(0) if m else ""
        # append_slot: s_barrier ... global_atomic_add (returning: sc0) ... s_barrier ... v_mbcnt_hi
        uses = re.search(r"s_barrier.*?global_atomic_add\s+v\d+,.*?sc0.*?s_barrier.*?v_mbcnt_hi_u32_b32", body, re.S) is not None and "ds_cmpst" not in body
        out.append((name, n, uses))
    return out


def main():
    bad = []
    for src in ("device/tracer.hip", "device/multi.hip"):
        for name, n, uses in sorted(kernels(src)):
            flag = ""
            if uses:
                flag = "  append_slot: needs > 16" + ("" if n >= MIN_VGPRS else "  <-- VIOLATION")
                if n < MIN_VGPRS:
                    bad.append(name)
            print("%4d  %s%s" % (n, name, flag))
    if bad:
        print("check_vgpr: kernels with append_slot at <= 16 VGPRs: %s (add ADYPT_VGPR_SLACK, shade.hpp)" % ", ".join(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
