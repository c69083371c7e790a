"""Build check: lists the VGPR count of every gfx950 kernel of the library and marks those that land exactly on an allocation-granule
boundary (a multiple of 8).  Fails only for the small kernels that call append_slot and carry ADYPT_VGPR_SLACK for that reason (DESIGN.md
§10); the traversal kernel is MEANT to sit on 80 (6 waves per SIMD).  python tools/check_vgpr.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
FLAGS = "-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -mllvm -disable-machine-licm -DADYPT_BUILD --cuda-device-only -S".split()
bad = []
for src in ("device/tracer.hip", "device/multi.hip"):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", t.name], stderr=subprocess.DEVNULL)
        text = open(t.name).read()
    for name, n in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        n = int(n)
        flag = "  <-- on a granule boundary" if n % 8 == 0 else ""
        print("%4d  %s%s" % (n, name, flag))
        if n % 8 == 0 and ("k_gen_primary" in name or "k_viewer" in name or "k_shadow_resolve" in name):
            bad.append(name)
sys.exit(1 if bad else 0)
