"""Build check: lists the VGPR count of every gfx950 kernel of the library and fails when one lands exactly on an allocation-granule
boundary (a multiple of 8) — see the append_slot note in adypt_amd/csrc/device/shade.hpp.  python tools/check_vgpr.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
FLAGS = "-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -DADYPT_BUILD --cuda-device-only -S".split()
bad = []
for src in ("device/tracer.hip", "device/multi.hip"):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", t.name], stderr=subprocess.DEVNULL)
        text = open(t.name).read()
    for name, n in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        n = int(n)
        flag = "  <-- on a granule boundary" if n % 8 == 0 else ""
        print("%4d  %s%s" % (n, name, flag))
        if n % 8 == 0:
            bad.append(name)
sys.exit(1 if bad else 0)
