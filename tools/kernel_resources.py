"""Developer tool: per-kernel register / scratch / LDS use of the device code, read from the gfx950 assembly (no GPU needed).
    python tools/kernel_resources.py [extra hipcc flags ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adypt_amd", "csrc")
FLAGS = "-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -mllvm -disable-machine-licm -DADYPT_BUILD --cuda-device-only -S".split()


def kernel_resources(src="device/tracer.hip", extra=(), keep=None):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        out = keep or t.name
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + [os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
    res = {}
    for block in text.split("  - .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\.%s:\s+(\S+)" % name, block)
            return m.group(1) if m else None
        res[field("name")] = {k: int(field(k) or 0) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return res


if __name__ == "__main__":
    for src in ("device/tracer.hip", "device/multi.hip"):
        for name, r in sorted(kernel_resources(src, sys.argv[1:]).items()):
            print("%-70s vgpr %3d sgpr %3d spill v%d s%d scratch %d lds %d" % (name[:70], r["vgpr_count"], r["sgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"],
                                                                                 r["private_segment_fixed_size"], r["group_segment_fixed_size"]))
