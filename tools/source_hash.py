"""SHA-256 over the device sources of the library (adypt_amd/csrc/device/*.hpp, *.hip, in name order) and the Makefile that holds their
compiler flags: written into every profiles/*_pmc_*.json
by tools/pmc_profile.py and compared by bench.py, so that per-ray counter figures of an older kernel are never applied to a newer one."""
import glob, hashlib, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "adypt_amd", "csrc", "device", "*.h*"))) + [os.path.join(ROOT, "adypt_amd", "csrc", "Makefile")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(device_source_hash())
