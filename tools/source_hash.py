"""SHA-256 over the CODE of the library's device sources (adypt_amd/csrc/device/*.hpp, *.hip, *.inc, in name order; comments removed, runs of
white space collapsed — a reworded comment does not make a counter profile stale, a changed token does) and the Makefile that holds their compiler
flags: written into every profiles/*_pmc_*.json by tools/pmc_profile.py and compared by bench.py, so that per-ray counter figures of an older
kernel are never applied to a newer one."""
import glob, hashlib, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text: str) -> str:
    """C / C++ comments out, string and character literals kept as they are."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" "); i = n if j < 0 else j + 2
        else:
            out.append(c); i += 1
    return re.sub(r"\s+", " ", "".join(out)).strip()


def device_source_hash() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "adypt_amd", "csrc", "device")
    files = sorted(f for pat in ("*.hpp", "*.hip", "*.inc") for f in glob.glob(os.path.join(d, pat)))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(strip_comments(open(f, "r", encoding="utf-8").read()).encode() + b"\0")
    h.update(open(os.path.join(ROOT, "adypt_amd", "csrc", "Makefile"), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(device_source_hash())
