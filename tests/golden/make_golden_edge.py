"""Generates tests/golden/edge_cases.json (build container only: needs oracle/_ref/adypt_ref = the reference's own CPU
code compiled by oracle/Makefile): degenerate tiny scenes — 2..25 random triangles, collinear (zero-area) triangles,
coplanar strips, triangles collapsed to one point — and the `.bvh` bytes the REFERENCE's SBVHBuilder + WideBVHBuilder
write for them.  Fixtures are data: the OBJ text (input) and the reference's output, hex encoded.

The reference crashes (SIGSEGV in the collapse) on a ONE-triangle scene; that case is recorded with "bvh": null and the
product defines it (a root node with one leaf child, tests/test_host_golden.py)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402


def obj_text(n, mode, rs):
    lines = []
    for i in range(n):
        if mode == "rand":
            p = rs.uniform(-1, 1, (3, 3))
        elif mode == "degenerate":
            p = np.array([[i, 0, 0], [i + 1, 0, 0], [i + 2, 0, 0]], float)
        elif mode == "flat":
            p = np.array([[i, 0, 0], [i + 1, 0, 0], [i, 1, 0]], float)
        else:  # "point"
            p = np.zeros((3, 3))
        lines += ["v %r %r %r" % tuple(float(x) for x in v) for v in p]
    lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(n)]
    return "\n".join(lines) + "\n"


def main():
    assert os.path.exists(O.REF_BIN), "build oracle/_ref first (make -C oracle _ref)"
    rs = np.random.RandomState(5)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for mode, sizes in (("rand", (1, 2, 3, 4, 5, 9, 25)), ("degenerate", (2, 5, 9)), ("flat", (3, 9)), ("point", (2, 4))):
            for n in sizes:
                name = "%s_%d" % (mode, n)
                obj, bvh = os.path.join(d, name + ".obj"), os.path.join(d, name + ".bvh")
                txt = obj_text(n, mode, rs)
                open(obj, "w").write(txt)
                r = subprocess.run([O.REF_BIN, "build", obj, bvh, "48", "0.3", "1.0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                ok = r.returncode == 0 and os.path.exists(bvh)
                out[name] = {"obj": txt, "bvh": open(bvh, "rb").read().hex() if ok else None, "ref_returncode": r.returncode}
                print(name, "reference rc", r.returncode, "bytes", len(out[name]["bvh"]) // 2 if ok else None)
    json.dump(out, open(os.path.join(HERE, "edge_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
