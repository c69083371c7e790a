"""Generates tests/golden/obj_syntax_cases.json (build container only: needs oracle/_ref/adypt_ref = the reference's own Scene.cpp +
tinyobjloader compiled by oracle/Makefile): OBJ / MTL texts that exercise the corners of the file format a user's assets may contain, and
what the REFERENCE's loader makes of them — its 100-byte triangles and 64-byte GPU materials, hex encoded, or "rejected".  Fixtures are
data: input text and the reference's output."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

CUBE_V = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\n"
MTL = "newmtl a\nKd 0.1 0.2 0.3\nKs 0.4 0.5 0.6\nKe 1 2 3\nNs 40\nNi 1.3\nillum 2\nnewmtl b\nKd 0.9 0.8 0.7\nillum 7\nNi 1.5\n"

CASES = {
    "plain_triangles": ("mtllib m.mtl\n" + CUBE_V + "usemtl a\nf 1 2 3\nf 1 3 4\nusemtl b\nf 5 6 7\n", MTL),
    "negative_indices": ("mtllib m.mtl\n" + CUBE_V + "vn 0 0 1\nvt 0.25 0.75\nusemtl a\nf -8 -7 -6\nf -4/-1/-1 -3/-1/-1 -2/-1/-1\n", MTL),
    "index_forms": ("mtllib m.mtl\n" + CUBE_V + "vt 0 0\nvt 1 0\nvt 1 1\nvn 0 0 -1\nvn 1 0 0\nusemtl a\nf 1/1 2/2 3/3\nf 1//1 3//1 4//2\nf 5/1/2 6/2/2 7/3/2\nf 5 7 8\n", MTL),
    "polygons_fan": ("mtllib m.mtl\n" + CUBE_V + "v 0.5 1.5 0\nv 2 0.5 0\nusemtl b\nf 1 2 3 4\nf 1 2 10 3 9 4\nf 5 6 7 8 4\n", MTL),
    "concave_polygon": ("mtllib m.mtl\nv 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 0.5 0\nv 0 2 0\nusemtl a\nf 1 2 3 4 5\n", MTL),
    "mixed_normals_last_vertex_rule": ("mtllib m.mtl\n" + CUBE_V + "vn 0 1 0\nusemtl a\nf 1//1 2//1 3\nf 1 2 3//1\nf 5 6 7\n", MTL),
    "crlf_tabs_comments": ("# head\r\nmtllib m.mtl\r\n" + CUBE_V.replace("\n", "\r\n").replace(" ", "\t", 3) + "\r\n# mid\r\n  usemtl a  \r\n\tf 1 2 3   \r\n\r\nf  1   3\t4\r\n#tail", MTL.replace("\n", "\r\n")),
    "groups_objects_smoothing": ("mtllib m.mtl\no first\n" + CUBE_V + "g grp1 grp2\ns 1\nusemtl a\nf 1 2 3\ns off\ng\no second\nusemtl b\nf 5 6 7\nf 6 7 8\n", MTL),
    "no_usemtl_and_unknown_material": ("mtllib m.mtl\n" + CUBE_V + "f 1 2 3\nusemtl nosuch\nf 1 3 4\nusemtl b\nf 5 6 7\n", MTL),
    "missing_mtllib_file": ("mtllib nosuch.mtl\n" + CUBE_V + "usemtl a\nf 1 2 3\n", None),
    "no_mtllib_at_all": (CUBE_V + "f 1 2 3\nf 5 6 7\n", None),
    "float_spellings": ("mtllib m.mtl\nv 1e-3 +2.5 -.5\nv 5. 0.25E+1 -1.5e-2\nv 0 1 0\nv 1.0000001 2.0000002 3.0000003\nusemtl a\nf 1 2 3\nf 2 3 4\n", MTL),
    "vertex_extras": ("mtllib m.mtl\nv 0 0 0 1.0\nv 1 0 0 0.5 0.5 0.5\nv 0 1 0\nvt 0.1 0.2 0.3\nvt 0.4 0.5\nvt 0.6\nusemtl a\nf 1/1 2/2 3/3\n", MTL),
    "mtl_corners": ("mtllib m.mtl\n" + CUBE_V + "usemtl x\nf 1 2 3\nusemtl y\nf 1 3 4\nusemtl z\nf 5 6 7\nusemtl x\nf 6 7 8\n",
                    "# c\nnewmtl x\nKa 1 1 1\nKd 0.5\nKs 0.1 0.2\nTf 1 1 1\nd 0.5\nTr 0.2\nillum 5\nNs 12.5\nsharpness 3\nunknownkey 1 2 3\n\nnewmtl y\nillum 10\nKe 0 0 0\nNi 0\n"
                    "newmtl z\nKd 1 0 0\nnewmtl z\nKd 0 1 0\n"),
    "degenerate_faces": ("mtllib m.mtl\n" + CUBE_V + "usemtl a\nf 1 2\nf 1\nf 1 2 3\nf 4 4 4\n", MTL),
    "many_materials_order": ("mtllib m.mtl\n" + CUBE_V + "usemtl b\nf 1 2 3\nusemtl a\nf 1 3 4\nusemtl b\nf 5 6 7\n", MTL),
    "bad_numbers": ("mtllib m.mtl\nv abc 1 2\nv 1e 2 3\nv 1,5 2 3\nv 0 1 0 \nv 1 0\nv 7\nusemtl a\nf 1 2 3\nf 2 3 4\nf 4 5 6\n", MTL),
    "odd_face_tokens": ("mtllib m.mtl\n" + CUBE_V + "vt 0 0\nvn 0 0 1\nusemtl a\nf 1/ 2/ 3/\nf 1/1/ 2/1/ 3/1/\nf 4/1/1/9 5/1/1/9 6/1/1/9\nf 1 2 3 #c\n", MTL),
    "material_names_with_spaces": ("mtllib m.mtl\n" + CUBE_V + "usemtl my mat\nf 1 2 3\nusemtl b \nf 1 3 4\nusemtl  b\nf 5 6 7\n", "newmtl my mat\nKd 1 0 0\nnewmtl b\nKd 0 1 0\n"),
    "two_mtllibs_on_a_line": ("mtllib nosuch.mtl m.mtl\n" + CUBE_V + "usemtl a\nf 1 2 3\n", MTL),
    "mtllib_twice": ("mtllib m.mtl\n" + CUBE_V + "usemtl a\nf 1 2 3\nmtllib m.mtl\nusemtl b\nf 1 3 4\nusemtl a\nf 5 6 7\n", MTL),
    "uppercase_and_unknown_statements": ("mtllib m.mtl\n" + CUBE_V + "V 9 9 9\nvp 1 2 3\nl 1 2\np 1\ncurv 0 1 1 2\nusemtl a\nF 1 2 3\nf 1 2 3\n", MTL),
    "long_polygon": ("mtllib m.mtl\n" + "".join("v %d %d 0\n" % (i, (i * 7) % 5) for i in range(40)) + "usemtl a\nf " + " ".join(str(i + 1) for i in range(40)) + "\n", MTL),
    "unnormalised_and_zero_normals": ("mtllib m.mtl\n" + CUBE_V + "vn 0 0 5\nvn 0 0 0\nusemtl a\nf 1//1 2//1 3//1\nf 1//2 3//2 4//2\n", MTL),
    "no_trailing_newline_face": ("mtllib m.mtl\n" + CUBE_V + "usemtl a\nf 1 2 3", MTL),
    "empty_file": ("", None),
    "only_vertices": (CUBE_V, None),
}


def main():
    assert os.path.exists(O.REF_BIN), "build oracle/_ref first (make -C oracle _ref)"
    out = {}
    for name, (obj, mtl) in CASES.items():
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "c.obj"), "w", newline="") as f:
                f.write(obj)
            if mtl is not None:
                with open(os.path.join(d, "m.mtl"), "w", newline="") as f:
                    f.write(mtl)
            r = subprocess.run([O.REF_BIN, "scene", os.path.join(d, "c.obj"), os.path.join(d, "t.bin"), os.path.join(d, "m.bin")],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
            case = {"obj": obj, "mtl": mtl}
            if r.returncode == 0:
                case["triangles"] = open(os.path.join(d, "t.bin"), "rb").read().hex()
                case["materials"] = open(os.path.join(d, "m.bin"), "rb").read().hex()
            else:
                case["rejected"] = True
                case["reference_exit"] = r.returncode  # negative = killed by a signal (the reference crashed)
            out[name] = case
            print(name, "rejected (%d)" % r.returncode if r.returncode else "%d triangles, %d materials" % (len(case["triangles"]) // 200, len(case["materials"]) // 128))
    with open(os.path.join(HERE, "obj_syntax_cases.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
