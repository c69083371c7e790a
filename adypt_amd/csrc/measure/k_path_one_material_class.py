"""Measurement variant (NOT product code; WRONG images): every material answers as plain diffuse (illum 1) — the glossy, mirror and dielectric branches of
Render()'s illum switch drop out of the shading code.  An upper bound on what sorting a workgroup's to-shade paths by material class could save: a
shading round of 64 mixed paths executes every branch some lane needs.
    tools/build_variant.sh oneclass --transform adypt_amd/csrc/measure/k_path_one_material_class.py"""
import sys
p = sys.argv[1] + "/shade.hpp"
s = open(p).read()
old = "\tconst int illum0 = si.illum0;\n\tconst float shininess = si.shininess, ior = si.ior;\n"
assert s.count(old) == 1
s = s.replace(old, "\tconst int illum0 = 1; (void)si.illum0;\n\tconst float shininess = si.shininess, ior = si.ior;\n")
open(p, "w").write(s)
