"""integration/HipPathTracer.hpp — the binding a maintainer drops into Adypt's src/Tracer — must compile against the
reference's own headers (Scene, WideBVH, InstanceConfig, tinyobj, stb_image, tinyexr, glm; C++11 like the reference's
CMakeLists.txt).  Build container only: skipped where /root/reference does not exist (GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference sources not present")
def test_binding_header_compiles_against_reference_headers(tmp_path):
    tracer = tmp_path / "Tracer"
    tracer.mkdir()
    os.symlink(os.path.join(ROOT, "integration", "HipPathTracer.hpp"), tracer / "HipPathTracer.hpp")
    for name in ("Util", "BVH", "InstanceConfig.hpp"):
        os.symlink(os.path.join(REF, "src", name), tmp_path / name)
    (tmp_path / "check.cpp").write_text(
        '#include "Tracer/HipPathTracer.hpp"\n'
        "// every member is used once so that the bodies are instantiated and type-checked\n"
        "int use(const InstanceConfig::PT *cfg, const Scene &scene, const WideBVH &bvh)\n"
        "{\n"
        "    HipPathTracer pt;\n"
        "    if(!pt.Initialize(cfg, scene, bvh, 1920, 1080)) return 1;\n"
        "    pt.SetCamera(glm::mat4(1.0f), glm::mat4(1.0f), glm::vec3(0.0f));\n"
        "    pt.Trace(false); pt.Trace(true);\n"
        "    std::vector<uint8_t> screen; pt.ReadScreen(&screen);\n"
        '    pt.SaveResult("o.exr", true);\n'
        "    return pt.GetSPP() + (int)pt.m_viewer_type;\n"
        "}\n")
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-DGLM_FORCE_SWIZZLE", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(REF, "dep"), "-I" + str(tmp_path), str(tmp_path / "check.cpp")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
