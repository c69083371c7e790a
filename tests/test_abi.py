"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    syms = set()
    for h in ("adypt_hip.h", "adypt_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(adypt_[a-z0-9_]+)\s*\(", text))
    return syms


def test_every_declared_symbol_is_exported():
    from adypt_amd import _native as N
    lib = ctypes.CDLL(N.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 40
    for s in sorted(declared):
        assert hasattr(lib, s), "libadypt_hip.so does not export %s" % s
    assert declared == set(N.EXPORTS), "python binding and headers disagree: %s" % (declared ^ set(N.EXPORTS))


def test_abi_version_and_struct_sizes():
    from adypt_amd import _native as N
    assert N.lib.adypt_abi_version() == 4
    assert ctypes.sizeof(N.Hit) == 36 and ctypes.sizeof(N.PtParams) == 40 and ctypes.sizeof(N.BvhParams) == 12


def test_create_without_device_fails_loudly():
    """No CPU fallback: on a box without a GPU adypt_create must return ADYPT_E_NO_DEVICE."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    from adypt_amd import api, _native as N
    from tests.helpers import golden_scene
    _, idx, nodes, tris, mats, _ = golden_scene("tiny2")
    sc = api.Scene.FromArrays(tris, mats)
    b = api.WideBVH()
    b.nodes, b.tri_indices = nodes.view("u1").reshape(-1), idx
    hs = api.HipScene()
    hs.Initialize(sc, b)
    pt = api.HipPathTracer()
    with pytest.raises(N.AdyptError) as e:
        pt.Initialize(api.InstanceConfig().pt_params(), hs, 64, 36)
    assert e.value.code == N.E_NO_DEVICE


def test_invalid_bvh_is_rejected_before_touching_the_gpu():
    from adypt_amd import api, _native as N
    from tests.helpers import golden_scene
    _, idx, nodes, tris, mats, _ = golden_scene("tiny1")
    bad = nodes.copy()
    bad["child_base"][0] = 10_000_000  # child range far outside the node array
    sc = api.Scene.FromArrays(tris, mats)
    b = api.WideBVH()
    b.nodes, b.tri_indices = bad.view("u1").reshape(-1), idx
    hs = api.HipScene()
    hs.Initialize(sc, b)
    with pytest.raises(N.AdyptError) as e:
        api.HipPathTracer().Initialize(api.InstanceConfig().pt_params(), hs, 64, 36)
    assert e.value.code == N.E_INVALID and "out of bounds" in str(e.value)
    bad_idx = idx.copy()
    bad_idx[3] = len(tris) + 5
    b.nodes, b.tri_indices = nodes.view("u1").reshape(-1), bad_idx
    with pytest.raises(N.AdyptError) as e:
        api.HipPathTracer().Initialize(api.InstanceConfig().pt_params(), hs, 64, 36)
    assert e.value.code == N.E_INVALID


def test_test_hooks_need_the_magic_and_the_environment_is_read_in_one_place():
    """adypt_enable_test_hooks refuses anything but ADYPT_TEST_HOOKS_MAGIC, and csrc/device looks at the environment in ONE function (read_tunables):
    a hook variable cannot reach the library by any other road."""
    from adypt_amd import _native as N
    assert N.lib.adypt_enable_test_hooks(0) == N.E_INVALID and N.lib.adypt_enable_test_hooks(0x7465737468303030) == N.E_INVALID
    magic = int(re.search(r"#define ADYPT_TEST_HOOKS_MAGIC (0x[0-9a-f]+)ull", open(os.path.join(ROOT, "include", "adypt_hip.h")).read()).group(1), 16)
    from adypt_amd import api
    assert api.TEST_HOOKS_MAGIC == magic
    dev = os.path.join(ROOT, "adypt_amd", "csrc", "device")
    uses = []
    for f in sorted(os.listdir(dev)):
        text = open(os.path.join(dev, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        uses += [(f, m.start()) for m in re.finditer(r"\bgetenv\s*\(", text)]
    assert uses and all(f == "tracer.hip" for f, _ in uses)
    src = re.sub(r"//[^\n]*", "", open(os.path.join(dev, "tracer.hip")).read())
    a, b = src.index("Tunables read_tunables()"), src.index("CtxInfo ctx_info(adypt_ctx *c)")
    assert all(a < pos < b for _, pos in uses), "getenv outside read_tunables()"
