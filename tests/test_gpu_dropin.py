"""-m gpu: the drop-in end to end.  oracle/_ref/adypt_dropin (built in the build container by oracle/Makefile) is the
REFERENCE's own CPU half — InstanceConfig, Scene/tinyobj, SBVHBuilder, WideBVHBuilder compiled from the reference sources
— driving libadypt_hip.so through integration/HipPathTracer.hpp, one Trace(true) per frame like Instance::Update.  Its
image must equal, bit for bit, the one the product's own loader + builder + CLI (batched frames) renders from the same
.config; both EXR files are decoded by the reference's tinyexr (oracle/_ref/adypt_ref exrload)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "adypt_dropin")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "adypt_ref")
CLI = os.path.join(ROOT, "adypt_amd", "adypt_hip")


def _decode(exr, out):
    r = subprocess.run([REFBIN, "exrload", exr, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    w, h = map(int, r.stdout.split()[:2])
    return np.fromfile(out, dtype=np.float32).reshape(h, w, 4)


def _require_ref_binaries():
    """oracle/_ref/ is git-ignored but travels to the GPU box with the snapshot.  Where the reference sources exist the
    binaries can always be built (oracle/Makefile), so their absence is a failure there; elsewhere it is a visible skip."""
    if os.path.exists(DROPIN) and os.path.exists(REFBIN):
        return
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert os.path.exists(DROPIN) and os.path.exists(REFBIN), "oracle/_ref binaries missing although /root/reference exists: run `make -C oracle`"
        return
    pytest.skip("DROP-IN TEST NOT RUN: oracle/_ref/{adypt_dropin,adypt_ref} are absent and there are no reference sources here to build "
                "them from (they are built in the build container by __graft_entry__.build() and shipped with the gpurun snapshot)")


@pytest.mark.parametrize("name,w,h,spp,lookahead", [("tiny0", 96, 64, 5, 32), ("sibenik", 160, 90, 3, 32), ("tiny0", 96, 64, 7, 1), ("tiny0", 96, 64, 4, 0)])
def test_reference_cpu_half_plus_this_library_equals_the_product_pipeline(name, w, h, spp, lookahead, tmp_path):
    _require_ref_binaries()
    from adypt_amd import scenes
    spec = scenes.make_scene(name, str(tmp_path), width=w, height=h, pt={"maxBounce": 5, "stackSize": 24, "tmpLifetime": 2})
    ref_exr, our_exr = str(tmp_path / "dropin.exr"), str(tmp_path / "ours.exr")
    # (lookahead 32: whole passes traced ahead; 1: one frame per pass, the next one started ahead on a second stream; 0: strictly one frame per call)
    r = subprocess.run([DROPIN, spec.config_path, ref_exr, str(spp), "0", str(lookahead)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and ("[DROPIN]spp %d" % spp).encode() in r.stdout, (r.stdout + r.stderr).decode()[-2000:]
    r = subprocess.run([CLI, spec.config_path, "--spp", str(spp), "--out", our_exr, "--seed", "12345"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]
    a, b = _decode(ref_exr, str(tmp_path / "a.bin")), _decode(our_exr, str(tmp_path / "b.bin"))
    assert a.shape == b.shape == (h, w, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[..., :3].max() > 0.05  # not a black frame
