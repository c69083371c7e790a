"""BASELINE.json configs 3, 4 and 5 at their STATED length on the GPU, checked by the oracle on a sparse pixel mask:
  C3  sponza stand-in    1920x1080   64 spp
  C4  10 M triangles     1920x1080   16 spp
  C5  salle stand-in     4096x4096 1024 spp, progressive (several calls)
The oracle renders only the masked pixels (every ~N-th pixel, N prime so that the picks drift through every column, row, 32x32 block
and wave tile) through the SAME frame loop — Sobol point of every frame, per-pixel shift, sub-pixel offset and primary-hit cache of
every tmpLifetime group, fp32 running mean applied 64 / 16 / 1024 times — and the GPU image must equal it bit for bit there; on the
whole image: finite, within the clamp, exact ray conservation against a second run, determinism.  What the long runs add over the short
full-image comparisons of test_gpu_parity.py / test_gpu_large.py: the Sobol sequence far along (every direction-number row in use),
the sub-pixel index wrapped around its 8 x 8 grid at full image size, many batches of frames in flight back to back and, since round 3,
many pipelined sub-batch chains."""
import numpy as np
import pytest

from oracle import oracle_py as O
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance
from tests.test_gpu_parity import make_instance

pytestmark = pytest.mark.gpu


def sparse_mask(w, h, stride, phase=7):
    m = np.zeros(w * h, np.uint8)
    m[phase::stride] = 1
    return m.reshape(h, w)


def run_and_check(inst, seed, spp, stride, sobol_matrices, calls=(None,)):
    c, pt = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    mask = sparse_mask(c.width, c.height, stride)
    assert 3000 < int(mask.sum()) < 20000
    pt.Reset()
    pt.ResetStats()
    state = O.PathTracerState(c.width, c.height)
    shift = O.shift_bytes(seed, c.width, c.height)
    done = 0
    for n in [spp if k is None else k for k in calls]:
        pt.Trace(True, n)
        O.pt_frames(osc, P, shift, sobol_matrices, state, n, mask=mask)
        done += n
        img = pt.ReadResult()
        m = mask.astype(bool)
        assert pt.GetSPP() == done
        assert np.array_equal(bits(img[m]), bits(state.accum[..., :3][m])), "after %d spp: %d of %d masked pixels differ" % (
            done, int((bits(img[m]) != bits(state.accum[..., :3][m])).any(-1).sum()), int(m.sum()))
    assert done == spp
    st = pt.GetStats()
    assert st["stack_overflows"] == 0 and st["bad_materials"] == 0
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= c.clamp
    return img, st


def test_config3_sponza_1080p_64spp(scene_cache, sobol_matrices):
    inst = make_instance(scene_cache, "sponza", 1920, 1080, seed=12345)
    c = inst.m_config.c
    assert (c.max_bounce, c.tmp_lifetime, c.subpixel) == (8, 16, 8)
    img, st = run_and_check(inst, 12345, 64, 509, sobol_matrices)
    # the same 64 frames again, one frame per wavefront pass and without sub-batch pipelining: same image, same number of rays
    pt = inst.m_path_tracer
    pt.SetFramesInFlight(1)
    pt.SetPipeline(1)
    pt.Reset()
    pt.ResetStats()
    pt.Trace(True, 64)
    assert np.array_equal(bits(pt.ReadResult()), bits(img)) and pt.GetStats()["rays"] == st["rays"]
    pt.destroy()


def test_config4_sanmiguel_10M_1080p_16spp(scene_cache, sobol_matrices):
    inst = make_instance(scene_cache, "sanmiguel", 1920, 1080, seed=4242)
    assert inst.scene.n_tris > 9_000_000 and inst.m_config.c.max_bounce == 8
    run_and_check(inst, 4242, 16, 257, sobol_matrices)
    inst.m_path_tracer.destroy()


def test_config5_salle_4096_1024spp_progressive(scene_cache, sobol_matrices):
    inst = make_instance(scene_cache, "salle", 4096, 4096, seed=777)
    assert inst.m_config.c.max_bounce == 8
    # progressive: the image is read (and checked) after 1, 64, 256 and 1024 samples
    run_and_check(inst, 777, 1024, 4099, sobol_matrices, calls=(1, 63, 192, 768))
    inst.m_path_tracer.destroy()
