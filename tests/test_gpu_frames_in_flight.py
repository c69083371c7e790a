"""-m gpu: several frames in flight per wavefront pass must not change a single bit — frames are independent
samples, the running mean is applied in frame order, and a batch may span several tmpLifetime groups (the frames
that re-trace their primary rays, spp % tmpLifetime == 0, run a primary-only pass into the cache image of their
group; image 1 ends up holding the last group's hits, as it does frame by frame)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from adypt_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance  # noqa: E402


def _instance(cache, name, w, h, pt, seed=31):
    spec = scenes.make_scene(name, cache, width=w, height=h, pt=pt)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=seed), api.InstanceConfig.last_error()
    return inst


@pytest.mark.parametrize("name,w,h,life,spp", [("tiny0", 100, 75, 4, 11), ("tiny0", 96, 64, 16, 20), ("sibenik", 160, 90, 3, 7),
                                                   ("tiny0", 64, 48, 1, 9), ("tiny0", 72, 40, 2, 37)])
def test_frames_in_flight_is_bit_invariant(name, w, h, life, spp, scene_cache, sobol_matrices):
    pt_cfg = {"tmpLifetime": life, "maxBounce": 6, "subpixel": 3}
    ref_img, ref_rays = None, None
    for fif in (1, 2, 5, 8, 13, 32):
        inst = _instance(scene_cache, name, w, h, pt_cfg)
        p = inst.m_path_tracer
        p.SetFramesInFlight(fif)
        assert p.GetFramesInFlight() == fif
        p.SetInstrumentation(counters=True)
        p.ResetStats()
        p.Trace(True, spp - 3)
        p.Trace(True, 3)  # batches also continue correctly across calls
        img, st = p.ReadResult(), p.GetStats()
        assert p.GetSPP() == spp
        if ref_img is None:
            ref_img, ref_rays = img, st
            c = inst.m_config.c
            osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
            state = O.PathTracerState(c.width, c.height)
            ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, spp).as_dict()
            assert np.array_equal(bits(img), bits(state.accum[..., :3]))
            assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])
        else:
            assert np.array_equal(bits(img), bits(ref_img)), "frames in flight = %d changed the image" % fif
            for k in ("rays", "nodes_visited", "tris_tested", "hits", "shaded"):
                assert st[k] == ref_rays[k], k
        # the primary-hit cache image ends up identical too
        tri, uv = p.ReadHits()
        if fif == 1:
            ref_tri, ref_uv = tri, uv
        else:
            assert np.array_equal(tri, ref_tri) and np.array_equal(bits(uv), bits(ref_uv))
