"""BASELINE.json's large configurations on the GPU (SURVEY.md §8c "maximum sizes"): the ~1.2 M-triangle stand-in at
4096x4096 (config 5: 'salle-de-bain' is not redistributable, `scenes.make_scene("salle")` is the same generator as
the bench scene at higher tessellation) against the oracle, plus the size-independent properties: determinism, tile
shards reassembling, frames-in-flight invariance, ray conservation."""
import numpy as np
import pytest

from adypt_amd import distributed as D
from oracle import oracle_py as O
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance, random_rays
from tests.test_gpu_parity import make_instance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def salle(scene_cache):
    return make_instance(scene_cache, "salle", 4096, 4096, seed=777)


def test_salle_random_rays_bit_exact(salle):
    osc = oracle_scene_from_instance(salle)
    rays = random_rays(salle.scene.triangles, 400000, 3)
    g = salle.m_path_tracer.TraceRays(rays, with_stats=True)
    o = O.trace(osc, rays, salle.m_config.c.stack_size)
    assert g.tobytes() == o.tobytes()
    assert (g["tri_id"] >= 0).mean() > 0.5  # the rays really exercise the hierarchy
    ga = salle.m_path_tracer.TraceRays(rays, with_stats=True, any_hit=True)
    assert ga.tobytes() == O.trace(osc, rays, salle.m_config.c.stack_size, any_hit=True).tobytes()


def test_salle_4096_frame_bit_exact(salle, sobol_matrices):
    c, pt = salle.m_config.c, salle.m_path_tracer
    osc, P = oracle_scene_from_instance(salle), oracle_params_from_config(c)
    pt.Reset()
    pt.ResetStats()
    pt.Trace(True, 2)
    a = pt.ReadResult()
    st = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(777, c.width, c.height), sobol_matrices, st, 2).as_dict()
    assert np.array_equal(bits(a), bits(st.accum[..., :3]))
    gst = pt.GetStats()
    assert gst["rays"] == ost["rays"] and gst["stack_overflows"] == 0
    assert np.isfinite(a).all() and a.max() <= c.clamp
    # determinism and frames-in-flight invariance at this size
    auto = pt.GetFramesInFlight()
    pt.SetFramesInFlight(1 if auto != 1 else 2)
    pt.Reset()
    pt.Trace(True, 2)
    assert np.array_equal(bits(pt.ReadResult()), bits(a))
    pt.SetFramesInFlight(auto)


def test_salle_4096_shards_reassemble(salle, scene_cache):
    pt = salle.m_path_tracer
    pt.Reset()
    pt.ResetStats()
    pt.Trace(True, 1)
    ref, rays_full = pt.ReadResult(), pt.GetStats()["rays"]
    w = h = 4096
    out, rays = np.zeros_like(ref), 0
    for r in range(2):
        part = make_instance(scene_cache, "salle", w, h, seed=777, rank=r, world=2)
        part.m_path_tracer.Trace(True, 1)
        mask = D.owner_mask(w, h, r, 2).astype(bool)
        out[mask] = part.m_path_tracer.ReadResult()[mask]
        rays += part.m_path_tracer.GetStats()["rays"]
        part.m_path_tracer.destroy()
    assert np.array_equal(bits(out), bits(ref)) and rays == rays_full
