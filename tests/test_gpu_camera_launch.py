"""-m gpu: k_trace_camera — camera rays generated and primary hits stored INSIDE the traversal launch (primaryray.glsl:23-94 in one dispatch;
the re-tracing frames of a batch: pathtracer.glsl:51-71, 115-127).  A lane computes its ray from the queue position it reserved, so every way
the positions can reach the lanes — reservation size, bite, refill threshold, contiguous or dealt chunks, reservations that straddle a
256-position chunk — must leave exactly the oracle's image, primary-hit cache and counters; so must shards, image sizes that are not whole
32x32 blocks, and batches whose frames belong to several tmpLifetime groups."""
import contextlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from adypt_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance  # noqa: E402


@contextlib.contextmanager
def environment(**kv):
    """Tunables are read at adypt_create: set them around the creation of ONE instance."""
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _instance(cache, name, w, h, pt=None, seed=31, env=None, **kw):
    spec = scenes.make_scene(name, cache, width=w, height=h, pt=pt or {})
    with environment(**(env or {})):
        inst = api.Instance()
        assert inst.InitializeFromFile(spec.config_path, shift_seed=seed, **kw), api.InstanceConfig.last_error()
    return inst


TUNABLES = [
    {},                                                                    # the defaults: a whole 8x8 tile per wave
    {"ADYPT_REFILL_MIN_PRIMARY": 1, "ADYPT_BITE_PRIMARY": 1},               # a ray at a time
    {"ADYPT_REFILL_MIN_PRIMARY": 17, "ADYPT_BITE_PRIMARY": 23, "ADYPT_CHUNK": 48},   # nothing aligned to anything: refills straddle 256-position chunks
    {"ADYPT_REFILL_MIN_PRIMARY": 64, "ADYPT_BITE_PRIMARY": 4096, "ADYPT_CHUNK": 4096},  # reservations of many chunks
    {"ADYPT_CHUNK": 16, "ADYPT_ENDGAME": 0},                                # no workgroup pool at all
    {"ADYPT_GEN_DEAL": 0},                                                  # contiguous segments instead of dealt chunks
    {"ADYPT_LDS_STACK_DEPTH": 1},                                                # (the stack spills to HBM: the spill column is addressed at use)
]


@pytest.mark.parametrize("env", TUNABLES)
def test_primary_only_call_matches_oracle_however_the_positions_are_handed_out(env, scene_cache):
    inst = _instance(scene_cache, "sibenik", 200, 117, env=env)  # 7 x 4 blocks, the last column and row partly outside the image
    c, pt = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    pt.SetInstrumentation(counters=True)
    for vt in (0, 4):
        pt.m_viewer_type = vt
        pt.ResetStats()
        pt.Trace(False)
        rgba, hits, ost = O.primary_frame(osc, P, vt)
        assert np.array_equal(bits(pt.ReadResult()), bits(rgba[..., :3])), (env, vt)
        tri, uv = pt.ReadHits()
        assert np.array_equal(tri, hits["tri_id"]) and np.array_equal(bits(uv), bits(np.stack([hits["u"], hits["v"]], -1))), env
        st = pt.GetStats()
        assert st["rays"] == 200 * 117, (env, st["rays"])  # one ray per pixel of the image, none for the rest of the border blocks
        o = ost.as_dict()
        assert (st["nodes_visited"], st["tris_tested"], st["hits"]) == (o["nodes"], o["tris"], o["hits"]), env


@pytest.mark.parametrize("env", [{}, {"ADYPT_REFILL_MIN_PRIMARY": 17, "ADYPT_BITE_PRIMARY": 23, "ADYPT_CHUNK": 48}, {"ADYPT_GEN_DEAL": 0}])
def test_retracing_frames_of_a_batch_in_several_tmplifetime_groups(env, scene_cache, sobol_matrices):
    """13 frames with tmpLifetime 3: one camera launch traces frames 0, 3, 6, 9, 12 (sub-pixel bias per group) into five cache images."""
    ptc = {"tmpLifetime": 3, "maxBounce": 4, "subpixel": 3}
    inst = _instance(scene_cache, "tiny0", 104, 70, pt=ptc, env=dict(env, ADYPT_FRAMES_IN_FLIGHT=16))
    c, p = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    p.SetInstrumentation(counters=True)
    p.ResetStats()
    p.Trace(True, 13)
    state = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, 13).as_dict()
    assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), env
    tri, uv = p.ReadHits()  # image 1 after the batch: the primary hits of the LAST group (frame 12)
    assert np.array_equal(tri, state.cache_tri) and np.array_equal(bits(uv), bits(state.cache_uv)), env
    st = p.GetStats()
    assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"]), env


@pytest.mark.parametrize("nranks", [2, 5])
def test_primary_only_on_tile_shards(nranks, scene_cache):
    whole = _instance(scene_cache, "tiny0", 170, 100)
    whole.m_path_tracer.Trace(False)
    ref = whole.m_path_tracer.ReadResult()
    ref_tri, ref_uv = whole.m_path_tracer.ReadHits()
    img = np.zeros_like(ref)
    tri = np.full_like(ref_tri, -2)
    rays = 0
    for r in range(nranks):
        part = _instance(scene_cache, "tiny0", 170, 100, tile_rank=r, tile_nranks=nranks)
        p = part.m_path_tracer
        p.SetInstrumentation(counters=True)
        p.ResetStats()
        p.Trace(False)
        owned = np.zeros(ref.shape[:2], bool)
        for by in range((100 + 31) // 32):
            for bx in range((170 + 31) // 32):
                if (bx + by) % nranks == r:
                    owned[by * 32:(by + 1) * 32, bx * 32:(bx + 1) * 32] = True
        img[owned] = p.ReadResult()[owned]
        tri[owned] = p.ReadHits()[0][owned]
        assert p.GetStats()["rays"] == int(owned.sum())
        rays += int(owned.sum())
    assert rays == 170 * 100
    assert np.array_equal(bits(img), bits(ref)) and np.array_equal(tri, ref_tri)


def test_stack_overflow_of_a_camera_launch_reaches_the_host(scene_cache):
    from adypt_amd import _native as N
    inst = _instance(scene_cache, "sibenik", 96, 54, pt={"stackSize": 1})
    p = inst.m_path_tracer
    with pytest.raises(N.AdyptError) as e:
        p.Trace(False)
    assert e.value.code == N.E_STACK_OVERFLOW
    p.ResetStats()  # the report is sticky until the statistics are reset (as the device counter it mirrors)
    params = inst.m_config.pt_params(31)
    params.stack_size = 24
    p.SetConfig(params)
    p.Trace(False)
