"""-m gpu: k_path — every bounce after the first of a batch in ONE persistent launch (shaders/pathtracer.glsl:107-202 inside one dispatch,
src/Tracer/OglPathTracer.cpp:60) — must leave exactly what the launch-per-bounce pipeline leaves and what the oracle computes: image bits, the
primary-hit cache, ray / node / triangle / hit / shaded counts, the stack-overflow report.  The kernel's tunables (path slots are compile time;
LDS stack depth, shading-batch size, refill threshold, workgroups per CU) change scheduling only."""
import contextlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from adypt_amd import api, scenes, _native as N  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import bits, oracle_params_from_config, oracle_scene_from_instance  # noqa: E402

COUNTS = ("rays", "nodes_visited", "tris_tested", "hits", "shaded", "bad_materials", "stack_overflows")


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


def _instance(cache, name, w, h, pt, seed=31, **kw):
    spec = scenes.make_scene(name, cache, width=w, height=h, pt=pt)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=seed, **kw), api.InstanceConfig.last_error()
    return inst


def _render(cache, name, w, h, pt, spp, fused, env=None, calls=None, sun=None, **kw):
    with environment(**(env or {})):
        inst = _instance(cache, name, w, h, pt, **kw)
    p = inst.m_path_tracer
    p.SetFusedBounces(fused)
    if sun is not None:
        p.SetSunVisibility(True, sun)   # every escaped path sends its occlusion query: inside k_path<., SUN> / through the launch-per-bounce pipeline's query queue
    p.SetInstrumentation(counters=True)
    p.ResetStats()
    for n in (calls or [spp]):
        p.Trace(True, n)
    assert p.GetSPP() == spp
    out = {"img": p.ReadResult(), "hits": p.ReadHits(), "stats": p.GetStats(), "fused": p.GetFusedBounces(), "inst": inst}
    return out


def _same(a, b, what):
    assert np.array_equal(bits(a["img"]), bits(b["img"])), what + ": images differ"
    assert np.array_equal(a["hits"][0], b["hits"][0]) and np.array_equal(bits(a["hits"][1]), bits(b["hits"][1])), what + ": primary-hit caches differ"
    for k in COUNTS:
        assert a["stats"][k] == b["stats"][k], (what, k, a["stats"][k], b["stats"][k])


@pytest.mark.parametrize("name,w,h,pt,spp,calls", [
    ("tiny0", 100, 75, {"tmpLifetime": 4, "maxBounce": 6, "subpixel": 3}, 11, [8, 3]),
    ("tiny0", 96, 64, {"tmpLifetime": 16, "maxBounce": 8}, 40, [40]),            # more frames than fit one pass: several batches
    ("sibenik", 160, 90, {"tmpLifetime": 3, "maxBounce": 5, "subpixel": 2}, 7, [7]),
    ("tiny0", 64, 48, {"tmpLifetime": 1, "maxBounce": 2}, 9, [2, 7]),             # one bounce inside the launch
    ("tiny0", 72, 40, {"tmpLifetime": 2, "maxBounce": 32, "clamp": 2.5}, 6, [6]), # the longest paths the parameters allow
    ("tiny0", 16, 8, {"tmpLifetime": 16, "maxBounce": 8}, 5, [5]),                # fewer paths than one workgroup has slots
])
def test_one_launch_for_all_bounces_matches_per_bounce_launches_and_oracle(name, w, h, pt, spp, calls, scene_cache, sobol_matrices):
    ref = _render(scene_cache, name, w, h, pt, spp, fused=False, calls=calls)
    one = _render(scene_cache, name, w, h, pt, spp, fused=True, calls=calls)
    assert one["fused"] and not ref["fused"]
    assert ref["stats"]["path_rays"] == 0 and 0 < one["stats"]["path_rays"] < one["stats"]["rays"]  # k_path's own ray counter: every ray but the re-traced primaries
    _same(ref, one, "k_path against k_trace + k_shade")
    inst = one["inst"]
    c = inst.m_config.c
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    state = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, spp).as_dict()
    assert np.array_equal(bits(one["img"]), bits(state.accum[..., :3])), "k_path against the oracle"
    st = one["stats"]
    assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])


def test_max_bounce_one_needs_no_launch_at_all(scene_cache):
    pt = {"tmpLifetime": 4, "maxBounce": 1}
    ref = _render(scene_cache, "tiny0", 64, 36, pt, 6, fused=False)
    one = _render(scene_cache, "tiny0", 64, 36, pt, 6, fused=True)
    assert one["stats"]["path_rays"] == 0  # bounce 0 ends every path (k_shade_first): k_path is not launched
    _same(ref, one, "maxBounce 1")


@pytest.mark.parametrize("env", [
    {"ADYPT_PATH_LDS_DEPTH": 1},                                   # nearly every stack entry goes through the HBM spill array
    {"ADYPT_SHADE_MIN": 1, "ADYPT_REFILL_MIN": 1},                 # shading rounds of single paths, an exchange at every finished ray
    {"ADYPT_SHADE_MIN": 64, "ADYPT_REFILL_MIN": 64},               # a wave exchanges only when all its lanes are idle
    {"ADYPT_PATH_BLOCKS_PER_CU": 1},                               # 256 workgroups: long queues per workgroup
    {"ADYPT_PATH_BLOCKS_PER_CU": 5, "ADYPT_SHADE_MIN": 17, "ADYPT_REFILL_MIN": 7},
    {"ADYPT_FRAMES_IN_FLIGHT": 3},
    {"ADYPT_RARE_MIN": 0},                                         # glossy / dielectric hits are shaded where they are found
    {"ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_RARE_MIN": 1},          # every such hit is deferred and shaded in a round of its own
    {"ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_RARE_MIN": 64, "ADYPT_SHADE_MIN": 24, "ADYPT_REFILL_MIN": 5},
    {"ADYPT_RARE_MIN": 7, "ADYPT_SHADE_MIN": 1, "ADYPT_REFILL_MIN": 1},
    {"ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_DEFER_MAX": 64, "ADYPT_RARE_MIN": 16},   # a round defers however many it finds
    {"ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_DEFER_MAX": 1},                          # ... only a lone one
    {"ADYPT_REF_TRIANGLES_MAX_MB": 0},                              # no per-reference copy of the triangle records: the shading round applies uTriIndices (PathArgs::tri_remap)
    {"ADYPT_REF_TRIANGLES_MAX_MB": 0, "ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_RARE_MIN": 1},   # ... and a deferred hit is remapped again when its round comes
])
def test_scheduling_tunables_do_not_change_a_bit(env, scene_cache):
    pt = {"tmpLifetime": 4, "maxBounce": 7, "subpixel": 2, "stackSize": 24}
    ref = _render(scene_cache, "tiny0", 120, 68, pt, 10, fused=False, env={k: v for k, v in env.items() if k == "ADYPT_FRAMES_IN_FLIGHT"})
    one = _render(scene_cache, "tiny0", 120, 68, pt, 10, fused=True, env=env)
    assert one["stats"]["path_rays"] > 0  # (with 3 frames in flight the tenth frame is a batch of one, traced bounce by bounce)
    _same(ref, one, str(env))


@pytest.mark.parametrize("env", [
    {},
    {"ADYPT_SHADE_MIN": 1, "ADYPT_REFILL_MIN": 1},
    {"ADYPT_PATH_BLOCKS_PER_CU": 1, "ADYPT_RARE_MIN": 1, "ADYPT_PATH_LDS_DEPTH": 1},
    {"ADYPT_FRAMES_IN_FLIGHT": 3, "ADYPT_REF_TRIANGLES_MAX_MB": 0},
])
def test_sun_visibility_queries_inside_the_launch_under_scheduling_tunables(env, scene_cache):
    """The escaped paths' occlusion queries (rays that end at their first accepted triangle, among closest-hit rays, in one launch) against the launch-per-bounce
    pipeline's query queue + any-hit kernel: images, primary-hit cache and the exact ray / node / triangle / hit census, whatever the scheduling."""
    pt = {"tmpLifetime": 3, "maxBounce": 6, "subpixel": 2, "stackSize": 24}
    for name, w, h in (("tiny0", 120, 68), ("sibenik", 96, 54)):
        ref = _render(scene_cache, name, w, h, pt, 8, fused=False, env={k: v for k, v in env.items() if k == "ADYPT_FRAMES_IN_FLIGHT"}, sun=[-0.3, 0.8, 0.5])
        one = _render(scene_cache, name, w, h, pt, 8, fused=True, env=env, sun=[-0.3, 0.8, 0.5])
        assert one["stats"]["path_rays"] > 0 and one["stats"]["rays"] > _render(scene_cache, name, w, h, pt, 8, fused=True, env=env)["stats"]["rays"]  # (the queries are rays)
        _same(ref, one, name + str(env))


@pytest.mark.parametrize("nranks", [2, 3, 7])
def test_tile_shards_with_one_launch_reassemble_bit_exact(nranks, scene_cache):
    pt = {"tmpLifetime": 4, "maxBounce": 6}
    whole = _render(scene_cache, "tiny0", 160, 100, pt, 9, fused=False)
    img = np.zeros_like(whole["img"])
    rays = 0
    for r in range(nranks):
        part = _render(scene_cache, "tiny0", 160, 100, pt, 9, fused=True, tile_rank=r, tile_nranks=nranks)
        assert part["fused"] or part["stats"]["rays"] == 0
        mine = part["img"]
        owned = np.zeros(mine.shape[:2], bool)
        for by in range((100 + 31) // 32):
            for bx in range((160 + 31) // 32):
                if (bx + by) % nranks == r:
                    owned[by * 32:(by + 1) * 32, bx * 32:(bx + 1) * 32] = True
        img[owned] = mine[owned]
        rays += part["stats"]["rays"]
    assert np.array_equal(bits(img), bits(whole["img"]))
    assert rays == whole["stats"]["rays"]


def test_stack_overflow_is_reported_by_the_fused_launch_too(scene_cache):
    spec = scenes.make_scene("sibenik", scene_cache, width=96, height=54, pt={"tmpLifetime": 4, "maxBounce": 4, "stackSize": 1})
    for fused, sun in ((False, False), (True, False), (True, True)):   # (with the sun-visibility queries in the launch a failed push is committed when its node is visited)
        inst = api.Instance()
        assert inst.InitializeFromFile(spec.config_path, shift_seed=5)
        p = inst.m_path_tracer
        p.SetFusedBounces(fused)
        if sun:
            p.SetSunVisibility(True)
        with pytest.raises(N.AdyptError) as e:
            p.Trace(True, 6)
        assert e.value.code == N.E_STACK_OVERFLOW


def test_which_batches_take_the_fused_launch(scene_cache):
    inst = _instance(scene_cache, "tiny0", 64, 36, {"tmpLifetime": 4, "maxBounce": 5})
    p = inst.m_path_tracer
    p.Trace(True, 6)
    assert p.GetFusedBounces()                      # default on for batches
    p.Trace(True, 1)
    assert p.GetFusedBounces()                      # a single frame runs as a batch of one through the same four launches
    p.SetSunVisibility(True)
    p.Trace(True, 6)
    assert p.GetFusedBounces()                      # the escaped paths' sun-visibility queries ride in the same launch (k_path<., SUN>)
    p.SetPipeline(2)
    p.Trace(True, 6)
    assert not p.GetFusedBounces()                  # ... the sub-batch pipeline sends them through the query queue
    p.SetPipeline(1)
    p.SetSunVisibility(False)
    p.SetPipeline(2)
    p.Trace(True, 6)
    assert not p.GetFusedBounces()                  # the sub-batch pipeline keeps its launches
    p.SetPipeline(1)
    p.Trace(True, 6)
    assert p.GetFusedBounces()
    with environment(ADYPT_FUSED_BOUNCES=0):
        inst2 = _instance(scene_cache, "tiny0", 64, 36, {"tmpLifetime": 4, "maxBounce": 5})
    inst2.m_path_tracer.Trace(True, 6)
    assert not inst2.m_path_tracer.GetFusedBounces()
    with environment(ADYPT_SINGLE_FUSED=0):
        inst3 = _instance(scene_cache, "tiny0", 64, 36, {"tmpLifetime": 4, "maxBounce": 5})
    inst3.m_path_tracer.Trace(True, 1)
    assert not inst3.m_path_tracer.GetFusedBounces()  # ... unless told to keep the launch-per-bounce frame
    inst3.m_path_tracer.Trace(True, 6)
    assert inst3.m_path_tracer.GetFusedBounces()


@pytest.mark.parametrize("single_fused", [1, 0])
def test_frame_by_frame_calls_match_oracle_through_either_single_frame_pipeline(single_fused, scene_cache, sobol_matrices):
    """adypt_trace_spp(ctx, 1) per call without look-ahead (what Instance::Update does): as a batch of one (camera launch, k_shade_first, k_path,
    k_resolve) or as gen -> [trace -> shade] x maxBounce; frames that re-trace their primaries (every 3rd here) and frames that start from the cache."""
    pt = {"tmpLifetime": 3, "maxBounce": 5, "subpixel": 2}
    with environment(ADYPT_SINGLE_FUSED=single_fused):
        inst = _instance(scene_cache, "tiny0", 104, 70, pt)
    p = inst.m_path_tracer
    p.SetLookahead(False)
    p.SetInstrumentation(counters=True)
    p.ResetStats()
    c = inst.m_config.c
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    state = O.PathTracerState(c.width, c.height)
    shift = O.shift_bytes(31, c.width, c.height)
    rays = 0
    for i in range(7):
        p.Trace(True, 1)
        assert p.GetFusedBounces() == bool(single_fused)
        rays += O.pt_frames(osc, P, shift, sobol_matrices, state, 1).as_dict()["rays"]
        assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), i
        tri, uv = p.ReadHits()
        assert np.array_equal(tri, state.cache_tri) and np.array_equal(bits(uv), bits(state.cache_uv)), i
    assert p.GetStats()["rays"] == rays and p.GetSPP() == 7


def test_material_zoo_through_the_fused_launch(tmp_path):
    """Every branch of Render's illum switch (diffuse, glossy, mirrors, dielectrics, pass-through, emitters: the scene of
    test_gpu_edge_cases.py::test_material_zoo_matches_oracle, which pins it to the oracle), several bounces deep, through both pipelines."""
    from tests.test_gpu_edge_cases import _tracer, write_zoo
    obj, _ = write_zoo(tmp_path)
    imgs = []
    for fused in (False, True):
        sc, b, p, params = _tracer(obj, 96, 54)
        params.max_bounce = 7
        p.SetConfig(params)
        ip, iv = api.camera_matrices(75.0, 0.0, -55.0, 96, 54)
        p.SetCamera(ip, iv, [6.0, 4.5, 8.5])
        p.SetFusedBounces(fused)
        p.SetInstrumentation(counters=True)
        p.Trace(True, 9)
        imgs.append((p.ReadResult(), p.GetStats(), p.GetFusedBounces()))
    assert imgs[1][2] and not imgs[0][2]
    assert np.array_equal(bits(imgs[0][0]), bits(imgs[1][0]))
    for k in COUNTS:
        assert imgs[0][1][k] == imgs[1][1][k], k
