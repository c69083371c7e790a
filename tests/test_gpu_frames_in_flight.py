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


@pytest.mark.parametrize("name,w,h,life,fif", [("tiny0", 100, 75, 4, 13), ("tiny0", 96, 64, 16, 32), ("sibenik", 160, 90, 3, 8), ("tiny0", 72, 40, 1, 5)])
def test_lookahead_hands_out_identical_frames_one_per_call(name, w, h, life, fif, scene_cache, sobol_matrices):
    """adypt_set_lookahead: Instance::Update's pattern — ONE Trace(true) per call (src/Instance.cpp:44-57) — served from
    whole wavefront passes traced ahead.  After EVERY call the image, the spp counter and image 1 (primary-hit cache) equal
    what the oracle's frame-by-frame loop has after that frame; a camera change / reset drops the parked frames."""
    inst = _instance(scene_cache, name, w, h, {"tmpLifetime": life, "maxBounce": 6, "subpixel": 3})
    c, p = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    shift = O.shift_bytes(31, c.width, c.height)
    p.SetFramesInFlight(fif)
    p.SetLookahead(True)
    state = O.PathTracerState(c.width, c.height)
    total = 2 * fif + 3
    for k in range(total):
        p.Trace(True, 1)
        O.pt_frames(osc, P, shift, sobol_matrices, state, 1)
        assert p.GetSPP() == k + 1
        assert p.GetLookaheadFrames() == (fif - 1 - k % fif)
        assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), "frame %d" % k
        tri, uv = p.ReadHits()
        assert np.array_equal(tri, state.cache_tri), "image 1 after frame %d" % k
        m = tri >= 0
        assert np.array_equal(bits(uv)[m], bits(state.cache_uv)[m])
    # mixed call sizes continue the same sequence (parked frames + a new pass + ...)
    p.Trace(True, fif + 2)
    O.pt_frames(osc, P, shift, sobol_matrices, state, fif + 2)
    assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])) and p.GetSPP() == total + fif + 2
    n = p.GetLookaheadFrames()
    p.Trace(True, n)
    if n:
        O.pt_frames(osc, P, shift, sobol_matrices, state, n)
    assert p.GetLookaheadFrames() == 0 and np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3]))
    # a camera change drops the parked frames: the next frame is traced with the new camera, the cache image keeps the old
    # primary hits until the next re-tracing frame — exactly the reference's behaviour (pathtracer.glsl:113-127)
    p.Trace(True, 1)
    O.pt_frames(osc, P, shift, sobol_matrices, state, 1)
    assert p.GetLookaheadFrames() == fif - 1
    ip, iv = O.camera(c.fov + 7.0, c.yaw + 11.0, c.pitch - 3.0, c.width, c.height)
    p.SetCamera(ip, iv, list(c.position))
    assert p.GetLookaheadFrames() == 0
    P2 = O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce, subpixel=c.subpixel,
                       tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))
    for k in range(life + 2):
        p.Trace(True, 1)
        O.pt_frames(osc, P2, shift, sobol_matrices, state, 1)
        assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), "after camera change, frame %d" % k
    # reset restarts the sequence
    p.Reset()
    assert p.GetLookaheadFrames() == 0
    p.Trace(True, 2)
    st2 = O.PathTracerState(c.width, c.height)
    O.pt_frames(osc, P2, shift, sobol_matrices, st2, 2)
    assert np.array_equal(bits(p.ReadResult()), bits(st2.accum[..., :3]))


def test_lookahead_work_counters_match_after_full_hand_out(scene_cache, sobol_matrices):
    inst = _instance(scene_cache, "tiny0", 96, 64, {"tmpLifetime": 4, "maxBounce": 6, "subpixel": 3})
    c, p = inst.m_config.c, inst.m_path_tracer
    p.SetFramesInFlight(8)
    p.SetLookahead(True)
    p.SetInstrumentation(counters=True)
    p.ResetStats()
    for _ in range(16):
        p.Trace(True, 1)
    assert p.GetLookaheadFrames() == 0
    state = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(oracle_scene_from_instance(inst), oracle_params_from_config(c), O.shift_bytes(31, c.width, c.height), sobol_matrices, state, 16).as_dict()
    st = p.GetStats()
    assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])
    assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3]))


def test_long_progressive_run_is_bit_exact(scene_cache, sobol_matrices):
    """BASELINE config 5 in miniature: > 1024 samples accumulated progressively.  Far along the Sobol sequence (gray-code bit 10, every
    direction-number row of the table in use), the sub-pixel index has wrapped around its 8 x 8 grid (spp / tmpLifetime >= 64), and the
    running mean (out * spp + r) / (spp + 1) has been applied 1100 times in fp32: still every bit equal to the oracle's."""
    inst = _instance(scene_cache, "tiny0", 24, 16, {"tmpLifetime": 16, "maxBounce": 3, "subpixel": 8}, seed=8)
    p, c = inst.m_path_tracer, inst.m_config.c
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    state = O.PathTracerState(c.width, c.height)
    shift = O.shift_bytes(8, c.width, c.height)
    for n in (500, 1, 599):
        p.Trace(True, n)
        O.pt_frames(osc, P, shift, sobol_matrices, state, n)
        assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), "after %d samples" % p.GetSPP()
    assert p.GetSPP() == 1100


@pytest.mark.parametrize("name,w,h,life,fif,spp,sunvis", [("tiny0", 100, 75, 4, 13, 29, False), ("tiny0", 96, 64, 16, 32, 40, False),
                                                            ("sibenik", 160, 90, 3, 7, 17, False), ("tiny0", 72, 40, 1, 5, 12, True),
                                                            ("tiny0", 64, 48, 5, 3, 8, True)])
def test_pipeline_of_sub_batches_is_bit_invariant(name, w, h, life, fif, spp, sunvis, scene_cache, sobol_matrices, monkeypatch):
    """adypt_set_pipeline: a batch cut into 1..4 sub-batches, each the chain camera rays -> [traversal -> shade] x maxBounce on
    its own HIP stream in its own window of the ray queues (the chains overlap on the GPU).  Image, spp, image 1 and the exact
    work counters equal the oracle's / the serial chain's for every split, including sub-batches of unequal size, a batch
    smaller than the number of pipes, the sun-visibility queues and traversal stacks that spill out of LDS."""
    monkeypatch.setenv("ADYPT_LDS_STACK_DEPTH", "2")  # deeper entries go to the per-pipe spill arrays: two pipes' launches overlap
    pt_cfg = {"tmpLifetime": life, "maxBounce": 6, "subpixel": 3}
    ref = None
    for pipes in (1, 2, 3, 4):
        inst = _instance(scene_cache, name, w, h, pt_cfg)
        p, c = inst.m_path_tracer, inst.m_config.c
        p.SetFramesInFlight(fif)
        p.SetPipeline(pipes)
        assert p.GetPipeline() == pipes
        if sunvis:
            p.SetSunVisibility(True)
        p.SetInstrumentation(counters=True)
        p.ResetStats()
        p.Trace(True, spp - 2)
        p.Trace(True, 2)
        img, st = p.ReadResult(), p.GetStats()
        tri, uv = p.ReadHits()
        assert p.GetSPP() == spp and st["stack_overflows"] == 0
        if ref is None:
            ref = (img, st, tri, uv)
            if not sunvis:
                osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
                state = O.PathTracerState(c.width, c.height)
                ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, spp).as_dict()
                assert np.array_equal(bits(img), bits(state.accum[..., :3]))
                assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])
        else:
            assert np.array_equal(bits(img), bits(ref[0])), "%d pipes changed the image" % pipes
            for k in ("rays", "nodes_visited", "tris_tested", "hits", "shaded", "max_stack"):
                assert st[k] == ref[1][k], (pipes, k)
            assert np.array_equal(tri, ref[2]) and np.array_equal(bits(uv), bits(ref[3]))


@pytest.mark.parametrize("name,w,h,life,fif,spp", [("tiny0", 100, 75, 4, 11, 23), ("sibenik", 160, 90, 16, 1, 3), ("tiny0", 96, 64, 2, 32, 33)])
def test_segment_assignment_of_new_paths_is_bit_invariant(name, w, h, life, fif, spp, scene_cache, sobol_matrices, monkeypatch):
    """k_gen_primary deals chunks of 256 paths round-robin over the 8 queue segments (default) or gives each segment one contiguous
    run (ADYPT_GEN_DEAL=0, rounds 1-2): which segment — which XCD — a path lives in is scheduling only.  Likewise bounce 0 of a batch in
    one kernel with the surface fetched once per pixel and tmpLifetime group (k_shade_first, default) or as k_gen_primary + k_shade.  Image, image 1 and the
    exact work counters are the oracle's either way, for batches, single frames, and batches that do not fill their last chunk."""
    results = []
    for deal in ("1", "0"):
        monkeypatch.setenv("ADYPT_GEN_DEAL", deal)
        monkeypatch.setenv("ADYPT_FIRST_FUSED", deal)  # "0": camera rays and bounce 0 of a batch as k_gen_primary + k_shade instead of k_shade_first
        inst = _instance(scene_cache, name, w, h, {"tmpLifetime": life, "maxBounce": 5, "subpixel": 2})
        p, c = inst.m_path_tracer, inst.m_config.c
        p.SetFramesInFlight(fif)
        p.SetInstrumentation(counters=True)
        p.ResetStats()
        p.Trace(True, spp)
        img, st = p.ReadResult(), p.GetStats()
        tri, uv = p.ReadHits()
        results.append((img, st, tri, uv))
        if deal == "1":
            osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
            state = O.PathTracerState(c.width, c.height)
            ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, spp).as_dict()
            assert np.array_equal(bits(img), bits(state.accum[..., :3]))
            assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])
        p.destroy()
    (a_img, a_st, a_tri, a_uv), (b_img, b_st, b_tri, b_uv) = results
    assert np.array_equal(bits(a_img), bits(b_img)) and np.array_equal(a_tri, b_tri) and np.array_equal(bits(a_uv), bits(b_uv))
    assert (a_st["rays"], a_st["nodes_visited"], a_st["tris_tested"]) == (b_st["rays"], b_st["nodes_visited"], b_st["tris_tested"])


def test_shade_binning_option_and_shader_clock(scene_cache, sobol_matrices, monkeypatch):
    """ADYPT_SHADE_BIN=1 (k_shade deals a workgroup's paths to its threads sorted by material class; off by default because it measured
    slower) is scheduling only: image, image 1 and counters equal the oracle's on the material-zoo scene with textures.  Also: the
    traversal launches report the shader clock they ran at (adypt_get_shader_clock), a plausible one."""
    monkeypatch.setenv("ADYPT_SHADE_BIN", "1")
    inst = _instance(scene_cache, "tiny0", 100, 75, {"tmpLifetime": 3, "maxBounce": 6, "subpixel": 2})
    p, c = inst.m_path_tracer, inst.m_config.c
    p.SetFramesInFlight(7)
    p.SetInstrumentation(counters=True)
    p.ResetStats()
    assert p.GetShaderClockGHz() == 0.0  # nothing traced since the reset
    p.Trace(True, 17)
    img, st = p.ReadResult(), p.GetStats()
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    state = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, state, 17).as_dict()
    assert np.array_equal(bits(img), bits(state.accum[..., :3]))
    assert (st["rays"], st["nodes_visited"], st["tris_tested"], st["shaded"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["shaded"])
    ghz = p.GetShaderClockGHz()
    assert 0.8 < ghz < 3.0, ghz
    p.destroy()


@pytest.mark.parametrize("name,w,h,life,fif,lookahead", [("tiny0", 100, 75, 4, 1, False), ("tiny0", 100, 75, 4, 1, True), ("tiny0", 96, 64, 3, 8, False),
                                                          ("sibenik", 160, 90, 2, 1, True), ("tiny0", 72, 40, 1, 1, True)])
def test_single_frames_in_a_row_overlap_and_stay_bit_exact(name, w, h, life, fif, lookahead, scene_cache, sobol_matrices, monkeypatch):
    """One frame per wavefront pass (Instance::Update's Trace(true) per window frame, src/Instance.cpp:44-57, without a pass traced ahead): frame k + 1's
    bounce 0 and k_path are enqueued on a second stream under the END of frame k's k_path — inside a call that asks for several frames, and across calls
    when look-ahead is on (one frame ahead, never across a tmpLifetime boundary).  After EVERY call image, spp and image 1 are the oracle's; traversal
    stacks spill out of LDS (two frames' launches overlap: their spill arrays must be their own); a camera change, a ray batch in between and a reset
    drop the frame started ahead; ADYPT_SINGLE_OVERLAP=0 (strictly one after the other) gives the same bits."""
    monkeypatch.setenv("ADYPT_LDS_STACK_DEPTH", "2")
    inst = _instance(scene_cache, name, w, h, {"tmpLifetime": life, "maxBounce": 6, "subpixel": 3})
    c, p = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    shift = O.shift_bytes(31, c.width, c.height)
    p.SetFramesInFlight(fif)
    p.SetLookahead(lookahead)
    state = O.PathTracerState(c.width, c.height)

    def step(n, tag):
        p.Trace(True, n)
        O.pt_frames(osc, P, shift, sobol_matrices, state, n)
        assert p.GetSPP() == state.spp
        assert np.array_equal(bits(p.ReadResult()), bits(state.accum[..., :3])), tag
        tri, uv = p.ReadHits()
        assert np.array_equal(tri, state.cache_tri), "image 1, " + tag
        m = tri >= 0
        assert np.array_equal(bits(uv)[m], bits(state.cache_uv)[m]), "image 1 uv, " + tag

    for k in range(2 * life + 3):          # one frame per call, across tmpLifetime boundaries
        step(1, "call %d" % k)
    if fif == 1:
        step(5, "five frames in one call")  # frames overlap inside the call
    assert p.GetFusedBounces()
    # a batch of rays through the same queues between two frames (drops a frame started ahead)
    rays = np.zeros((64, 8), np.float32)
    rays[:, 0:3] = np.asarray(c.position, np.float32); rays[:, 3] = c.ray_tmin
    rays[:, 4:7] = np.random.default_rng(5).normal(size=(64, 3)).astype(np.float32)
    h1 = p.TraceRays(rays, with_stats=False)
    step(1, "after a ray batch")
    h2 = p.TraceRays(rays, with_stats=False)
    assert np.array_equal(h1["ref_idx"], h2["ref_idx"])
    # a camera change: the frame started ahead saw the old camera and must not be used
    step(1, "before the camera change")
    ip, iv = O.camera(c.fov + 5.0, c.yaw - 9.0, c.pitch + 2.0, c.width, c.height)
    p.SetCamera(ip, iv, list(c.position))
    P = O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce, subpixel=c.subpixel,
                      tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))
    for k in range(life + 2):
        step(1, "after the camera change, frame %d" % k)
    # a reset restarts the sequence; strictly serial frames give the same bits
    p.Reset()
    state = O.PathTracerState(c.width, c.height)
    step(3, "after reset")
    img = p.ReadResult()
    p.destroy()
    monkeypatch.setenv("ADYPT_SINGLE_OVERLAP", "0")
    inst2 = _instance(scene_cache, name, w, h, {"tmpLifetime": life, "maxBounce": 6, "subpixel": 3})
    p2 = inst2.m_path_tracer
    p2.SetFramesInFlight(fif)
    p2.SetCamera(ip, iv, list(c.position))
    for _ in range(3):
        p2.Trace(True, 1)
    assert np.array_equal(bits(p2.ReadResult()), bits(img))
    p2.destroy()
