"""-m gpu: degenerate scenes through the HIP kernels vs the oracle, bit for bit.  Zero-area triangles (collinear,
collapsed to a point) have a singular Woop matrix — glm::inverse divides by a zero determinant, so their 48-byte
records hold inf/NaN (OglScene.cpp:93-116 does not guard it); the traversal must carry those through its comparisons
exactly like the restatement does.  Also: a one-triangle scene (the reference's collapse crashes; the product builds a
one-child root), rays that start on / run inside triangle planes, empty ray batches, 1x1 and odd-sized images."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from adypt_amd import api, _native as N  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.helpers import GOLDEN, bits  # noqa: E402


def _tracer(obj_path, w, h, stack=24):
    sc = api.Scene()
    assert sc.LoadFromFile(obj_path)
    cfg = api.InstanceConfig()
    b = api.WideBVH()
    b.Build(sc, cfg.bvh_params())
    hs = api.HipScene()
    hs.Initialize(sc, b)
    p = cfg.pt_params(77)
    p.stack_size, p.max_bounce, p.subpixel, p.tmp_lifetime = stack, 4, 2, 3
    p.sun[:] = [3.0, 2.0, 1.0]
    pt = api.HipPathTracer()
    pt.Initialize(p, hs, w, h)
    return sc, b, pt, p


def _rays(n, seed):
    rs = np.random.RandomState(seed)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(-2, 12, size=(n, 3)) * np.array([1, 0.2, 0.2], np.float32)
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(n, 3))
    k = n // 8
    rays[:k, 6] = 0.0                      # inside the z = 0 plane of the flat / degenerate cases
    rays[k:2 * k, 5:7] = 0.0               # along the x axis: through collinear triangles
    rays[2 * k:3 * k, :3] = 0.0            # origin on the shared vertex / the collapsed point
    rays[3 * k:4 * k, 4:7] = [0.0, 0.0, -1.0]
    return rays


@pytest.mark.parametrize("name", ["rand_1", "rand_2", "rand_9", "degenerate_5", "degenerate_9", "flat_9", "point_4"])
def test_degenerate_scene_traversal_and_frames_match_oracle(name, tmp_path, sobol_matrices):
    c = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))[name]
    obj = tmp_path / (name + ".obj")
    obj.write_text(c["obj"])
    w, h = 37, 23
    sc, b, pt, p = _tracer(str(obj), w, h)
    osc = O.Scene(b.nodes, b.tri_indices, sc.triangles, sc.materials)
    rays = _rays(4096, 3)
    assert pt.TraceRays(rays, with_stats=True).tobytes() == O.trace(osc, rays, p.stack_size).tobytes()
    assert pt.TraceRays(rays[:0], with_stats=True).shape[0] == 0            # empty batch
    ip, iv = api.camera_matrices(60.0, 0.0, -5.0, w, h)
    pos = [4.0, 0.3, 6.0]
    pt.SetCamera(ip, iv, pos)
    P = O.make_params(w, h, pos, ip, iv, stack_size=p.stack_size, max_bounce=p.max_bounce, subpixel=p.subpixel,
                      tmp_life=p.tmp_lifetime, tmin=p.ray_tmin, clamp=p.clamp, sun=list(p.sun))
    pt.Trace(False)
    rgba, _, _ = O.primary_frame(osc, P, 0)
    assert np.array_equal(bits(pt.ReadResult()), bits(rgba[..., :3]))
    pt.SetInstrumentation(counters=True)
    pt.ResetStats()
    pt.Trace(True, 7)                                                        # crosses two tmpLifetime groups
    st = O.PathTracerState(w, h)
    ost = O.pt_frames(osc, P, O.shift_bytes(77, w, h), sobol_matrices, st, 7).as_dict()
    assert np.array_equal(bits(pt.ReadResult()), bits(st.accum[..., :3]))
    g = pt.GetStats()
    assert (g["rays"], g["nodes_visited"], g["tris_tested"], g["hits"]) == (ost["rays"], ost["nodes"], ost["tris"], ost["hits"])


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (1, 3), (33, 31), (65, 2)])
def test_minimal_and_odd_image_sizes(w, h, tmp_path, sobol_matrices):
    c = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))["rand_25"]
    obj = tmp_path / "s.obj"
    obj.write_text(c["obj"])
    sc, b, pt, p = _tracer(str(obj), w, h)
    osc = O.Scene(b.nodes, b.tri_indices, sc.triangles, sc.materials)
    ip, iv = api.camera_matrices(70.0, 10.0, 0.0, w, h)
    pos = [0.1, 0.0, 3.0]
    pt.SetCamera(ip, iv, pos)
    P = O.make_params(w, h, pos, ip, iv, stack_size=p.stack_size, max_bounce=p.max_bounce, subpixel=p.subpixel,
                      tmp_life=p.tmp_lifetime, tmin=p.ray_tmin, clamp=p.clamp, sun=list(p.sun))
    pt.Trace(True, 5)
    st = O.PathTracerState(w, h)
    O.pt_frames(osc, P, O.shift_bytes(77, w, h), sobol_matrices, st, 5)
    img = pt.ReadResult()
    assert img.shape == (h, w, 3)
    assert np.array_equal(bits(img), bits(st.accum[..., :3]))


@pytest.mark.parametrize("lds_depth", [1, 2, 3])
def test_stack_spill_to_global_memory_is_bit_exact(lds_depth, scene_cache, sobol_matrices, monkeypatch):
    """The node-group stack keeps its first entries in LDS and spills deeper ones to a global array; ordinary scenes never
    get past the LDS part (8 entries), so the spill path is forced here by shrinking the LDS part to 1..3 entries."""
    from adypt_amd import scenes
    monkeypatch.setenv("ADYPT_LDS_STACK_DEPTH", str(lds_depth))
    spec = scenes.make_scene("sibenik", scene_cache, width=160, height=90, pt={"maxBounce": 4, "stackSize": 16})
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=3)
    pt, c = inst.m_path_tracer, inst.m_config.c
    osc = O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)
    rs = np.random.RandomState(9)
    p = np.frombuffer(inst.scene.triangles.tobytes(), dtype=O.TRI_DT)["p"].reshape(-1, 3)
    rays = np.zeros((20000, 8), np.float32)
    rays[:, :3] = rs.uniform(p.min(0), p.max(0), size=(20000, 3))
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(20000, 3))
    g, o = pt.TraceRays(rays, with_stats=True), O.trace(osc, rays, c.stack_size)
    assert g.tobytes() == o.tobytes()
    assert g["max_depth"].max() > lds_depth          # entries really went to the spill array
    ga, oa = pt.TraceRays(rays, with_stats=True, any_hit=True), O.trace(osc, rays, c.stack_size, any_hit=True)
    assert ga.tobytes() == oa.tobytes()
    ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
    P = O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce,
                      subpixel=c.subpixel, tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))
    pt.Trace(True, 3)
    st = O.PathTracerState(c.width, c.height)
    O.pt_frames(osc, P, O.shift_bytes(3, c.width, c.height), sobol_matrices, st, 3)
    assert np.array_equal(bits(pt.ReadResult()), bits(st.accum[..., :3]))


def test_c_abi_error_behaviour_on_the_device(tmp_path):
    """SURVEY.md §8b 'Errors': every entry point returns 0 or a negative ADYPT_E_* and never crashes on misuse — null
    handles / pointers, tracing before a camera is set, parameters outside their ranges, a device ordinal that does not
    exist — and the context stays usable afterwards."""
    import ctypes as C
    from adypt_amd import _native as N
    c = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))["rand_25"]
    obj = tmp_path / "s.obj"
    obj.write_text(c["obj"])
    sc = api.Scene()
    assert sc.LoadFromFile(str(obj))
    cfg = api.InstanceConfig()
    b = api.WideBVH()
    b.Build(sc, cfg.bvh_params())
    hs = api.HipScene()
    hs.Initialize(sc, b)
    lib = N.lib
    null = C.c_void_p()
    # null context everywhere
    buf = (C.c_float * 16)()
    for call in (lambda: lib.adypt_trace_spp(null, 1), lambda: lib.adypt_trace_spp_async(null, 1), lambda: lib.adypt_wait(null),
                 lambda: lib.adypt_trace_primary(null, 0), lambda: lib.adypt_reset(null), lambda: lib.adypt_get_spp(null),
                 lambda: lib.adypt_read_radiance(null, buf), lambda: lib.adypt_read_display(null, buf), lambda: lib.adypt_reset_stats(null),
                 lambda: lib.adypt_set_frames_in_flight(null, 2), lambda: lib.adypt_set_instrumentation(null, 1)):
        assert call() == N.E_INVALID
    lib.adypt_destroy(null)  # no-op
    # a device that does not exist
    pt = api.HipPathTracer()
    with pytest.raises(N.AdyptError) as e:
        pt.Initialize(cfg.pt_params(1), hs, 32, 32, device=63)
    assert e.value.code == N.E_INVALID and "device" in str(e.value)
    pt = api.HipPathTracer()
    pt.Initialize(cfg.pt_params(1), hs, 40, 24)
    ctx = pt._ctx
    # tracing before the camera is set
    assert lib.adypt_trace_spp(ctx, 1) == N.E_STATE and b"adypt_set_camera" in lib.adypt_last_error(ctx)
    assert lib.adypt_trace_primary(ctx, 0) == N.E_STATE
    # parameters outside their ranges are rejected and leave the active ones untouched
    for field, value in (("stack_size", 0), ("stack_size", 65), ("max_bounce", 0), ("max_bounce", 33), ("subpixel", 0), ("tmp_lifetime", 0)):
        p = cfg.pt_params(1)
        setattr(p, field, value)
        assert lib.adypt_set_params(ctx, C.byref(p)) == N.E_INVALID, field
    assert lib.adypt_set_frames_in_flight(ctx, 0) == N.E_INVALID and lib.adypt_set_frames_in_flight(ctx, 100000) == N.E_INVALID
    assert lib.adypt_trace_spp(ctx, -1) == N.E_INVALID
    assert lib.adypt_read_radiance(ctx, None) == N.E_INVALID and lib.adypt_read_display(ctx, None) == N.E_INVALID
    assert lib.adypt_trace_rays(ctx, None, 5, None, 0) == N.E_INVALID
    assert lib.adypt_assemble_radiance(ctx, None, 0, None) == N.E_INVALID
    # ... and the context still renders
    ip, iv = api.camera_matrices(60.0, 10.0, 0.0, 40, 24)
    pt.SetCamera(ip, iv, [0.1, 0.0, 3.0])
    pt.Trace(True, 2)
    assert pt.GetSPP() == 2 and np.isfinite(pt.ReadResult()).all()
    assert lib.adypt_trace_spp(ctx, 0) == 0 and pt.GetSPP() == 2


@pytest.mark.parametrize("name,w,h,fif", [("tiny0", 96, 64, 1), ("tiny0", 100, 75, 5), ("sponza", 192, 108, 32), ("sibenik", 160, 90, 3)])
def test_sun_visibility_option_matches_oracle(name, w, h, fif, scene_cache, sobol_matrices):
    """the occlusion query the reference has commented out (pathtracer.glsl:132) as an option — escaped
    paths receive the sun term only if an any-hit ray towards the sun finds nothing.  Bit-exact against the oracle with the
    same option, for single frames and batches; the shadow rays are counted as rays; off = unchanged."""
    from adypt_amd import scenes
    spec = scenes.make_scene(name, scene_cache, width=w, height=h, pt={"maxBounce": 5, "stackSize": 24, "tmpLifetime": 3})
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=21)
    pt, c = inst.m_path_tracer, inst.m_config.c
    osc = O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)
    ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
    common = dict(stack_size=c.stack_size, max_bounce=c.max_bounce, subpixel=c.subpixel, tmp_life=c.tmp_lifetime, tmin=c.ray_tmin,
                  clamp=c.clamp, sun=list(c.sun))
    pt.SetFramesInFlight(fif)
    results = {}
    for label, direction in (("off", None), ("default", None), ("custom", [-0.3, 0.8, 0.5])):
        on = label != "off"
        P = O.make_params(c.width, c.height, list(c.position), ip, iv, sun_visibility=on,
                          sun_dir=direction if direction is not None else (0.6, 1.0, 0.2), **common)
        st = O.PathTracerState(c.width, c.height)
        ost = O.pt_frames(osc, P, O.shift_bytes(21, c.width, c.height), sobol_matrices, st, 7).as_dict()
        # the queries ride inside the one-launch pipeline (k_path<., SUN>: rays that end at their first accepted triangle among the others) — with the
        # instrumented kernel (exact census: the queries visit what the oracle's any-hit queries visit) and with the product kernel — and, for comparison,
        # through the launch-per-bounce pipeline's query queue
        for counters, fused in ((True, True), (False, True), (True, False)):
            pt.SetInstrumentation(counters=counters)
            pt.SetFusedBounces(fused)
            pt.SetSunVisibility(on, direction)
            pt.Reset()
            pt.ResetStats()
            pt.Trace(True, 7)
            assert pt.GetFusedBounces() == fused
            img, g = pt.ReadResult(), pt.GetStats()
            assert np.array_equal(bits(img), bits(st.accum[..., :3])), (label, counters, fused)
            assert g["rays"] == ost["rays"], (label, counters, fused)
            if counters:
                assert (g["nodes_visited"], g["tris_tested"], g["hits"], g["shaded"]) == (ost["nodes"], ost["tris"], ost["hits"], ost["shaded"]), (label, fused)
        pt.SetFusedBounces(True)
        results[label] = (img, g["rays"])
    assert results["default"][1] >= results["off"][1]                       # the queries are rays
    assert (results["default"][0] <= results["off"][0]).all()               # occlusion only removes light
    with pytest.raises(Exception):
        pt.SetSunVisibility(True, [0.0, 0.0, 0.0])


def test_read_after_async_trace_without_wait(scene_cache, sobol_matrices):
    """adypt_read_radiance / adypt_read_hits synchronise with the context's (non-blocking) stream themselves: reading right
    after adypt_trace_spp_async, with no adypt_wait, returns the finished frames.  Also adypt_reset + adypt_set_params
    (new shift seed) right behind asynchronous work must not overwrite the shift image under the running kernels."""
    from tests.helpers import oracle_params_from_config, oracle_scene_from_instance
    from tests.test_gpu_parity import make_instance
    inst = make_instance(scene_cache, "sponza", 640, 360, seed=31, pt={"maxBounce": 8, "stackSize": 24})
    c, pt = inst.m_config.c, inst.m_path_tracer
    osc, P = oracle_scene_from_instance(inst), oracle_params_from_config(c)
    pt.SetFramesInFlight(4)
    pt.TraceAsync(24)          # six batches queued on the stream, the host returns at once
    a = pt.ReadResult()        # no Wait()
    tri, _ = pt.ReadHits()
    st = O.PathTracerState(c.width, c.height)
    O.pt_frames(osc, P, O.shift_bytes(31, c.width, c.height), sobol_matrices, st, 24)
    assert np.array_equal(bits(a), bits(st.accum[..., :3])) and np.array_equal(tri, st.cache_tri)
    pt.Wait()
    # new seed handed over while frames of the old seed are still in flight
    pt.Reset()
    pt.TraceAsync(24)
    pt.Reset()
    pt.SetConfig(inst.m_config.pt_params(77))
    pt.Trace(True, 3)
    st2 = O.PathTracerState(c.width, c.height)
    O.pt_frames(osc, P, O.shift_bytes(77, c.width, c.height), sobol_matrices, st2, 3)
    assert np.array_equal(bits(pt.ReadResult()), bits(st2.accum[..., :3]))


@pytest.mark.parametrize("w,h,world", [(64, 36, 4), (128, 128, 8), (96, 64, 8)])
def test_tile_shard_that_owns_no_block(w, h, world, scene_cache):
    """More ranks than 32x32 block diagonals: some contexts own nothing (n_local_px == 0).  They must trace nothing (the
    kernels divide by the local pixel count), keep the frame counter in step, read back nothing, and the union of all
    shards must still equal the 1-context frame."""
    from adypt_amd import distributed as D
    from tests.test_gpu_parity import make_instance
    full = make_instance(scene_cache, "tiny0", w, h, seed=5)
    full.m_path_tracer.Trace(True, 3)
    ref = full.m_path_tracer.ReadResult()
    out, empty, rays = np.zeros_like(ref), 0, 0
    for r in range(world):
        part = make_instance(scene_cache, "tiny0", w, h, seed=5, rank=r, world=world)
        pt = part.m_path_tracer
        n = pt.local_pixel_count()
        assert n == D.block_count(w, h, r, world) * 1024
        pt.Trace(False)
        pt.Trace(True, 2)
        pt.TraceAsync(1)
        img = pt.ReadResult()
        pt.Wait()
        assert pt.GetSPP() == 3
        mask = D.owner_mask(w, h, r, world).astype(bool)
        if n == 0:
            empty += 1
            assert not mask.any() and not img.any() and pt.GetStats()["rays"] == 0
            assert not pt.ReadDisplay().any() and (pt.ReadHits()[0] == -1).all()
        out[mask] = img[mask]
        rays += pt.GetStats()["rays"]
        pt.destroy()
    assert empty > 0
    assert np.array_equal(bits(out), bits(ref))


def test_corrupt_inner_child_bits_are_rejected():
    """validate_bvh mirrors the kernel's addressing: an inner slot whose child bits are not 0b001 would raise hit-mask bits
    outside imask (one node past the validated child range) — adypt_create must refuse the array."""
    from tests.helpers import golden_scene
    from tests.test_gpu_parity import golden_tracer
    _, idx, nodes, tris, mats, woop = golden_scene("tiny0")
    raw = nodes.view("u1").reshape(-1, 80).copy()
    hit = False
    for i in range(len(raw)):
        for s in range(8):
            m = int(raw[i, 24 + s])
            if m and (m & 0x18) == 0x18:   # inner slot: low five bits >= 24
                raw[i, 24 + s] = (m & 31) | (3 << 5)
                hit = True
                break
        if hit:
            break
    assert hit
    sc = api.Scene.FromArrays(tris, mats)
    b = api.WideBVH()
    b.nodes, b.tri_indices = raw.reshape(-1), idx
    hs = api.HipScene()
    hs.Initialize(sc, b, woop=woop)
    with pytest.raises(N.AdyptError) as e:
        api.HipPathTracer().Initialize(api.InstanceConfig().pt_params(0), hs, 64, 36)
    assert e.value.code == N.E_INVALID and "child bits" in str(e.value)


def test_subpixel_upper_bound():
    from tests.test_gpu_parity import golden_tracer
    pt = golden_tracer("tiny0")
    p = api.InstanceConfig().pt_params(0)
    p.subpixel = 46341
    with pytest.raises(N.AdyptError):
        pt.SetConfig(p)
    p.subpixel = 46340
    pt.SetConfig(p)


def _write_tga(path, img):
    """uncompressed 24-bit TGA, top-left origin (one of the formats stb_image — and this loader — reads)"""
    h, w, _ = img.shape
    hdr = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, 24, 0x20])
    with open(path, "wb") as f:
        f.write(hdr + np.ascontiguousarray(img[..., ::-1]).tobytes())


@pytest.mark.parametrize("tw,th", [(1, 1), (3, 5), (64, 64), (257, 2)])
def test_texture_sampling_edges_match_oracle(tw, th, tmp_path, sobol_matrices):
    """FetchInfo's bilinear RGB8 fetch with GL_REPEAT (pathtracer.glsl:88-96; OglScene.cpp:33-38) at its awkward inputs: a 1 x 1 texture,
    sizes that are not powers of two, texture coordinates far outside [0, 1] and negative (the wrap of the texel index and of its
    right / upper neighbour), and texel centres hit exactly.  Viewer type 0 shows the fetched colour directly; the path-traced frames
    then use it as Kd."""
    rs = np.random.RandomState(tw * 131 + th)
    _write_tga(str(tmp_path / "t.tga"), rs.randint(0, 256, size=(th, tw, 3)).astype(np.uint8))
    (tmp_path / "q.mtl").write_text("newmtl tex\nKd 1 1 1\nillum 1\nmap_Kd t.tga\nnewmtl lamp\nKd 0 0 0\nKe 6 5 4\nillum 1\n")
    # a floor quad whose texture coordinates run from -3.25 to 4.5 (and exact multiples of 1 / size along one edge), a lamp above it
    (tmp_path / "q.obj").write_text(
        "mtllib q.mtl\nv 0 0 0\nv 10 0 0\nv 10 0 10\nv 0 0 10\nv 2 6 2\nv 8 6 2\nv 8 6 8\nv 2 6 8\n"
        "vt -3.25 -1.5\nvt 4.5 -1.5\nvt 4.5 2.0\nvt -3.25 2.0\nvn 0 1 0\nvn 0 -1 0\n"
        "usemtl tex\nf 1/1/1 3/3/1 2/2/1\nf 1/1/1 4/4/1 3/3/1\nusemtl lamp\nf 5//2 6//2 7//2\nf 5//2 7//2 8//2\n")
    w, h = 61, 47
    sc, b, pt, p = _tracer(str(tmp_path / "q.obj"), w, h)
    assert len(sc.textures) == 1 and sc.textures[0].shape == (th, tw, 3) and not sc.warnings
    osc = O.Scene(b.nodes, b.tri_indices, sc.triangles, sc.materials, textures=sc.textures)
    ip, iv = api.camera_matrices(70.0, 0.0, -40.0, w, h)
    pos = [5.0, 5.0, 12.0]
    pt.SetCamera(ip, iv, pos)
    P = O.make_params(w, h, pos, ip, iv, stack_size=p.stack_size, max_bounce=p.max_bounce, subpixel=p.subpixel,
                      tmp_life=p.tmp_lifetime, tmin=p.ray_tmin, clamp=p.clamp, sun=list(p.sun))
    pt.Trace(False)                                                          # viewer type 0: the diffuse colour FetchInfo returns
    rgba, _, _ = O.primary_frame(osc, P, 0)
    got = pt.ReadResult()
    assert np.array_equal(bits(got), bits(rgba[..., :3]))
    assert len(np.unique(got.reshape(-1, 3), axis=0)) > (1 if tw * th == 1 else 50)   # the floor really shows the texture
    pt.Trace(True, 5)
    st = O.PathTracerState(w, h)
    O.pt_frames(osc, P, O.shift_bytes(77, w, h), sobol_matrices, st, 5)
    assert np.array_equal(bits(pt.ReadResult()), bits(st.accum[..., :3]))


def write_zoo(tmp_path):
    """The material zoo as OBJ + MTL: every branch of Render's illum switch (pathtracer.glsl:144-201) as patches of one floor under a lamp, with
    a second layer below so that pass-through and refracted paths keep going.  Returns (path of the .obj, the materials)."""
    mats = [("m0", "Kd 0.8 0.7 0.6\nillum 1"), ("m1", "Kd 0.5 0.5 0.5\nKs 0.4 0.4 0.4\nNs 200\nillum 2"), ("m2", "Kd 0.6 0.2 0.2\nKs 0.3 0.3 0.3\nNs 30\nillum 2"),
            ("m3", "Kd 0.2 0.6 0.2\nKs 0.5 0.5 0.5\nNs 31\nillum 2"), ("m4", "Kd 0.1 0.1 0.1\nKs 0.9 0.9 0.9\nNs 100000\nillum 2"), ("m5", "Ks 0.9 0.8 0.7\nillum 3"),
            ("m6", "Ks 1.5 1.5 1.5\nillum 4"), ("m7", "Ks 0.3 0.3 0.9\nillum 5"), ("m8", "Kd 1 1 1\nNi 1.5\nillum 6"), ("m9", "Kd 1 1 1\nNi 0.7\nillum 7"),
            ("m10", "Kd 1 1 1\nNi 1.0\nillum 7"), ("m11", "Kd 1 1 1\nNi 2.4\nillum 7"), ("m12", "Kd 0.9 0.1 0.1\nillum 0"), ("m13", "Kd 0.1 0.9 0.1\nillum 8"),
            ("m14", "Kd 0.1 0.1 0.9\nillum 9"), ("m15", "Kd 0.3 0.3 0.3"), ("m16", "Kd 0.2 0.2 0.2\nKe 2 3 4\nillum 1"), ("m17", "Kd 0.4 0.4 0.4\nKs 0.2 0.2 0.2\nNs 0\nillum 2")]
    (tmp_path / "zoo.mtl").write_text("".join("newmtl %s\n%s\n" % m for m in mats) + "newmtl lamp\nKd 0 0 0\nKe 9 8 7\nillum 1\nnewmtl base\nKd 0.7 0.7 0.7\nillum 1\n")
    v, f = [], []
    def quad(x0, z0, x1, z1, y, mat, up=True):
        i = len(v) + 1
        v.extend([(x0, y, z0), (x1, y, z0), (x1, y, z1), (x0, y, z1)])
        f.append("usemtl %s\nf %d %d %d\nf %d %d %d\n" % ((mat, i, i + 2, i + 1, i, i + 3, i + 2) if up else (mat, i, i + 1, i + 2, i, i + 2, i + 3)))
    for k, (name, _) in enumerate(mats):
        quad(2.0 * (k % 6), 2.0 * (k // 6), 2.0 * (k % 6) + 1.9, 2.0 * (k // 6) + 1.9, 1.0, name)
    quad(-2, -2, 14, 8, 0.0, "base")                 # below the patches: what pass-through / refracted paths reach
    quad(1, 1, 11, 5, 5.0, "lamp", up=False)         # the lamp faces down
    (tmp_path / "zoo.obj").write_text("mtllib zoo.mtl\n" + "".join("v %g %g %g\n" % p for p in v) + "".join(f))
    return str(tmp_path / "zoo.obj"), mats


@pytest.mark.parametrize("env", [
    {},
    {"ADYPT_RARE_MIN": "0"},                                                         # no shading round defers anything
    {"ADYPT_PATH_BLOCKS_PER_CU": "1", "ADYPT_RARE_MIN": "64"},                       # long queues per workgroup: rounds defer, the deferred ring fills up and overflows
    {"ADYPT_PATH_BLOCKS_PER_CU": "1", "ADYPT_RARE_MIN": "3", "ADYPT_SHADE_MIN": "9"},
    {"ADYPT_REF_TRIANGLES_MAX_MB": "0"},                                             # the uTriIndices remap inside the shading round (no per-reference copy of the records)
    {"ADYPT_REF_TRIANGLES_MAX_MB": "0", "ADYPT_PATH_BLOCKS_PER_CU": "1", "ADYPT_RARE_MIN": "1"},
])
def test_material_zoo_matches_oracle(env, tmp_path, sobol_matrices, monkeypatch):
    """Every branch of Render's illum switch (pathtracer.glsl:144-201) on the device: diffuse, glossy above and AT the e = Ns * 0.01 > 0.3
    threshold (falls through to diffuse), very sharp lobes, mirrors, dielectrics with Ni below / at / above 1, the values the switch does
    not name (0, 8, 9: the ray goes straight on), materials without illum (tinyobj default 0), emitters."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)   # (tunables are read at adypt_create)
    obj, mats = write_zoo(tmp_path)
    w, h = 96, 54
    sc, b, pt, p = _tracer(obj, w, h)
    p.max_bounce = 7
    pt.SetConfig(p)
    osc = O.Scene(b.nodes, b.tri_indices, sc.triangles, sc.materials)
    ip, iv = api.camera_matrices(75.0, 0.0, -55.0, w, h)
    pos = [6.0, 4.5, 8.5]
    pt.SetCamera(ip, iv, pos)
    P = O.make_params(w, h, pos, ip, iv, stack_size=p.stack_size, max_bounce=p.max_bounce, subpixel=p.subpixel,
                      tmp_life=p.tmp_lifetime, tmin=p.ray_tmin, clamp=p.clamp, sun=list(p.sun))
    for viewer in (0, 1, 2, 4, 5):
        pt.m_viewer_type = viewer
        pt.Trace(False)
        rgba, _, _ = O.primary_frame(osc, P, viewer)
        assert np.array_equal(bits(pt.ReadResult()), bits(rgba[..., :3])), viewer
    pt.SetInstrumentation(counters=True)
    pt.ResetStats()
    pt.Trace(True, 9)
    st = O.PathTracerState(w, h)
    ost = O.pt_frames(osc, P, O.shift_bytes(77, w, h), sobol_matrices, st, 9).as_dict()
    img = pt.ReadResult()
    assert np.array_equal(bits(img), bits(st.accum[..., :3]))
    g = pt.GetStats()
    assert (g["rays"], g["shaded"], g["bad_materials"]) == (ost["rays"], ost["shaded"], 0)
    tri, _ = pt.ReadHits()
    seen = set(np.unique(np.frombuffer(sc.triangles, dtype=np.uint8).reshape(-1, 100)[tri[tri >= 0]][:, 96:100].copy().view(np.int32)))
    assert len(seen) >= len(mats)                    # the camera really sees every patch
    assert np.isfinite(img).all()


@pytest.mark.parametrize("defer_max", ["64", "24"])   # every such hit of a round is moved (the ring overflows) | the default: rounds full of them keep them
def test_room_of_glossy_and_glass_fills_the_deferred_ring(defer_max, tmp_path, sobol_matrices, monkeypatch):
    """k_path's shading rounds move hits that need the glossy lobe or the dielectric branch to a second ring and shade them in rounds of their own
    (adypt_amd/csrc/device/path.hpp).  In a closed room whose every surface but the lamp is glossy or glass nearly every hit is one of those: the ring is
    full most of the time, what does not fit is shaded where it is found, and with one workgroup per CU the workgroups hold full path tables for most
    of the launch.  Every bit as the oracle computes it, and as the launch-per-bounce pipeline does."""
    mats = "newmtl gl\nKd 0.5 0.4 0.3\nKs 0.4 0.4 0.4\nNs 120\nillum 2\nnewmtl glass\nKd 1 1 1\nNi 1.45\nillum 7\nnewmtl lamp\nKd 0 0 0\nKe 7 6 5\nillum 1\n"
    (tmp_path / "room.mtl").write_text(mats)
    v, f = [], []
    def quad(a, b, c, d, mat):
        i = len(v) + 1
        v.extend([a, b, c, d])
        f.append("usemtl %s\nf %d %d %d\nf %d %d %d\n" % (mat, i, i + 1, i + 2, i, i + 2, i + 3))
    X, Y, Z = 6.0, 4.0, 5.0
    quad((0, 0, 0), (X, 0, 0), (X, 0, Z), (0, 0, Z), "gl"); quad((0, Y, 0), (0, Y, Z), (X, Y, Z), (X, Y, 0), "gl")
    quad((0, 0, 0), (0, Y, 0), (X, Y, 0), (X, 0, 0), "gl"); quad((0, 0, Z), (X, 0, Z), (X, Y, Z), (0, Y, Z), "gl")
    quad((0, 0, 0), (0, 0, Z), (0, Y, Z), (0, Y, 0), "glass"); quad((X, 0, 0), (X, Y, 0), (X, Y, Z), (X, 0, Z), "gl")
    quad((2, Y - 0.01, 2), (4, Y - 0.01, 2), (4, Y - 0.01, 3), (2, Y - 0.01, 3), "lamp")
    quad((2.5, 0.0, 1.5), (3.5, 0.0, 1.5), (3.5, 1.2, 2.0), (2.5, 1.2, 2.0), "glass")
    (tmp_path / "room.obj").write_text("mtllib room.mtl\n" + "".join("v %g %g %g\n" % p for p in v) + "".join(f))
    monkeypatch.setenv("ADYPT_PATH_BLOCKS_PER_CU", "1")
    monkeypatch.setenv("ADYPT_DEFER_MAX", defer_max)
    w, h = 160, 90
    sc, b, pt, p = _tracer(str(tmp_path / "room.obj"), w, h)
    p.max_bounce, p.tmp_lifetime = 8, 4
    pt.SetConfig(p)
    ip, iv = api.camera_matrices(60.0, 200.0, -5.0, w, h)
    pos = [4.5, 2.0, 4.0]
    pt.SetCamera(ip, iv, pos)
    osc = O.Scene(b.nodes, b.tri_indices, sc.triangles, sc.materials)
    P = O.make_params(w, h, pos, ip, iv, stack_size=p.stack_size, max_bounce=p.max_bounce, subpixel=p.subpixel,
                      tmp_life=p.tmp_lifetime, tmin=p.ray_tmin, clamp=p.clamp, sun=list(p.sun))
    pt.SetInstrumentation(counters=True)
    imgs = {}
    for fused in (True, False):
        pt.SetFusedBounces(fused)
        pt.Reset(); pt.ResetStats()
        pt.Trace(True, 6)
        imgs[fused] = (pt.ReadResult(), pt.GetStats())
    st = O.PathTracerState(w, h)
    ost = O.pt_frames(osc, P, O.shift_bytes(77, w, h), sobol_matrices, st, 6).as_dict()
    assert imgs[True][1]["path_rays"] > 0 and imgs[False][1]["path_rays"] == 0
    assert np.array_equal(bits(imgs[True][0]), bits(st.accum[..., :3]))
    assert np.array_equal(bits(imgs[False][0]), bits(st.accum[..., :3]))
    assert (imgs[True][1]["rays"], imgs[True][1]["shaded"]) == (ost["rays"], ost["shaded"])
    assert ost["shaded"] > 5 * w * h   # paths really bounce around in there

