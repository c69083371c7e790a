"""Pins the oracle (oracle/oracle.cpp) against golden vectors produced by the reference's own CPU code
(tests/golden/make_golden.py via oracle/_ref) and against an independent fp64 brute force."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from tests.helpers import GOLDEN, bits, golden_scene, oracle_scene_from_golden


@pytest.mark.parametrize("name", ["tiny0", "tiny1", "tiny2"])
def test_woop_matches_reference_init_triangles(name):
    _, idx, _, tris, _, woop_ref = golden_scene(name)
    w = O.woop_matrices(tris, idx)
    assert np.array_equal(bits(w), bits(woop_ref))


def test_camera_matches_reference_glm():
    cams = json.load(open(os.path.join(GOLDEN, "camera_cases.json")))
    ref = np.fromfile(os.path.join(GOLDEN, "camera_cases.f32"), dtype=np.float32).reshape(len(cams), 32)
    for c, r in zip(cams, ref):
        ip, iv = O.camera(*c)
        assert np.array_equal(bits(ip), bits(r[:16])) and np.array_equal(bits(iv), bits(r[16:]))


def test_sobol_matches_reference_stream(sobol_matrices):
    ref = np.fromfile(os.path.join(GOLDEN, "sobol_points_1000x16.f32"), dtype=np.float32).reshape(1000, 16)
    assert np.array_equal(bits(O.sobol(sobol_matrices, 16, 0, 1000)), bits(ref))
    assert np.array_equal(bits(O.sobol(sobol_matrices, 16, 337, 50)), bits(ref[337:387]))
    assert ref[0, 0] == 0.5  # first point of the gray-code sequence


def test_shift_bytes_are_mt19937_low_bytes():
    ref = np.fromfile(os.path.join(GOLDEN, "shift_bytes_seed0_seed4242.u8"), dtype=np.uint8).reshape(2, 64)
    assert np.array_equal(O.shift_bytes(0, 8, 4).reshape(-1), ref[0])
    assert np.array_equal(O.shift_bytes(4242, 8, 4).reshape(-1), ref[1])
    # known answer of MT19937 with the default seed 5489: first output 3499211612 -> low byte 0x5c
    assert O.shift_bytes(5489, 1, 1).reshape(-1)[0] == (3499211612 & 0xFF)


@pytest.mark.parametrize("name", ["tiny0", "tiny1", "tiny2"])
def test_traversal_known_answers(name):
    kat = np.load(os.path.join(GOLDEN, name + "_kat.npz"))
    sc = oracle_scene_from_golden(name)
    hits = O.trace(sc, kat["rays"], 32)
    assert hits.tobytes() == kat["hits"].tobytes()
    fov, yaw, pitch, px, py, pz = kat["cam"]
    ip, iv = O.camera(fov, yaw, pitch, 64, 36)
    P = O.make_params(64, 36, [px, py, pz], ip, iv, stack_size=32)
    rgba, phits, _ = O.primary_frame(sc, P, 0)
    assert phits.tobytes() == kat["primary_hits"].tobytes()
    assert np.array_equal(bits(rgba), bits(kat["primary_rgba"]))


@pytest.mark.parametrize("name", ["tiny0", "tiny1"])
def test_traversal_agrees_with_fp64_brute_force(name):
    _, _, _, tris, _, _ = golden_scene(name)
    sc = oracle_scene_from_golden(name)
    rs = np.random.RandomState(123)
    p = tris["p"].reshape(-1, 3)
    n = 3000
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(p.min(0) - 1, p.max(0) + 1, size=(n, 3))
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(n, 3))
    hits = O.trace(sc, rays, 32)
    bi, bt = O.brute_force(tris, rays)
    mism = hits["tri_id"] != bi
    # a different id is only acceptable for coplanar duplicates / silhouette ties: same distance
    assert np.all(np.abs(hits["t"][mism].astype(np.float64) - bt[mism]) <= 1e-4 * np.maximum(1.0, np.abs(bt[mism])))
    assert mism.mean() < 0.01
    hit = bi >= 0
    assert np.allclose(hits["t"][hit & ~mism], bt[hit & ~mism], rtol=1e-4, atol=1e-5)


def test_end_to_end_frame_golden(sobol_matrices):
    """G6: 32x18, 4 spp, subpixel 2, tmpLife 2 — exercises the primary-hit cache and sub-pixel cadence."""
    from adypt_amd import scenes
    sc = oracle_scene_from_golden("tiny0")
    cam = scenes._SCENE_TABLE["tiny0"][3]
    ip, iv = O.camera(cam["fov"], cam["yaw"], cam["pitch"], 32, 18)
    P = O.make_params(32, 18, cam["position"], ip, iv, stack_size=16, max_bounce=5, subpixel=2, tmp_life=2, tmin=1e-4, clamp=4.0, sun=(12.0, 11.0, 10.0))
    st = O.PathTracerState(32, 18)
    stats = O.pt_frames(sc, P, O.shift_bytes(4242, 32, 18), sobol_matrices, st, 4)
    ref = np.load(os.path.join(GOLDEN, "tiny0_frame_32x18_4spp.npy"))
    assert np.array_equal(bits(st.accum), bits(ref))
    assert stats.as_dict() == json.load(open(os.path.join(GOLDEN, "tiny0_frame_32x18_4spp.json")))
    # frames rendered one at a time (state carried) give the same image as one call
    st2 = O.PathTracerState(32, 18)
    for _ in range(4):
        O.pt_frames(sc, P, O.shift_bytes(4242, 32, 18), sobol_matrices, st2, 1, n_threads=1)
    assert np.array_equal(bits(st2.accum), bits(ref))
    # clamp is respected, no NaN
    assert np.isfinite(st.accum).all() and st.accum[..., :3].max() <= 4.0


def test_canonical_sincos_pow_close_to_libm():
    """The canonical binary64-series functions are (almost always) the correctly rounded binary32 results."""
    rs = np.random.RandomState(1)
    x = (rs.uniform(0, 1, 200000).astype(np.float32) * np.float32(6.28318530718)).astype(np.float32)
    s, c = O.sincos(x)
    rs64, rc64 = np.sin(x.astype(np.float64)).astype(np.float32), np.cos(x.astype(np.float64)).astype(np.float32)
    assert (bits(s) != bits(rs64)).mean() < 1e-4 and (bits(c) != bits(rc64)).mean() < 1e-4
    assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() < 1e-7
    base = rs.uniform(0, 1, 200000).astype(np.float32)
    for e in (0.0, 0.5, 2.0, 60.0, 400.0):
        y = np.full_like(base, np.float32(1.0) / (np.float32(e) + np.float32(1.0)))
        p = O.pow_(base, y)
        ref = np.power(base.astype(np.float64), y.astype(np.float64)).astype(np.float32)
        assert (bits(p) != bits(ref)).mean() < 1e-4
    # exponent 1 must be the identity (diffuse sampling: pow(1 - r.y, 1/(0+1)))
    assert np.array_equal(bits(O.pow_(base, np.ones_like(base))), bits(base))
    assert np.isnan(O.pow_(np.array([-0.5], np.float32), np.array([0.3], np.float32))[0])
    assert O.pow_(np.array([0.0], np.float32), np.array([0.3], np.float32))[0] == 0.0


def test_display_transform_and_png_writer(tmp_path):
    """shaders/screen.glsl:15-21 restated (gamma 1/2.2 for viewer types <= 3, normalize*0.5+0.5 for 4/5) into RGBA8, and
    the PNG writer of the headless preview: decoded with zlib, CRCs checked."""
    import struct
    import zlib
    from adypt_amd import api
    rs = np.random.RandomState(1)
    img = np.concatenate([rs.uniform(-0.2, 3, (20, 30, 3)).astype(np.float32), np.ones((20, 30, 1), np.float32)], -1)
    img[0, 0, :3] = [0, 1, np.nan]
    img[0, 1, :3] = [np.inf, -np.inf, 0.5]
    d = O.display(img, 3)
    assert d.dtype == np.uint8 and d.shape == (20, 30, 4) and (d[..., 3] == 255).all()
    assert d[0, 0].tolist() == [0, 255, 0, 255] and d[0, 1].tolist() == [255, 0, 186, 255]      # NaN -> 0, +-inf clamp
    ref = np.clip(np.nan_to_num(np.power(np.maximum(img[..., :3].astype(np.float64), 0), 1 / 2.2), nan=0, posinf=1), 0, 1)
    assert np.abs(d[..., :3].astype(int) - np.floor(ref * 255 + 0.5).astype(int)).max() <= 1
    for t in (0, 1, 2):
        assert np.array_equal(O.display(img, t), d)
    n4 = O.display(img[1:], 4)
    nrm = img[1:, :, :3] / np.linalg.norm(img[1:, :, :3], axis=-1, keepdims=True)
    assert np.abs(n4[..., :3].astype(int) - np.floor((nrm * 0.5 + 0.5) * 255 + 0.5).astype(int)).max() <= 1
    assert np.array_equal(O.display(img[1:], 5), n4)
    p = str(tmp_path / "o.png")
    api.save_png(p, d)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, = struct.unpack(">I", raw[pos:pos + 4])
        t, data = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(t + data) & 0xFFFFFFFF == crc
        chunks.append((t, data))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (w, h, depth, ctype) == (30, 20, 8, 6)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, w * 4 + 1)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 4), d)


def test_pow_of_exponent_one_is_the_identity_in_the_canonical_series():
    """canon_pow(x, 1) returns x without running its series (kernel and oracle alike): legitimate only because the series
    itself returns exactly x for every positive finite binary32 x — checked here on 4 M random bit patterns, every
    exponent's extremes, the denormals and the values the diffuse lobe feeds it (1 - r.y, r.y = k / 2^24)."""
    rs = np.random.RandomState(11)
    bits_ = rs.randint(1, 0x7f800000, size=4_000_000, dtype=np.int64).astype(np.uint32)
    edge = np.array([1, 2, 3, 0x007fffff, 0x00800000, 0x00800001, 0x3f7fffff, 0x3f800000, 0x3f800001, 0x7f7fffff, 0x7f7ffffe], np.uint32)
    expo = (np.arange(1, 255, dtype=np.uint32) << 23)
    lobe = (1.0 - np.arange(0, 1 << 24, 4099, dtype=np.float64) / (1 << 24)).astype(np.float32).view(np.uint32)
    x = np.concatenate([bits_, edge, expo, expo | 0x7fffff, expo | 1, lobe]).view(np.float32)
    one = np.ones_like(x)
    assert np.array_equal(O.pow_(x, one, series_only=True).view(np.uint32), x.view(np.uint32))
    assert np.array_equal(O.pow_(x, one).view(np.uint32), x.view(np.uint32))
    # outside the shortcut's domain both entry points still run the series
    bad = np.array([-1.0, -0.0, 0.0, np.inf, np.nan], np.float32)
    a, b = O.pow_(bad, np.ones_like(bad)), O.pow_(bad, np.ones_like(bad), series_only=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_sun_visibility_option_of_the_oracle(sobol_matrices):
    """The occlusion query the reference has commented out (pathtracer.glsl:132) as an option of the restatement: off is the
    reference as it runs (the committed golden frame), on only ever removes sun light and traces extra (any-hit) rays."""
    from tests.helpers import oracle_scene_from_golden
    import json
    sc = oracle_scene_from_golden("tiny0")
    cam = np.load(os.path.join(GOLDEN, "tiny0_kat.npz"))["cam"]
    ip, iv = O.camera(float(cam[0]), float(cam[1]), float(cam[2]), 32, 18)
    kw = dict(stack_size=16, max_bounce=5, subpixel=2, tmp_life=2, tmin=1e-4, clamp=4.0, sun=[12.0, 11.0, 10.0])
    imgs, rays = {}, {}
    for on in (False, True):
        P = O.make_params(32, 18, [float(x) for x in cam[3:6]], ip, iv, sun_visibility=on, **kw)
        st = O.PathTracerState(32, 18)
        s = O.pt_frames(sc, P, O.shift_bytes(4242, 32, 18), sobol_matrices, st, 4).as_dict()
        imgs[on], rays[on] = st.accum.copy(), s["rays"]
    ref = np.load(os.path.join(GOLDEN, "tiny0_frame_32x18_4spp.npy"))
    assert np.array_equal(imgs[False].view(np.uint32), ref.view(np.uint32))
    assert rays[True] > rays[False] and (imgs[True][..., :3] <= imgs[False][..., :3]).all()
    assert (imgs[True][..., :3] < imgs[False][..., :3]).any()


def test_unorm8_decode_formula():
    """The device decodes c / 255 as q = c * rn(1/255); q + rn(c - 255 q) * rn(1/255) with fused steps (csrc/device/canon_math.hpp:
    unorm8_to_float) where the oracle divides.  Exact rational arithmetic, every rounding to the nearest binary32: identical for all 256 c."""
    from fractions import Fraction

    def rn32(fr):
        f = np.float32(float(fr))
        cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
        return np.float32(min(cands, key=lambda x: (abs(Fraction(float(x)) - fr), int(np.float32(x).view(np.uint32)) & 1)))

    r = Fraction(float.fromhex("0x1.010102p-8"))
    assert rn32(Fraction(1, 255)) == np.float32(float(r))
    for c in range(256):
        q = Fraction(float(rn32(c * r)))
        e = Fraction(float(rn32(c - 255 * q)))       # fmaf(-q, 255, c)
        got = rn32(q + e * r)                           # fmaf(e, r, q)
        assert got == np.float32(c) / np.float32(255.0), c
