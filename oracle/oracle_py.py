"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement of Adypt's GPU hot path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
package (adypt_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "adypt_ref")

NODE_DT = np.dtype([("p", "<f4", 3), ("e", "u1", 3), ("imask", "u1"), ("child_base", "<u4"), ("tri_base", "<u4"),
                    ("meta", "u1", 8), ("qlox", "u1", 8), ("qloy", "u1", 8), ("qloz", "u1", 8),
                    ("qhix", "u1", 8), ("qhiy", "u1", 8), ("qhiz", "u1", 8)])
TRI_DT = np.dtype([("p", "<f4", (3, 3)), ("n", "<f4", (3, 3)), ("tc", "<f4", (3, 2)), ("matid", "<i4")])
MAT_DT = np.dtype([("dtex", "<i4"), ("kd", "<f4", 3), ("etex", "<i4"), ("ke", "<f4", 3), ("stex", "<i4"),
                   ("ks", "<f4", 3), ("illum", "<i4"), ("shininess", "<f4"), ("dissolve", "<f4"), ("ior", "<f4")])
HIT_DT = np.dtype([("ref_idx", "<i4"), ("tri_id", "<i4"), ("u", "<f4"), ("v", "<f4"), ("t", "<f4"),
                   ("nodes", "<u4"), ("tris", "<u4"), ("hash", "<u4"), ("max_depth", "<u4")])
assert NODE_DT.itemsize == 80 and TRI_DT.itemsize == 100 and MAT_DT.itemsize == 64 and HIT_DT.itemsize == 36


class _Texture(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("rgb", C.c_void_p)]


class _Scene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_int64), ("woop", C.c_void_p), ("tri_indices", C.c_void_p),
                ("n_refs", C.c_int64), ("triangles", C.c_void_p), ("n_tris", C.c_int64), ("materials", C.c_void_p),
                ("n_mats", C.c_int64), ("textures", C.c_void_p), ("n_tex", C.c_int64)]


class _Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stack_size", C.c_int32), ("max_bounce", C.c_int32),
                ("subpixel", C.c_int32), ("tmp_life", C.c_int32), ("tmin", C.c_float), ("clamp", C.c_float),
                ("sun", C.c_float * 3), ("origin", C.c_float * 3), ("inv_proj", C.c_float * 16),
                ("inv_view", C.c_float * 16), ("sun_visibility", C.c_int32), ("sun_dir", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("hits", C.c_uint64),
                ("shaded", C.c_uint64), ("texel_fetches", C.c_uint64), ("stack_overflows", C.c_uint64),
                ("max_depth", C.c_uint32), ("pad", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "pad"}


def build(force: bool = False) -> None:
    """(Re)build liboracle.so — and oracle/_ref when /root/reference is present (build container only)."""
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_BIN)):
        subprocess.check_call(["make", "-C", _HERE, "_ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_abi_version.restype = C.c_int
        assert _lib.orc_abi_version() == 2
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_threads() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box
    shows all 256 hardware threads in the mask but grants a 16-core share through cpu.max)."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


# ---------------------------------------------------------------------------------------------------
# file formats of the reference's CPU side
# ---------------------------------------------------------------------------------------------------
def load_bvh_file(path: str):
    """`.bvh` cache: src/BVH/WideBVH.cpp:9-23.  Returns (cfg tuple, tri_indices int32[], nodes NODE_DT[])."""
    raw = open(path, "rb").read()
    assert raw[:10] == b"CWBVH_1.0\0", "bad .bvh magic"
    cfg = np.frombuffer(raw, dtype=np.dtype([("depth", "<i4"), ("tri", "<f4"), ("node", "<f4")]), count=1, offset=10)[0]
    n = int(np.frombuffer(raw, dtype="<u4", count=1, offset=22)[0])
    idx = np.frombuffer(raw, dtype="<i4", count=n, offset=26).copy()
    nodes = np.frombuffer(raw, dtype=NODE_DT, offset=26 + 4 * n).copy()
    return (int(cfg["depth"]), float(cfg["tri"]), float(cfg["node"])), idx, nodes


class Scene:
    """Flat arrays the tracer consumes (SURVEY.md §2 resource table)."""

    def __init__(self, nodes, tri_indices, triangles, materials, woop=None, textures: Sequence[np.ndarray] = ()):
        self.nodes = np.ascontiguousarray(nodes).view(NODE_DT).reshape(-1)
        self.tri_indices = np.ascontiguousarray(tri_indices, dtype=np.int32)
        self.triangles = np.ascontiguousarray(triangles).view(TRI_DT).reshape(-1)
        self.materials = np.ascontiguousarray(materials).view(MAT_DT).reshape(-1)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]  # each (h, w, 3)
        self.woop = woop_matrices(self.triangles, self.tri_indices) if woop is None else np.ascontiguousarray(woop, dtype=np.float32)
        self._tex_arr = (_Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex_arr[i].w, self._tex_arr[i].h = t.shape[1], t.shape[0]
            self._tex_arr[i].rgb = t.ctypes.data
        s = _Scene()
        s.nodes, s.n_nodes = self.nodes.ctypes.data, len(self.nodes)
        s.woop, s.tri_indices, s.n_refs = self.woop.ctypes.data, self.tri_indices.ctypes.data, len(self.tri_indices)
        s.triangles, s.n_tris = self.triangles.ctypes.data, len(self.triangles)
        s.materials, s.n_mats = self.materials.ctypes.data, len(self.materials)
        s.textures, s.n_tex = C.addressof(self._tex_arr), len(self.textures)
        self._c = s


def make_params(width, height, origin, inv_proj, inv_view, *, stack_size=24, max_bounce=8, subpixel=8, tmp_life=16,
                tmin=1e-4, clamp=4.0, sun=(12.0, 11.0, 10.0), sun_visibility=False, sun_dir=(0.6, 1.0, 0.2)) -> _Params:
    p = _Params()
    p.width, p.height, p.stack_size, p.max_bounce, p.subpixel, p.tmp_life = width, height, stack_size, max_bounce, subpixel, tmp_life
    p.tmin, p.clamp = tmin, clamp
    p.sun[:] = [float(x) for x in sun]
    p.sun_visibility = 1 if sun_visibility else 0
    p.sun_dir[:] = [float(x) for x in sun_dir]
    p.origin[:] = [float(x) for x in origin]
    p.inv_proj[:] = [float(x) for x in np.asarray(inv_proj, dtype=np.float32).reshape(16)]
    p.inv_view[:] = [float(x) for x in np.asarray(inv_view, dtype=np.float32).reshape(16)]
    return p


# ---------------------------------------------------------------------------------------------------
# restated functions
# ---------------------------------------------------------------------------------------------------
def woop_matrices(triangles: np.ndarray, tri_indices: np.ndarray) -> np.ndarray:
    tri = np.ascontiguousarray(triangles)
    idx = np.ascontiguousarray(tri_indices, dtype=np.int32)
    out = np.empty((len(idx), 12), dtype=np.float32)
    lib().orc_woop(_p(tri), _p(idx), C.c_int64(len(idx)), _p(out))
    return out


def camera(fov: float, yaw: float, pitch: float, width: int, height: int) -> Tuple[np.ndarray, np.ndarray]:
    ip = np.empty(16, dtype=np.float32)
    iv = np.empty(16, dtype=np.float32)
    lib().orc_camera(C.c_float(fov), C.c_float(yaw), C.c_float(pitch), C.c_int(width), C.c_int(height), _p(ip), _p(iv))
    return ip, iv


def mat4_inverse(m: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.empty(16, dtype=np.float32)
    lib().orc_mat4_inverse(_p(m), _p(out))
    return out


def sobol(matrices: np.ndarray, dim: int, first: int, n: int) -> np.ndarray:
    m = np.ascontiguousarray(matrices, dtype=np.uint32)
    assert m.shape[0] >= dim and m.shape[1] == 32
    out = np.empty((n, dim), dtype=np.float32)
    lib().orc_sobol(_p(m), C.c_int(dim), C.c_int(first), C.c_int(n), _p(out))
    return out


def shift_bytes(seed: int, width: int, height: int) -> np.ndarray:
    out = np.empty((height, width, 2), dtype=np.uint8)
    lib().orc_shift_bytes(C.c_uint32(seed), C.c_int(width), C.c_int(height), _p(out))
    return out


def sincos(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = np.empty_like(x)
    c = np.empty_like(x)
    lib().orc_sincos(_p(x), C.c_int(x.size), _p(s), _p(c))
    return s, c


def pow_(x: np.ndarray, y: np.ndarray, series_only: bool = False) -> np.ndarray:
    """canonical pow; series_only skips the x^1 = x shortcut (lets the tests check that the series returns x there)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    (lib().orc_pow_series if series_only else lib().orc_pow)(_p(x), _p(y), C.c_int(x.size), _p(out))
    return out


def trace(scene: Scene, rays: np.ndarray, stack_size: int = 24, n_threads: Optional[int] = None, any_hit: bool = False) -> np.ndarray:
    """rays: (n, 8) float32 = ox oy oz tmin dx dy dz pad.  Returns HIT_DT[n].  any_hit: traversal.glsl:257-494."""
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    hits = np.empty(len(rays), dtype=HIT_DT)
    fn = lib().orc_trace_any if any_hit else lib().orc_trace
    fn(C.byref(scene._c), C.c_int(stack_size), _p(rays), C.c_int64(len(rays)), _p(hits), C.c_int(n_threads or default_threads()))
    return hits


def display(rgba: np.ndarray, viewer_type: int) -> np.ndarray:
    """shaders/screen.glsl:15-21 into an RGBA8 colour buffer: H x W x 4 float32 -> H x W x 4 uint8."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    assert rgba.shape[-1] == 4
    out = np.empty(rgba.shape, dtype=np.uint8)
    lib().orc_display(_p(rgba), C.c_int64(rgba.size // 4), C.c_int(viewer_type), _p(out))
    return out


def primary_frame(scene: Scene, params: _Params, viewer_type: int = 0, n_threads: Optional[int] = None):
    w, h = params.width, params.height
    rgba = np.empty((h, w, 4), dtype=np.float32)
    hits = np.empty((h, w), dtype=HIT_DT)
    st = Stats()
    lib().orc_primary_frame(C.byref(scene._c), C.byref(params), C.c_int(viewer_type), _p(rgba), _p(hits), C.byref(st),
                            C.c_int(n_threads or default_threads()))
    return rgba, hits, st


class PathTracerState:
    """accumulated image + primary-hit cache of the oracle path tracer (images 0 and 1 of the reference)."""

    def __init__(self, width: int, height: int):
        self.accum = np.zeros((height, width, 4), dtype=np.float32)
        self.cache_tri = np.full((height, width), -1, dtype=np.int32)
        self.cache_uv = np.zeros((height, width, 2), dtype=np.float32)
        self.spp = 0


def pt_frames(scene: Scene, params: _Params, shift: np.ndarray, sobol_matrices: np.ndarray, state: PathTracerState,
              n_spp: int, mask: Optional[np.ndarray] = None, n_threads: Optional[int] = None) -> Stats:
    dim = 2 * params.max_bounce
    pts = sobol(sobol_matrices, dim, state.spp, n_spp)
    shift = np.ascontiguousarray(shift, dtype=np.uint8)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    st = Stats()
    lib().orc_pt_frames(C.byref(scene._c), C.byref(params), _p(shift), _p(pts), C.c_int(state.spp), C.c_int(n_spp),
                        _p(state.accum), _p(state.cache_tri), _p(state.cache_uv), _p(mask), C.byref(st),
                        C.c_int(n_threads or default_threads()))
    state.spp += n_spp
    return st


def sample_hemisphere(r: np.ndarray, e: float) -> np.ndarray:
    """SampleHemisphere (pathtracer.glsl:52-64) for (r.x, r.y) pairs in [0,1) — test hook."""
    r = np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 2)
    out = np.empty((len(r), 3), dtype=np.float32)
    lib().orc_sample_hemisphere(_p(r), C.c_int(len(r)), C.c_float(e), _p(out))
    return out


def align_direction(direction: np.ndarray, target: np.ndarray) -> np.ndarray:
    """AlignDirection (pathtracer.glsl:66-71) — test hook."""
    d = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(target, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(d)
    lib().orc_align_direction(_p(d), _p(t), C.c_int(len(d)), _p(out))
    return out


def scatter(materials: np.ndarray, normal: np.ndarray, dir_in: np.ndarray, r: np.ndarray) -> np.ndarray:
    """One material response (pathtracer.glsl:141-201) per record — test hook.  Returns n x 8: new direction, throughput factor,
    Fresnel term (-1 unless dielectric), alive."""
    m = np.ascontiguousarray(materials)
    assert m.dtype == MAT_DT
    nrm = np.ascontiguousarray(normal, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dir_in, dtype=np.float32).reshape(-1, 3)
    rr = np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 2)
    assert len(m) == len(nrm) == len(d) == len(rr)
    out = np.empty((len(m), 8), dtype=np.float32)
    lib().orc_scatter(_p(m), _p(nrm), _p(d), _p(rr), C.c_int(len(m)), _p(out))
    return out


def brute_force(triangles: np.ndarray, rays: np.ndarray, n_threads: Optional[int] = None):
    tri = np.ascontiguousarray(triangles)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    ids = np.empty(len(rays), dtype=np.int32)
    t = np.empty(len(rays), dtype=np.float64)
    lib().orc_brute_force(_p(tri), C.c_int64(len(tri)), _p(rays), C.c_int64(len(rays)), _p(ids), _p(t),
                          C.c_int(n_threads or default_threads()))
    return ids, t


# ---------------------------------------------------------------------------------------------------
# reference binary (build container only)
# ---------------------------------------------------------------------------------------------------
def have_ref() -> bool:
    return os.path.exists(REF_BIN)


def ref(*args: str, capture: bool = False) -> subprocess.CompletedProcess:
    return subprocess.run([REF_BIN, *map(str, args)], stdout=subprocess.PIPE if capture else subprocess.DEVNULL,
                          stderr=subprocess.PIPE, check=False)
