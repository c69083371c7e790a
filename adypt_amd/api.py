"""Host-side mirror of the reference's interface for the hot path, on top of the C-ABI.

Names and call order follow the reference so that tests read like its own call sites:

    InstanceConfig.LoadFromFile / GetJson / SaveToFile        src/InstanceConfig.cpp:10-210
    Scene.LoadFromFile                                        src/Util/Scene.cpp:9-136
    WideBVH.LoadFromFile / SaveToFile, build_bvh()            src/BVH/WideBVH.cpp:9-66, src/Instance.cpp:20-32
    Camera.GetInvProjection / GetInvView                      src/Tracer/Camera.cpp:13-23 (+ inverses of SetCamera)
    HipScene.Initialize(scene, bvh)                           OglScene::Initialize, src/Tracer/OglScene.hpp:43
    HipPathTracer.Initialize / SetCamera / Trace / SaveResult / GetSPP
                                                              OglPathTracer, src/Tracer/OglPathTracer.hpp:66-82
    Instance.InitializeFromFile / Update                      src/Instance.cpp:10-69

Everything numeric happens in libadypt_hip.so; this module only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N

NODE_BYTES, TRI_BYTES, MAT_BYTES = 80, 100, 64
HIT_DT = np.dtype([("ref_idx", "<i4"), ("tri_id", "<i4"), ("u", "<f4"), ("v", "<f4"), ("t", "<f4"),
                   ("nodes", "<u4"), ("tris", "<u4"), ("hash", "<u4"), ("max_depth", "<u4")])


def _bytes_view(ptr: int, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.frombuffer((C.c_char * n).from_address(ptr), dtype=np.uint8).copy()


# ---------------------------------------------------------------------------------------------------------------
class InstanceConfig:
    """`.config` JSON <-> struct (src/InstanceConfig.hpp:12-48)."""

    def __init__(self) -> None:
        self.c = N.Config()
        self.SetDefault()

    def SetDefault(self) -> None:
        N.lib.adypt_config_default(C.byref(self.c))

    def LoadFromFile(self, filename: str) -> bool:
        return N.lib.adypt_config_load(filename.encode(), C.byref(self.c)) == N.ADYPT_OK

    def Parse(self, text: str) -> bool:
        return N.lib.adypt_config_parse(text.encode(), C.byref(self.c)) == N.ADYPT_OK

    def GetJson(self) -> str:
        n = N.lib.adypt_config_json(C.byref(self.c), None, 0)
        buf = C.create_string_buffer(n)
        N.lib.adypt_config_json(C.byref(self.c), buf, n)
        return buf.value.decode()

    def SaveToFile(self, filename: str) -> bool:
        return N.lib.adypt_config_save(filename.encode(), C.byref(self.c)) == N.ADYPT_OK

    @staticmethod
    def last_error() -> str:
        return (N.lib.adypt_host_last_error() or b"").decode()

    # convenient attribute access with the reference's member names
    m_width = property(lambda s: s.c.width)
    m_height = property(lambda s: s.c.height)
    m_obj_filename = property(lambda s: s.c.obj_filename.decode())
    m_bvh_filename = property(lambda s: s.c.bvh_filename.decode())

    def pt_params(self, shift_seed: int = 0) -> N.PtParams:
        p = N.PtParams()
        p.stack_size, p.max_bounce, p.subpixel, p.tmp_lifetime = self.c.stack_size, self.c.max_bounce, self.c.subpixel, self.c.tmp_lifetime
        p.ray_tmin, p.clamp = self.c.ray_tmin, self.c.clamp
        p.sun[:] = list(self.c.sun)
        p.shift_seed = shift_seed
        return p

    def bvh_params(self) -> N.BvhParams:
        return N.BvhParams(self.c.bvh.max_spatial_depth, self.c.bvh.triangle_sah, self.c.bvh.node_sah)


class Scene:
    """Triangle[] + materials + decoded textures of an OBJ file."""

    def __init__(self) -> None:
        self._h = C.c_void_p()
        self.triangles = np.zeros(0, dtype=np.uint8)
        self.materials = np.zeros(0, dtype=np.uint8)
        self.textures: List[np.ndarray] = []
        self.warnings = ""

    def LoadFromFile(self, filename: str) -> bool:
        self._free()
        if N.lib.adypt_scene_load(filename.encode(), C.byref(self._h)) != N.ADYPT_OK:
            return False
        self._pull()
        return True

    @classmethod
    def FromArrays(cls, triangles: np.ndarray, materials: np.ndarray) -> "Scene":
        s = cls()
        t = np.ascontiguousarray(triangles).view(np.uint8).reshape(-1)
        m = np.ascontiguousarray(materials).view(np.uint8).reshape(-1)
        N.check_host(N.lib.adypt_scene_from_arrays(t.ctypes.data, len(t) // TRI_BYTES, m.ctypes.data, len(m) // MAT_BYTES, C.byref(s._h)))
        s._pull()
        return s

    def _pull(self) -> None:
        p = C.c_void_p()
        n = N.lib.adypt_scene_triangles(self._h, C.byref(p))
        self.triangles = _bytes_view(p.value, n * TRI_BYTES)
        n = N.lib.adypt_scene_materials(self._h, C.byref(p))
        self.materials = _bytes_view(p.value, n * MAT_BYTES)
        nt = N.lib.adypt_scene_textures(self._h, C.byref(p))
        self.textures = []
        if nt:
            arr = (N.Texture * nt).from_address(p.value)
            for t in arr:
                self.textures.append(_bytes_view(t.rgb, t.width * t.height * 3).reshape(t.height, t.width, 3))
        w = N.lib.adypt_scene_warnings(self._h)
        self.warnings = (w or b"").decode("utf-8", "replace")  # textures that could not be decoded (their materials render black)

    def GetTriangles(self) -> np.ndarray:
        return self.triangles

    def GetMaterials(self) -> np.ndarray:
        return self.materials

    def GetAABB(self) -> Tuple[np.ndarray, np.ndarray]:
        lo = np.zeros(3, dtype=np.float32)
        hi = np.zeros(3, dtype=np.float32)
        N.lib.adypt_scene_aabb(self._h, lo.ctypes.data, hi.ctypes.data)
        return lo, hi

    @property
    def n_tris(self) -> int:
        return len(self.triangles) // TRI_BYTES

    def _free(self) -> None:
        if self._h:
            N.lib.adypt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self._free()
        except Exception:
            pass


class WideBVH:
    """80-byte CWBVH8 nodes + reference->triangle indices (src/BVH/WideBVH.hpp:28-43)."""

    def __init__(self) -> None:
        self._h = C.c_void_p()
        self.nodes = np.zeros(0, dtype=np.uint8)
        self.tri_indices = np.zeros(0, dtype=np.int32)
        self.build_info: Optional[N.BuildInfo] = None

    def _pull(self) -> None:
        p = C.c_void_p()
        n = N.lib.adypt_bvh_nodes(self._h, C.byref(p))
        self.nodes = _bytes_view(p.value, n * NODE_BYTES)
        n = N.lib.adypt_bvh_tri_indices(self._h, C.byref(p))
        self.tri_indices = _bytes_view(p.value, n * 4).view(np.int32)

    def LoadFromFile(self, filename: str, expected: N.BvhParams) -> bool:
        self._free()
        if N.lib.adypt_bvh_load(filename.encode(), C.byref(expected), C.byref(self._h)) != N.ADYPT_OK:
            return False
        self._pull()
        return True

    def SaveToFile(self, filename: str, cfg: N.BvhParams) -> bool:
        return N.lib.adypt_bvh_save(self._h, filename.encode(), C.byref(cfg)) == N.ADYPT_OK

    def Build(self, scene: Scene, cfg: N.BvhParams) -> None:
        """SBVHBuilder{cfg,&sbvh,scene}.Run(); WideBVHBuilder{cfg,&wbvh,sbvh}.Run() (src/Instance.cpp:24-26)."""
        self._free()
        info = N.BuildInfo()
        N.check_host(N.lib.adypt_bvh_build(scene._h, C.byref(cfg), C.byref(self._h), C.byref(info)))
        self.build_info = info
        self._pull()

    def GetNodes(self) -> np.ndarray:
        return self.nodes

    def GetTriIndices(self) -> np.ndarray:
        return self.tri_indices

    def _free(self) -> None:
        if self._h:
            N.lib.adypt_bvh_free(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self._free()
        except Exception:
            pass


TEST_HOOKS_MAGIC = 0x7465737468303031  # ADYPT_TEST_HOOKS_MAGIC (include/adypt_hip.h)


def enable_test_hooks() -> None:
    """TEST-ONLY (adypt_enable_test_hooks): makes the library honour ADYPT_MULTI_SHARED_DEVICE, ADYPT_COMM_TRANSPORT=host, ADYPT_AUDIT_SELFTEST and
    ADYPT_GATHER_STALL_TEST for the rest of this process.  tests/conftest.py, tools/comm_world.py and `bench.py --rehearsal` call it; nothing else does."""
    r = N.lib.adypt_enable_test_hooks(TEST_HOOKS_MAGIC)
    if r != N.ADYPT_OK:
        raise RuntimeError("adypt_enable_test_hooks refused")


def woop_matrices(triangles: np.ndarray, tri_indices: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(triangles).view(np.uint8).reshape(-1)
    idx = np.ascontiguousarray(tri_indices, dtype=np.int32)
    out = np.empty((len(idx), 12), dtype=np.float32)
    N.lib.adypt_woop_matrices(t.ctypes.data, idx.ctypes.data, len(idx), out.ctypes.data)
    return out


def sobol_points(dim: int, first: int, n: int) -> np.ndarray:
    out = np.empty((n, dim), dtype=np.float32)
    N.check_host(N.lib.adypt_sobol_points(dim, first, n, out.ctypes.data))
    return out


def shift_bytes(seed: int, width: int, height: int) -> np.ndarray:
    out = np.empty((height, width, 2), dtype=np.uint8)
    N.lib.adypt_shift_bytes(seed, width, height, out.ctypes.data)
    return out


def save_exr(path: str, rgb: np.ndarray, save_as_fp16: bool = False) -> None:
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    N.check_host(N.lib.adypt_save_exr(path.encode(), rgb.ctypes.data, w, h, 1 if save_as_fp16 else 0))


def save_png(path: str, rgba8: np.ndarray) -> None:
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = rgba8.shape[:2]
    assert rgba8.shape == (h, w, 4)
    N.check_host(N.lib.adypt_save_png(path.encode(), rgba8.ctypes.data, w, h))


def load_image_rgb8(path: str) -> np.ndarray:
    """stbi_load(path, &w, &h, &c, 3) of OglScene::load_texture: H x W x 3 uint8, row 0 = top."""
    p, w, h = C.c_void_p(), C.c_int32(), C.c_int32()
    N.check_host(N.lib.adypt_load_image_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.array(_bytes_view(p.value, w.value * h.value * 3).reshape(h.value, w.value, 3))
    finally:
        N.lib.adypt_free(p)


def load_exr(path: str) -> np.ndarray:
    p = C.c_void_p()
    w = C.c_int()
    h = C.c_int()
    N.check_host(N.lib.adypt_load_exr(path.encode(), C.byref(p), C.byref(w), C.byref(h)))
    out = np.frombuffer((C.c_float * (w.value * h.value * 3)).from_address(p.value), dtype=np.float32).copy().reshape(h.value, w.value, 3)
    N.lib.adypt_free(p)
    return out


class Camera:
    """Camera::Initialize / GetView / GetProjection / Control (src/Tracer/Camera.hpp:14-37)."""

    def Initialize(self, cfg: InstanceConfig, width: int, height: int) -> None:
        self.cfg, self.width, self.height = cfg, width, height

    def matrices(self) -> Tuple[np.ndarray, np.ndarray]:
        ip = np.empty(16, dtype=np.float32)
        iv = np.empty(16, dtype=np.float32)
        c = self.cfg.c
        N.lib.adypt_camera_matrices(c.fov, c.yaw, c.pitch, self.width, self.height, ip.ctypes.data, iv.ctypes.data)
        return ip, iv

    @property
    def position(self) -> np.ndarray:
        return np.array(list(self.cfg.c.position), dtype=np.float32)

    KEY_W, KEY_A, KEY_S, KEY_D, KEY_SPACE, KEY_LEFT_SHIFT = 1, 2, 4, 8, 16, 32

    def Control(self, keys: int = 0, mouse_dx: float = 0.0, mouse_dy: float = 0.0, frame_seconds: float = 0.0) -> None:
        """Camera::Control (src/Tracer/Camera.cpp:25-59) with the key / mouse state passed in instead of read from GLFW."""
        N.lib.adypt_camera_control(C.byref(self.cfg.c), keys, mouse_dx, mouse_dy, frame_seconds)


def camera_matrices(fov: float, yaw: float, pitch: float, width: int, height: int) -> Tuple[np.ndarray, np.ndarray]:
    ip = np.empty(16, dtype=np.float32)
    iv = np.empty(16, dtype=np.float32)
    N.lib.adypt_camera_matrices(fov, yaw, pitch, width, height, ip.ctypes.data, iv.ctypes.data)
    return ip, iv


def device_free_bytes(device: int) -> Optional[int]:
    """Free memory of HIP device `device` (hipMemGetInfo through the runtime libadypt_hip.so is linked against), None when the runtime does not answer.
    Measurement scripts only (bench.py's self-check asks before it creates a second context of a large scene); the library itself never asks."""
    try:
        hip = C.CDLL("libamdhip64.so")
        free, total = C.c_size_t(0), C.c_size_t(0)
        if hip.hipSetDevice(C.c_int(device)) != 0 or hip.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
            return None
        return int(free.value)
    except OSError:
        return None


class HipScene:
    """OglScene: the flat arrays the kernels consume.  Initialize(scene, bvh) only records host arrays; the upload
    to HBM happens in HipPathTracer.Initialize (adypt_create) because the C-ABI creates scene + images together."""

    def Initialize(self, scene: Scene, bvh: WideBVH, woop: Optional[np.ndarray] = None) -> None:
        self.scene, self.bvh = scene, bvh
        self.woop = None if woop is None else np.ascontiguousarray(woop, dtype=np.float32)

    def GetTextures(self) -> Sequence[np.ndarray]:
        return self.scene.textures


class ViewerTypes:
    kDiffuse, kSpecular, kEmissive, kPTRadiance, kNormal, kPosition = range(6)


class HipPathTracer:
    """OglPathTracer (src/Tracer/OglPathTracer.hpp:66-82) on HIP."""

    def __init__(self) -> None:
        self._ctx = C.c_void_p()
        self.m_viewer_type = ViewerTypes.kDiffuse

    def Initialize(self, config: N.PtParams, scene: HipScene, width: int, height: int, device: int = 0,
                   tile_rank: int = 0, tile_nranks: int = 1) -> None:
        self.destroy()
        self.width, self.height = width, height
        self.tile_rank, self.tile_nranks = tile_rank, tile_nranks
        s, b = scene.scene, scene.bvh
        tex_arr = (N.Texture * max(1, len(s.textures)))()
        self._keep = [s.triangles, s.materials, b.nodes, b.tri_indices, scene.woop, tex_arr] + list(s.textures)
        for i, t in enumerate(s.textures):
            tex_arr[i].width, tex_arr[i].height, tex_arr[i].rgb = t.shape[1], t.shape[0], t.ctypes.data
        d = N.SceneDesc()
        d.nodes, d.n_nodes = b.nodes.ctypes.data, len(b.nodes) // NODE_BYTES
        d.tri_indices, d.n_refs = b.tri_indices.ctypes.data, len(b.tri_indices)
        d.woop = None if scene.woop is None else scene.woop.ctypes.data
        d.triangles, d.n_tris = s.triangles.ctypes.data, len(s.triangles) // TRI_BYTES
        d.materials, d.n_mats = (s.materials.ctypes.data if len(s.materials) else None), len(s.materials) // MAT_BYTES
        d.textures, d.n_textures = (C.addressof(tex_arr) if s.textures else None), len(s.textures)
        d.width, d.height, d.device, d.tile_rank, d.tile_nranks = width, height, device, tile_rank, tile_nranks
        N.check(N.lib.adypt_create(C.byref(self._ctx), C.byref(d)))
        self.SetConfig(config)

    def SetConfig(self, config: N.PtParams) -> None:
        N.check(N.lib.adypt_set_params(self._ctx, C.byref(config)), self._ctx)

    def SetCamera(self, inv_projection: np.ndarray, inv_view: np.ndarray, position) -> None:
        ip = np.ascontiguousarray(inv_projection, dtype=np.float32).reshape(16)
        iv = np.ascontiguousarray(inv_view, dtype=np.float32).reshape(16)
        pos = np.ascontiguousarray(position, dtype=np.float32).reshape(3)
        N.check(N.lib.adypt_set_camera(self._ctx, pos.ctypes.data, ip.ctypes.data, iv.ctypes.data), self._ctx)

    def Trace(self, enable_pt: bool, n_spp: int = 1) -> None:
        """Trace(true): n_spp more frames; Trace(false): one primary-ray viewer frame of m_viewer_type."""
        if enable_pt:
            self.m_viewer_type = ViewerTypes.kPTRadiance
            N.check(N.lib.adypt_trace_spp(self._ctx, n_spp), self._ctx)
        else:
            if self.m_viewer_type == ViewerTypes.kPTRadiance:
                self.m_viewer_type = ViewerTypes.kDiffuse
            N.check(N.lib.adypt_trace_primary(self._ctx, self.m_viewer_type), self._ctx)

    def SetSunVisibility(self, enabled: bool, direction=None) -> None:
        """The occlusion query the reference has commented out (pathtracer.glsl:132); off = the reference as it runs."""
        d = None if direction is None else np.ascontiguousarray(direction, dtype=np.float32).reshape(3)
        N.check(N.lib.adypt_set_sun_visibility(self._ctx, 1 if enabled else 0, None if d is None else d.ctypes.data), self._ctx)

    def TraceAsync(self, n_spp: int = 1) -> None:
        """Trace(true) n_spp times without waiting for the GPU (adypt_trace_spp_async); pair with Wait()."""
        self.m_viewer_type = ViewerTypes.kPTRadiance
        N.check(N.lib.adypt_trace_spp_async(self._ctx, n_spp), self._ctx)

    def Wait(self) -> None:
        N.check(N.lib.adypt_wait(self._ctx), self._ctx)

    def Reset(self) -> None:
        N.check(N.lib.adypt_reset(self._ctx), self._ctx)

    def GetSPP(self) -> int:
        return N.lib.adypt_get_spp(self._ctx)

    def SetFramesInFlight(self, n: int) -> None:
        N.check(N.lib.adypt_set_frames_in_flight(self._ctx, n), self._ctx)

    def GetFramesInFlight(self) -> int:
        return N.lib.adypt_get_frames_in_flight(self._ctx)

    def SetPipeline(self, n_pipes: int) -> None:
        """Sub-batches per batch, each a chain of kernels on its own HIP stream (adypt_set_pipeline); 1 = serial."""
        N.check(N.lib.adypt_set_pipeline(self._ctx, n_pipes), self._ctx)

    def GetPipeline(self) -> int:
        return N.lib.adypt_get_pipeline(self._ctx)

    def SetFusedBounces(self, enabled: bool) -> None:
        """Every bounce after the first of a batch in one launch (k_path); default on.  Images are bit-identical either way."""
        N.check(N.lib.adypt_set_fused_bounces(self._ctx, 1 if enabled else 0), self._ctx)

    def GetFusedBounces(self) -> bool:
        """Whether the last batch of frames ran its bounces in one launch."""
        return N.lib.adypt_get_fused_bounces(self._ctx) == 1

    def CommRanks(self) -> int:
        return N.lib.adypt_comm_ranks(self._ctx)

    def SetLookahead(self, enabled: bool) -> None:
        """One Trace(true) per call (Instance::Update) at batched throughput: see adypt_set_lookahead."""
        N.check(N.lib.adypt_set_lookahead(self._ctx, 1 if enabled else 0), self._ctx)

    def GetLookaheadFrames(self) -> int:
        return N.lib.adypt_get_lookahead_frames(self._ctx)

    def ReadResult(self) -> np.ndarray:
        rgb = np.zeros((self.height, self.width, 3), dtype=np.float32)
        N.check(N.lib.adypt_read_radiance(self._ctx, rgb.ctypes.data), self._ctx)
        return rgb

    def ReadDisplay(self) -> np.ndarray:
        """What OglPathTracer::DrawScreen puts on screen (shaders/screen.glsl:15-21): H x W x 4 uint8."""
        rgba = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        N.check(N.lib.adypt_read_display(self._ctx, rgba.ctypes.data), self._ctx)
        return rgba

    def SavePreview(self, filename: str) -> None:
        save_png(filename, self.ReadDisplay())

    def ReadHits(self) -> Tuple[np.ndarray, np.ndarray]:
        tri = np.full((self.height, self.width), -1, dtype=np.int32)
        uv = np.zeros((self.height, self.width, 2), dtype=np.float32)
        N.check(N.lib.adypt_read_hits(self._ctx, tri.ctypes.data, uv.ctypes.data), self._ctx)
        return tri, uv

    def SaveResult(self, filename: str, save_as_fp16: bool) -> None:
        save_exr(filename, self.ReadResult(), save_as_fp16)

    def TraceRays(self, rays: np.ndarray, with_stats: bool = True, any_hit: bool = False) -> np.ndarray:
        """closest-hit batch (traversal.glsl:14-255) or, with any_hit, the occlusion overload (traversal.glsl:257-494)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), dtype=HIT_DT)
        fn = N.lib.adypt_trace_rays_any if any_hit else N.lib.adypt_trace_rays
        N.check(fn(self._ctx, rays.ctypes.data, len(rays), hits.ctypes.data, 1 if with_stats else 0), self._ctx)
        return hits

    def SetInstrumentation(self, timing: bool = False, counters: bool = False, audit: bool = False) -> None:
        """audit: the slot-claim audit of the ray queues (GetStats()["audit_errors"] must stay 0); a debugging aid, slow."""
        N.check(N.lib.adypt_set_instrumentation(self._ctx, (1 if timing else 0) | (2 if counters else 0) | (4 if audit else 0)), self._ctx)

    def GetStats(self) -> dict:
        st = N.Stats()
        N.check(N.lib.adypt_get_stats(self._ctx, C.byref(st)), self._ctx)
        return st.as_dict()

    def GetWaveProfile(self) -> dict:
        out = (C.c_uint64 * 8)()
        N.check(N.lib.adypt_get_wave_profile(self._ctx, out), self._ctx)
        keys = ("trips", "trip_lanes", "tri_iters", "tri_lanes", "node_phases", "node_lanes", "refills", "empty_trips")
        return dict(zip(keys, [int(v) for v in out]))

    def GetShaderClockGHz(self) -> float:
        """Clock the chip held under the traversal launches since ResetStats (s_memtime / s_memrealtime of workgroup 0); 0.0 if none ran."""
        out = (C.c_uint64 * 2)()
        N.check(N.lib.adypt_get_shader_clock(self._ctx, out), self._ctx)
        return float(out[0]) / float(out[1]) * 0.1 if out[1] else 0.0

    def ResetStats(self) -> None:
        N.check(N.lib.adypt_reset_stats(self._ctx), self._ctx)

    def local_pixel_count(self) -> int:
        return N.lib.adypt_local_pixel_count(self._ctx)

    def local_radiance_device_ptr(self) -> int:
        p = C.c_void_p()
        N.check(N.lib.adypt_local_radiance_device(self._ctx, C.byref(p)), self._ctx)
        return p.value

    def copy_local_radiance(self, dst_device_ptr: int, capacity_float4: int) -> None:
        N.check(N.lib.adypt_copy_local_radiance(self._ctx, dst_device_ptr, capacity_float4), self._ctx)

    def assemble_radiance(self, gathered_device_ptr: int, stride_float4: int, rgb_device_ptr: int) -> None:
        N.check(N.lib.adypt_assemble_radiance(self._ctx, gathered_device_ptr, stride_float4, rgb_device_ptr), self._ctx)

    # ---- one process per GPU: the library's own RCCL communicator (adypt_comm_*) ----
    def CommInit(self, unique_id: bytes) -> None:
        assert len(unique_id) == 128
        N.check(N.lib.adypt_comm_init(self._ctx, unique_id), self._ctx)

    def CommGatherDevice(self) -> Optional[int]:
        """Collective: the one gather.  Rank 0 gets the device pointer of the assembled W x H x 3 fp32 image, the others None."""
        p = C.c_void_p()
        N.check(N.lib.adypt_comm_gather_radiance(self._ctx, C.byref(p)), self._ctx)
        return p.value

    def CommReadResult(self) -> Optional[np.ndarray]:
        """Collective: gather + device-to-host copy on rank 0 (None elsewhere)."""
        rgb = np.zeros((self.height, self.width, 3), dtype=np.float32) if self.tile_rank == 0 else None
        N.check(N.lib.adypt_comm_read_radiance(self._ctx, None if rgb is None else rgb.ctypes.data), self._ctx)
        return rgb

    def CommAllReduce(self, values, op: str = "sum"):
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        N.check(N.lib.adypt_comm_allreduce(self._ctx, arr, len(values), {"sum": 0, "max": 1}[op]), self._ctx)
        return [float(v) for v in arr]

    def CommBarrier(self) -> None:
        N.check(N.lib.adypt_comm_barrier(self._ctx), self._ctx)

    def DeviceSynchronize(self) -> None:
        N.check(N.lib.adypt_device_synchronize(self._ctx), self._ctx)

    def destroy(self) -> None:
        if self._ctx:
            N.lib.adypt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.destroy()
        except Exception:
            pass


class MultiPathTracer:
    """One host process driving N GPUs through the library's own multi-device boundary (adypt_create_multi): tile rank i on
    devices[i], scene replicated, the radiance gathered on devices[0] with RCCL inside adypt_multi_read_radiance.  Same method
    names as HipPathTracer / OglPathTracer."""

    def __init__(self) -> None:
        self._m = C.c_void_p()
        self.m_viewer_type = ViewerTypes.kDiffuse

    def _check(self, code: int) -> None:
        if code != N.ADYPT_OK:
            msg = N.lib.adypt_multi_last_error(self._m if self._m else None)
            raise N.AdyptError(code, (msg or b"").decode("utf-8", "replace"))

    def Initialize(self, config: N.PtParams, scene: HipScene, width: int, height: int, devices: Sequence[int] = (0,)) -> None:
        self.destroy()
        self.width, self.height = width, height
        s, b = scene.scene, scene.bvh
        tex_arr = (N.Texture * max(1, len(s.textures)))()
        self._keep = [s.triangles, s.materials, b.nodes, b.tri_indices, scene.woop, tex_arr] + list(s.textures)
        for i, t in enumerate(s.textures):
            tex_arr[i].width, tex_arr[i].height, tex_arr[i].rgb = t.shape[1], t.shape[0], t.ctypes.data
        d = N.SceneDesc()
        d.nodes, d.n_nodes = b.nodes.ctypes.data, len(b.nodes) // NODE_BYTES
        d.tri_indices, d.n_refs = b.tri_indices.ctypes.data, len(b.tri_indices)
        d.woop = None if scene.woop is None else scene.woop.ctypes.data
        d.triangles, d.n_tris = s.triangles.ctypes.data, len(s.triangles) // TRI_BYTES
        d.materials, d.n_mats = (s.materials.ctypes.data if len(s.materials) else None), len(s.materials) // MAT_BYTES
        d.textures, d.n_textures = (C.addressof(tex_arr) if s.textures else None), len(s.textures)
        d.width, d.height, d.device, d.tile_rank, d.tile_nranks = width, height, 0, 0, 1
        devs = (C.c_int * len(devices))(*devices)
        self._check(N.lib.adypt_create_multi(C.byref(self._m), C.byref(d), devs, len(devices)))
        self.SetConfig(config)

    def SetupSeconds(self, i: int) -> float:
        """Seconds adypt_create took for devices[i] (the contexts are created concurrently, one host thread each)."""
        return float(N.lib.adypt_multi_setup_seconds(self._m, i))

    def SetConfig(self, config: N.PtParams) -> None:
        self._check(N.lib.adypt_multi_set_params(self._m, C.byref(config)))

    def SetCamera(self, inv_projection: np.ndarray, inv_view: np.ndarray, position) -> None:
        ip = np.ascontiguousarray(inv_projection, dtype=np.float32).reshape(16)
        iv = np.ascontiguousarray(inv_view, dtype=np.float32).reshape(16)
        pos = np.ascontiguousarray(position, dtype=np.float32).reshape(3)
        self._check(N.lib.adypt_multi_set_camera(self._m, pos.ctypes.data, ip.ctypes.data, iv.ctypes.data))

    def SetLookahead(self, enabled: bool) -> None:
        self._check(N.lib.adypt_multi_set_lookahead(self._m, 1 if enabled else 0))

    def CommInit(self) -> None:
        self._check(N.lib.adypt_multi_comm_init(self._m))

    def CommRanks(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount); 0 = none (one device, the shared-device test hook)."""
        return N.lib.adypt_multi_comm_ranks(self._m)

    def _contexts(self):
        return [N.lib.adypt_multi_context(self._m, i) for i in range(self.DeviceCount())]

    def GetFramesInFlight(self) -> int:
        return N.lib.adypt_get_frames_in_flight(self._contexts()[0])

    def GetFusedBounces(self) -> bool:
        return N.lib.adypt_get_fused_bounces(self._contexts()[0]) == 1

    def ResetStats(self) -> None:
        for c in self._contexts():
            N.check(N.lib.adypt_reset_stats(c), c)

    def DeviceSynchronize(self) -> None:
        for c in self._contexts():
            N.check(N.lib.adypt_device_synchronize(c), c)

    def GetShaderClockGHz(self) -> float:
        out = (C.c_uint64 * 2)()
        c = self._contexts()[0]
        N.check(N.lib.adypt_get_shader_clock(c, out), c)
        return float(out[0]) / float(out[1]) * 0.1 if out[1] else 0.0

    def Trace(self, enable_pt: bool, n_spp: int = 1) -> None:
        if enable_pt:
            self.m_viewer_type = ViewerTypes.kPTRadiance
            self._check(N.lib.adypt_multi_trace_spp(self._m, n_spp))
        else:
            if self.m_viewer_type == ViewerTypes.kPTRadiance:
                self.m_viewer_type = ViewerTypes.kDiffuse
            self._check(N.lib.adypt_multi_trace_primary(self._m, self.m_viewer_type))

    def Reset(self) -> None:
        self._check(N.lib.adypt_multi_reset(self._m))

    def GetSPP(self) -> int:
        return N.lib.adypt_multi_get_spp(self._m)

    def DeviceCount(self) -> int:
        return N.lib.adypt_multi_device_count(self._m)

    def SetSunVisibility(self, enabled: bool, direction=None) -> None:
        d = None if direction is None else np.ascontiguousarray(direction, dtype=np.float32).reshape(3)
        self._check(N.lib.adypt_multi_set_sun_visibility(self._m, 1 if enabled else 0, None if d is None else d.ctypes.data))

    def SetInstrumentation(self, timing: bool = False, counters: bool = False) -> None:
        self._check(N.lib.adypt_multi_set_instrumentation(self._m, (1 if timing else 0) | (2 if counters else 0)))

    def GetStats(self) -> dict:
        """Counts summed over the devices, kernel times of the slowest one (they run concurrently)."""
        st = N.Stats()
        self._check(N.lib.adypt_multi_get_stats(self._m, C.byref(st)))
        return st.as_dict()

    def ReadDisplay(self) -> np.ndarray:
        """What the reference's window shows (screen.glsl:15-21), H x W x 4 uint8: every device converts its own tiles."""
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._check(N.lib.adypt_multi_read_display(self._m, out.ctypes.data))
        return out

    def ContextStats(self, i: int) -> dict:
        st = N.Stats()
        N.check(N.lib.adypt_get_stats(N.lib.adypt_multi_context(self._m, i), C.byref(st)))
        return st.as_dict()

    def ReadResult(self) -> np.ndarray:
        rgb = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self._check(N.lib.adypt_multi_read_radiance(self._m, rgb.ctypes.data))
        return rgb

    def GatherDevice(self) -> int:
        """Device pointer (on devices[0]) of the assembled W x H x 3 fp32 image; owned by the library."""
        p = C.c_void_p()
        self._check(N.lib.adypt_multi_gather_radiance(self._m, C.byref(p)))
        return p.value

    def SaveResult(self, filename: str, save_as_fp16: bool) -> None:
        save_exr(filename, self.ReadResult(), save_as_fp16)

    def destroy(self) -> None:
        if self._m:
            N.lib.adypt_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.destroy()
        except Exception:
            pass


class Instance:
    """Headless Instance (src/Instance.cpp): config -> scene -> (.bvh cache | build) -> upload -> tracer -> camera."""

    def __init__(self) -> None:
        self.m_config = InstanceConfig()
        self.m_path_tracer = HipPathTracer()
        self.m_hipscene = HipScene()
        self.m_camera = Camera()
        self.m_valid = False

    def InitializeFromFile(self, filename: str, shift_seed: int = 12345, device: int = 0, tile_rank: int = 0,
                           tile_nranks: int = 1, devices: Optional[Sequence[int]] = None) -> bool:
        self.m_filename = filename
        if not self.m_config.LoadFromFile(filename):
            return False
        return self.Initialize(shift_seed, device, tile_rank, tile_nranks, devices)

    def Initialize(self, shift_seed: int = 12345, device: int = 0, tile_rank: int = 0, tile_nranks: int = 1,
                   devices: Optional[Sequence[int]] = None) -> bool:
        """devices: a list of HIP device ordinals -> ONE process drives all of them through adypt_create_multi (tile rank i on
        devices[i]; m_path_tracer is then a MultiPathTracer), as integration/HipPathTracer.hpp does for the reference's Instance."""
        cfg = self.m_config
        self.scene = Scene()
        if not self.scene.LoadFromFile(cfg.m_obj_filename):
            return False
        self.bvh = WideBVH()
        bp = cfg.bvh_params()
        if not self.bvh.LoadFromFile(cfg.m_bvh_filename, bp):
            self.bvh.Build(self.scene, bp)
            if not self.bvh.SaveToFile(cfg.m_bvh_filename, bp):
                return False
        self.m_hipscene.Initialize(self.scene, self.bvh)
        if devices is not None:
            self.m_path_tracer = MultiPathTracer()
            self.m_path_tracer.Initialize(cfg.pt_params(shift_seed), self.m_hipscene, cfg.m_width, cfg.m_height, list(devices))
        else:
            self.m_path_tracer.Initialize(cfg.pt_params(shift_seed), self.m_hipscene, cfg.m_width, cfg.m_height, device, tile_rank, tile_nranks)
        self.m_camera.Initialize(cfg, cfg.m_width, cfg.m_height)
        ip, iv = self.m_camera.matrices()
        self.m_path_tracer.SetCamera(ip, iv, self.m_camera.position)
        self.m_valid = True
        return True

    def Update(self, enable_pt: bool, n_spp: int = 1, keys: int = 0, mouse: Tuple[float, float] = (0.0, 0.0),
               frame_seconds: float = 0.0) -> None:
        """Instance::Update (src/Instance.cpp:44-57) without the window: while not path tracing the camera follows the input
        state (Camera::Control) and is handed to the tracer, then one Trace(enable_pt)."""
        if not enable_pt:
            self.m_camera.Control(keys, mouse[0], mouse[1], frame_seconds)
            ip, iv = self.m_camera.matrices()
            self.m_path_tracer.SetCamera(ip, iv, self.m_camera.position)
        self.m_path_tracer.Trace(enable_pt, n_spp)
