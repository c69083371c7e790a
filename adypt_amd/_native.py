"""ctypes binding of adypt_amd/libadypt_hip.so (C-ABI: include/adypt_hip.h + include/adypt_host.h).

The library is built in-tree by ``adypt_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no Python or
CPU fallback for the GPU path: if the shared object is missing, importing this module raises, and without a HIP
device ``adypt_create`` fails with ``ADYPT_E_NO_DEVICE``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADYPT_LIB", os.path.join(_HERE, "libadypt_hip.so"))  # ADYPT_LIB: measurement-only variant builds

ADYPT_OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_OOM, E_STACK_OVERFLOW, E_BAD_MATERIAL, E_IO, E_PARSE, E_STATE = range(-1, -10, -1)
_ERR_NAMES = {E_INVALID: "ADYPT_E_INVALID", E_NO_DEVICE: "ADYPT_E_NO_DEVICE", E_HIP: "ADYPT_E_HIP", E_OOM: "ADYPT_E_OOM",
              E_STACK_OVERFLOW: "ADYPT_E_STACK_OVERFLOW", E_BAD_MATERIAL: "ADYPT_E_BAD_MATERIAL", E_IO: "ADYPT_E_IO",
              E_PARSE: "ADYPT_E_PARSE", E_STATE: "ADYPT_E_STATE"}


class AdyptError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s (%d): %s" % (_ERR_NAMES.get(code, "ADYPT_E_?"), code, msg))
        self.code = code


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_int64), ("tri_indices", C.c_void_p), ("n_refs", C.c_int64),
                ("woop", C.c_void_p), ("triangles", C.c_void_p), ("n_tris", C.c_int64), ("materials", C.c_void_p),
                ("n_mats", C.c_int64), ("textures", C.c_void_p), ("n_textures", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("device", C.c_int32), ("tile_rank", C.c_int32), ("tile_nranks", C.c_int32)]


class PtParams(C.Structure):
    _fields_ = [("stack_size", C.c_int32), ("max_bounce", C.c_int32), ("subpixel", C.c_int32), ("tmp_lifetime", C.c_int32),
                ("ray_tmin", C.c_float), ("clamp", C.c_float), ("sun", C.c_float * 3), ("shift_seed", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("ref_idx", C.c_int32), ("tri_id", C.c_int32), ("u", C.c_float), ("v", C.c_float), ("t", C.c_float),
                ("nodes", C.c_uint32), ("tris", C.c_uint32), ("hash", C.c_uint32), ("max_depth", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("hits", C.c_uint64),
                ("shaded", C.c_uint64), ("stack_overflows", C.c_uint64), ("bad_materials", C.c_uint64),
                ("max_stack", C.c_uint32), ("trace_launches", C.c_uint32), ("trace_ms", C.c_double), ("shade_ms", C.c_double),
                ("path_ms", C.c_double), ("path_launches", C.c_uint32), ("audit_errors", C.c_uint32), ("path_rays", C.c_uint64),
                ("path_nodes", C.c_uint64), ("path_tris", C.c_uint64), ("path_hits", C.c_uint64), ("path_shaded", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BvhParams(C.Structure):
    _fields_ = [("max_spatial_depth", C.c_int32), ("triangle_sah", C.c_float), ("node_sah", C.c_float)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bvh", BvhParams),
                ("invocation_size", C.c_int32), ("stack_size", C.c_int32), ("max_bounce", C.c_int32), ("subpixel", C.c_int32),
                ("tmp_lifetime", C.c_int32), ("ray_tmin", C.c_float), ("clamp", C.c_float), ("sun", C.c_float * 3),
                ("speed", C.c_float), ("mouse_sensitive", C.c_float), ("fov", C.c_float), ("yaw", C.c_float),
                ("pitch", C.c_float), ("position", C.c_float * 3), ("obj_filename", C.c_char * 1024),
                ("bvh_filename", C.c_char * 1024)]


class BuildInfo(C.Structure):
    _fields_ = [("sbvh_nodes", C.c_int64), ("refs", C.c_int64), ("wide_nodes", C.c_int64), ("sbvh_ms", C.c_double),
                ("wide_ms", C.c_double)]


# every symbol include/*.h declares, with its signature (tests/test_abi.py checks the export list against the headers)
_SIGS = {
    # adypt_hip.h
    "adypt_abi_version": (C.c_int, []),
    "adypt_enable_test_hooks": (C.c_int, [C.c_uint64]),
    "adypt_test_hooks_enabled": (C.c_int, []),
    "adypt_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(SceneDesc)]),
    "adypt_destroy": (None, [C.c_void_p]),
    "adypt_last_error": (C.c_char_p, [C.c_void_p]),
    "adypt_set_params": (C.c_int, [C.c_void_p, C.POINTER(PtParams)]),
    "adypt_set_camera": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adypt_trace_primary": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_trace_spp": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_reset": (C.c_int, [C.c_void_p]),
    "adypt_get_spp": (C.c_int, [C.c_void_p]),
    "adypt_set_frames_in_flight": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_get_frames_in_flight": (C.c_int, [C.c_void_p]),
    "adypt_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_get_pipeline": (C.c_int, [C.c_void_p]),
    "adypt_set_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_get_lookahead_frames": (C.c_int, [C.c_void_p]),
    "adypt_read_radiance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adypt_read_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "adypt_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "adypt_trace_rays_any": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "adypt_set_instrumentation": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "adypt_reset_stats": (C.c_int, [C.c_void_p]),
    "adypt_get_wave_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "adypt_get_shader_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "adypt_local_pixel_count": (C.c_int64, [C.c_void_p]),
    "adypt_set_sun_visibility": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "adypt_trace_spp_async": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_wait": (C.c_int, [C.c_void_p]),
    "adypt_read_display": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adypt_local_radiance_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_copy_local_radiance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "adypt_assemble_radiance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "adypt_shard_block_count": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "adypt_untile_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    # native multi-GPU (RCCL inside the library)
    "adypt_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int]),
    "adypt_destroy_multi": (None, [C.c_void_p]),
    "adypt_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "adypt_multi_setup_seconds": (C.c_double, [C.c_void_p, C.c_int]),
    "adypt_multi_device_count": (C.c_int, [C.c_void_p]),
    "adypt_multi_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "adypt_multi_set_params": (C.c_int, [C.c_void_p, C.POINTER(PtParams)]),
    "adypt_multi_set_camera": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adypt_multi_set_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_multi_trace_primary": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_multi_trace_spp": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_multi_reset": (C.c_int, [C.c_void_p]),
    "adypt_multi_get_spp": (C.c_int, [C.c_void_p]),
    "adypt_multi_set_sun_visibility": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "adypt_multi_set_instrumentation": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_multi_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "adypt_multi_read_display": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adypt_multi_read_radiance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adypt_multi_gather_radiance": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_multi_comm_init": (C.c_int, [C.c_void_p]),
    "adypt_multi_comm_ranks": (C.c_int, [C.c_void_p]),
    "adypt_comm_ranks": (C.c_int, [C.c_void_p]),
    "adypt_set_fused_bounces": (C.c_int, [C.c_void_p, C.c_int]),
    "adypt_get_fused_bounces": (C.c_int, [C.c_void_p]),
    "adypt_comm_unique_id": (C.c_int, [C.c_char_p]),
    "adypt_comm_init": (C.c_int, [C.c_void_p, C.c_char_p]),
    "adypt_comm_gather_radiance": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_comm_read_radiance": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adypt_comm_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "adypt_comm_barrier": (C.c_int, [C.c_void_p]),
    "adypt_device_synchronize": (C.c_int, [C.c_void_p]),
    # adypt_host.h
    "adypt_config_default": (None, [C.POINTER(Config)]),
    "adypt_config_load": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "adypt_config_parse": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "adypt_config_json": (C.c_size_t, [C.POINTER(Config), C.c_char_p, C.c_size_t]),
    "adypt_config_save": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "adypt_host_last_error": (C.c_char_p, []),
    "adypt_camera_control": (None, [C.POINTER(Config), C.c_uint32, C.c_float, C.c_float, C.c_float]),
    "adypt_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "adypt_scene_free": (None, [C.c_void_p]),
    "adypt_scene_triangles": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_scene_materials": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_scene_textures": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_scene_aabb": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "adypt_scene_warnings": (C.c_char_p, [C.c_void_p]),
    "adypt_scene_from_arrays": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "adypt_host_set_threads": (C.c_int, [C.c_int]),
    "adypt_host_get_threads": (C.c_int, []),
    "adypt_host_selftest_sort": (C.c_int, [C.c_int64, C.c_uint32, C.c_int, C.c_int, C.c_int64]),
    "adypt_bvh_build": (C.c_int, [C.c_void_p, C.POINTER(BvhParams), C.POINTER(C.c_void_p), C.POINTER(BuildInfo)]),
    "adypt_bvh_load": (C.c_int, [C.c_char_p, C.POINTER(BvhParams), C.POINTER(C.c_void_p)]),
    "adypt_bvh_save": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(BvhParams)]),
    "adypt_bvh_free": (None, [C.c_void_p]),
    "adypt_bvh_nodes": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_bvh_tri_indices": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adypt_woop_matrices": (None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "adypt_camera_matrices": (None, [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "adypt_sobol_points": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adypt_shift_bytes": (None, [C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "adypt_save_exr": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "adypt_save_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "adypt_load_image_rgb8": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "adypt_load_exr": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "adypt_free": (None, [C.c_void_p]),
}

EXPORTS = tuple(_SIGS)

if not os.path.exists(LIB_PATH):
    raise ImportError("adypt_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no fallback implementation" % LIB_PATH)

def _share_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and load it
    by file name, so a process that loads libadypt_hip.so (bound to /opt/rocm's copy) *before* `import torch` ends up with
    two runtimes, and the one that touches the GPU second finds no device.  Pre-loading torch's copy — without importing
    torch — makes both bind to the same runtime whatever the import order.  ADYPT_NO_TORCH_RUNTIME=1 disables this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("ADYPT_NO_TORCH_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError:
            return
        # the RCCL that goes with that runtime (the library dlopens RCCL on first use of the native multi-GPU path)
        rccl = os.path.join(os.path.dirname(rt), "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("ADYPT_RCCL_LIB", rccl)


_share_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what the headers declare
    _fn.restype = _res
    _fn.argtypes = _args
if lib.adypt_abi_version() != 4:
    raise ImportError("adypt_amd: ABI version mismatch")


def check(code: int, ctx=None) -> None:
    if code == ADYPT_OK:
        return
    msg = lib.adypt_last_error(ctx)
    if (not msg) and ctx is not None:
        msg = b""
    host = lib.adypt_host_last_error()
    text = (msg or b"").decode("utf-8", "replace") or (host or b"").decode("utf-8", "replace")
    raise AdyptError(code, text)


def check_host(code: int) -> None:
    if code != ADYPT_OK:
        raise AdyptError(code, (lib.adypt_host_last_error() or b"").decode("utf-8", "replace"))
