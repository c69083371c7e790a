"""Shared helpers of the test-suite: load fixtures into oracle / product objects."""
import os

import numpy as np

from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_scene(name):
    """(cfg, tri_indices, nodes, tris, mats, woop) produced by the REFERENCE code for fixture `name`."""
    cfg, idx, nodes = O.load_bvh_file(os.path.join(GOLDEN, name + ".bvh"))
    tris = np.fromfile(os.path.join(GOLDEN, name + ".tris"), dtype=O.TRI_DT)
    mats = np.fromfile(os.path.join(GOLDEN, name + ".mats"), dtype=O.MAT_DT)
    woop = np.fromfile(os.path.join(GOLDEN, name + ".woop"), dtype=np.float32).reshape(-1, 12)
    return cfg, idx, nodes, tris, mats, woop


def oracle_scene_from_golden(name):
    _, idx, nodes, tris, mats, woop = golden_scene(name)
    return O.Scene(nodes, idx, tris, mats, woop=woop)


def oracle_scene_from_instance(inst):
    return O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)


def oracle_params_from_config(c):
    ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
    return O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce,
                         subpixel=c.subpixel, tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_rays(tris_bytes, n, seed):
    rs = np.random.RandomState(seed)
    p = np.frombuffer(np.ascontiguousarray(tris_bytes).tobytes(), dtype=O.TRI_DT)["p"].reshape(-1, 3)
    lo, hi = p.min(0), p.max(0)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(lo, hi, size=(n, 3))
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(n, 3))
    k = max(1, n // 20)
    rays[:k, 4] = 0
    rays[k:2 * k, 5] = 0
    rays[2 * k:3 * k, 4:6] = 0
    rays[3 * k:4 * k, 4:7] *= 1e-30
    return rays
