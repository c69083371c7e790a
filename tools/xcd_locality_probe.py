"""Developer tool (GPU box): would giving every XCD (its L2) the rays that START in one octant of the scene raise the traversal rate?
adypt_trace_rays puts consecutive eighths of a batch into the 8 XCD-affine queue segments, so ordering the batch by the octant of the ray
origin emulates a k_shade that appends survivors to the segment of their octant.  python tools/xcd_locality_probe.py [scene]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza"
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
lo, hi = inst.scene.GetAABB()
rs = np.random.RandomState(7)
n = 8 << 20
c = 0.5 * (lo + hi); e = 0.5 * (hi - lo)
u = rs.rand(n, 3).astype(np.float32) * 2 - 1
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = c + u * e * np.array([0.8, 0.5, 0.8], np.float32)
d = rs.randn(n, 3).astype(np.float32)
rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
rays[:, 3] = 1e-4
octant = (u[:, 0] > 0).astype(np.int64) | ((u[:, 1] > 0).astype(np.int64) << 1) | ((u[:, 2] > 0).astype(np.int64) << 2)
q = np.clip(((u * 0.5 + 0.5) * 1024).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v
morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
orders = {"random": np.arange(n), "by_octant_of_origin": np.argsort(octant, kind="stable"), "by_morton_code_of_origin": np.argsort(morton, kind="stable")}
p.SetInstrumentation(timing=True)
ref = None
for name, order in orders.items():
    batch = np.ascontiguousarray(rays[order])
    p.TraceRays(batch[: 1 << 20], with_stats=False)
    best = 1e9
    for _ in range(3):
        p.ResetStats()
        h = p.TraceRays(batch, with_stats=False)
        best = min(best, p.GetStats()["trace_ms"])
    back = np.empty_like(h); back[order] = h
    if ref is None: ref = back
    print(json.dumps({"scene": scene, "order": name, "kernel_ms": round(best, 4), "Mrays_s": round(n / best / 1e3, 1),
                      "same_hits": bool(np.array_equal(back["tri_id"], ref["tri_id"]))}))
    sys.stdout.flush()
