"""Developer tool (GPU box): what a k_trace launch costs besides its rays.  Incoherent rays (random origins around the camera
path inside the atrium, random directions) in batches of growing size: kernel time = a + b * n; a is the ramp + drain ("tail") of
the persistent kernel.  Also the distribution of node visits per ray (the drain lasts as long as the longest ray in flight)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
lo, hi = inst.scene.GetAABB()
rs = np.random.RandomState(7)
nmax = 8 << 20
rays = np.zeros((nmax, 8), np.float32)
c = 0.5 * (lo + hi); e = 0.5 * (hi - lo)
rays[:, 0:3] = c + (rs.rand(nmax, 3).astype(np.float32) * 2 - 1) * e * np.array([0.8, 0.5, 0.25], np.float32)
d = rs.randn(nmax, 3).astype(np.float32)
rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
rays[:, 3] = 1e-4
h = p.TraceRays(rays[: 1 << 20], with_stats=True)
nd = h["nodes"]
print(json.dumps({"nodes_mean": float(nd.mean()), "tris_mean": float(h["tris"].mean()), "hit_frac": float((h["tri_id"] >= 0).mean()),
                  "nodes_percentiles_50_90_99_99.9_max": [int(np.percentile(nd, q)) for q in (50, 90, 99, 99.9)] + [int(nd.max())]}))
p.SetInstrumentation(timing=True)
pts = []
for n in (64, 4096, 1 << 15, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
    p.TraceRays(rays[:n], with_stats=False)
    best = 1e9
    for _ in range(3):
        p.ResetStats()
        p.TraceRays(rays[:n], with_stats=False)
        best = min(best, p.GetStats()["trace_ms"])
    pts.append((n, best))
    print(json.dumps({"rays": n, "kernel_ms": round(best, 4), "Mrays_s": round(n / best / 1e3, 1)}))
    sys.stdout.flush()
x = np.array([q[0] for q in pts[5:]], float); y = np.array([q[1] for q in pts[5:]], float)
b, a = np.polyfit(x, y, 1)
print(json.dumps({"fit_ms": {"fixed_per_launch": round(a, 4), "per_Mray": round(b * 1e6, 4)}, "asymptotic_Mrays_s": round(1e-3 / b, 1)}))
