"""Developer tool (GPU box): BASELINE config 2 — sponza stand-in, 1920x1080, primary rays only (adypt_trace_primary, viewer type 0):
one closest-hit query per pixel and the viewer colouring, K calls; wall time and traversal-kernel time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
p.SetInstrumentation(timing=True)
for _ in range(10):
    p.Trace(False)
p.ResetStats(); p.DeviceSynchronize()
t0 = time.perf_counter()
for _ in range(K):
    p.Trace(False)
p.DeviceSynchronize()
dt = time.perf_counter() - t0
s = p.GetStats()
print(json.dumps({"config": "C2: sponza stand-in 1920x1080, primary rays only, %d calls of adypt_trace_primary" % K, "rays": int(s["rays"]),
                  "wall_Mrays_s": round(s["rays"] / dt / 1e6, 1), "ms_per_call": round(dt * 1e3 / K, 4),
                  "trace_kernel_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "trace_kernel_ms_per_call": round(s["trace_ms"] / K, 4),
                  "hits": int(s["hits"])}))
