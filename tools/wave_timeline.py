"""Developer tool (GPU box), measurement-only build (-DADYPT_MEASUREMENT_BUILD -DADYPT_ABLATE_WAVE_TIMELINE, ADYPT_LIB=...): the wave timeline of
ONE k_trace<false,false> launch over n incoherent rays — when the persistent waves start, get their first rays, find the queue dry, end."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes, _native as N
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
lo, hi = inst.scene.GetAABB()
rs = np.random.RandomState(7)
nmax = 8 << 20
rays = np.zeros((nmax, 8), np.float32)
c = 0.5 * (lo + hi); e = 0.5 * (hi - lo)
rays[:, 0:3] = c + (rs.rand(nmax, 3).astype(np.float32) * 2 - 1) * e * np.array([0.8, 0.5, 0.25], np.float32)
d = rs.randn(nmax, 3).astype(np.float32); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 3] = 1e-4
p.SetInstrumentation(timing=True)
N.lib.adypt_debug_read_timeline.argtypes = [ctypes.c_void_p]
for n in (1 << 16, 1 << 19, 1 << 21, 1 << 22, 1 << 23):
    p.TraceRays(rays[:n], with_stats=False)
    p.ResetStats()
    p.TraceRays(rays[:n], with_stats=False)
    ms = p.GetStats()["trace_ms"]
    t = np.zeros((8192, 4), np.uint64)
    assert N.lib.adypt_debug_read_timeline(t.ctypes.data) == 0
    t = t[t[:, 0] > 0].astype(np.int64)  # every persistent wave of the launch (6144 at 6 workgroups per CU)
    t0 = t[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    start, first, dry, end = us(t[:, 0]), us(np.where(t[:, 1] > 0, t[:, 1], t[:, 3])), us(np.where(t[:, 2] > 0, t[:, 2], t[:, 3])), us(t[:, 3])
    got = t[:, 1] > 0
    pct = lambda a: [round(float(np.percentile(a, q)), 1) for q in (0, 10, 50, 90, 99, 100)]
    print(json.dumps({"rays": n, "waves": int(len(t)), "kernel_ms_hip_events": round(ms, 4), "waves_that_got_rays": int(got.sum()),
                      "percentiles_0_10_50_90_99_100_us": {"wave_start": pct(start), "first_rays": pct(first[got]) if got.any() else None, "queue_found_dry": pct(dry), "wave_end": pct(end),
                                                          "dry_to_end_of_waves_with_rays": pct((end - dry)[got]) if got.any() else None},
                      "waves_still_running_at_us": {str(k): int((end > k).sum()) for k in (50, 100, 150, 200, 300, 400, 600, 800, 1000, 1200)}}))
