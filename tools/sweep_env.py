"""Developer tool (GPU box): time the bench workload for a list of environment settings (tunables read at adypt_create).
    python tools/sweep_env.py ADYPT_TRI_MIN=1 ADYPT_TRI_MIN=8 "ADYPT_TRI_MIN=8 ADYPT_REFILL_MIN=8" ...
Each setting runs in this process on a fresh context; prints one JSON line per setting (traversal-kernel and wall rates)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
scene = os.environ.get("SWEEP_SCENE", "sponza")
frames = int(os.environ.get("SWEEP_FRAMES", "32"))
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
touched = set()
def run(setting):
    for k in touched:
        os.environ.pop(k, None)
    env = dict(kv.split("=", 1) for kv in setting.split()) if setting else {}
    touched.update(env)
    os.environ.update(env)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True)
    p.Trace(True, 16); p.ResetStats()
    t0 = time.perf_counter(); p.Trace(True, frames); dt = time.perf_counter() - t0
    s = p.GetStats()
    print(json.dumps({"env": setting, "trace_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "wall_Mrays_s": round(s["rays"] / dt / 1e6, 1),
                      "trace_ms_per_frame": round(s["trace_ms"] / frames, 3), "shade_ms_per_frame": round(s["shade_ms"] / frames, 3), "image_sum": float(p.ReadResult().sum())}))
    sys.stdout.flush()
    p.destroy()
for s in (sys.argv[1:] or [""]):
    run(s)
