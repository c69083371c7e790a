"""Developer tool (GPU box): sweep the tunables of the persistent traversal kernel (env-driven, read at adypt_create)."""
import itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080, pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24})
def run(env, frames=32):
    for k in ("ADYPT_REFILL_MIN", "ADYPT_CHUNK", "ADYPT_TRACE_BLOCKS_PER_CU", "ADYPT_FRAMES_IN_FLIGHT"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True)
    p.Trace(True, 16); p.ResetStats()
    t0 = time.perf_counter(); p.Trace(True, frames); dt = time.perf_counter() - t0
    s = p.GetStats()
    print(json.dumps({"env": env, "trace_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "wall_Mrays_s": round(s["rays"] / dt / 1e6, 1), "trace_ms_per_frame": round(s["trace_ms"] / frames, 3)}))
    sys.stdout.flush()
    p.destroy()
run({})
for r in (4, 8, 12, 24, 32): run({"ADYPT_REFILL_MIN": r})
for c in (64, 96, 192, 256, 384, 512): run({"ADYPT_CHUNK": c})
for b in (3, 4, 5, 6): run({"ADYPT_TRACE_BLOCKS_PER_CU": b})
run({"ADYPT_REFILL_MIN": 8, "ADYPT_TRACE_BLOCKS_PER_CU": 5})
for f in (1, 2, 4, 8, 16): run({"ADYPT_FRAMES_IN_FLIGHT": f})
