"""Developer tool (GPU box): predict the N-GPU strong-scaling of bench.py on ONE GPU by rendering rank 0's pixel-tile shard
of an N-way split (the ranks never communicate while rendering, so a rank's time is independent of the others)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
life = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ranks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": life, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
base = None
for n in ranks:
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=n)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True)
    p.Trace(True, max(16, life)); p.Reset(); p.ResetStats()
    t0 = time.perf_counter(); p.Trace(True, steps); dt = time.perf_counter() - t0
    s = p.GetStats()
    ms = dt * 1e3 / steps
    if base is None: base = ms
    print(json.dumps({"nranks": n, "life": life, "fif": p.GetFramesInFlight(), "ms_per_step_rank0": round(ms, 4), "rays_rank0": int(s["rays"]),
                      "trace_ms": round(s["trace_ms"] / steps, 4), "shade_ms": round(s["shade_ms"] / steps, 4),
                      "predicted_efficiency": round(base / (n * ms), 3)}))
    sys.stdout.flush()
    del p, inst
