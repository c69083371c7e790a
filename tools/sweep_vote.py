"""Developer tool (GPU box): A/B the voted-phase traversal kernel against the default one + parity of its results."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080, pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24})
ref = None
def run(env, frames=16):
    global ref
    for k in ("ADYPT_TRACE_VOTE", "ADYPT_TRI_MIN", "ADYPT_REFILL_MIN", "ADYPT_TRACE_BLOCKS_PER_CU"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True)
    p.Trace(True, 4); p.Reset(); p.ResetStats()
    t0 = time.perf_counter(); p.Trace(True, frames); dt = time.perf_counter() - t0
    s = p.GetStats()
    img = p.ReadResult()
    if ref is None: ref = img
    same = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))
    print(json.dumps({"env": env, "same_image": same, "trace_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "wall_Mrays_s": round(s["rays"] / dt / 1e6, 1)}))
    sys.stdout.flush()
    p.destroy()
run({})
for t in (1, 8, 16, 24, 32, 48): run({"ADYPT_TRACE_VOTE": 1, "ADYPT_TRI_MIN": t})
run({"ADYPT_TRACE_VOTE": 1, "ADYPT_TRI_MIN": 16, "ADYPT_REFILL_MIN": 8})
run({"ADYPT_TRACE_VOTE": 1, "ADYPT_TRI_MIN": 24, "ADYPT_TRACE_BLOCKS_PER_CU": 5})
