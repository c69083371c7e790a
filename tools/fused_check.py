"""Developer tool (GPU box): the one-launch bounce loop (k_path) against the launch-per-bounce pipeline (k_trace + k_shade) — images and ray
counts must be identical bit for bit.  python tools/fused_check.py [scene] [width] [height] [frames]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "tiny0"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 192
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 20
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=W, height=H,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})


def run(fused, counters=False):
    os.environ["ADYPT_FUSED_BOUNCES"] = "1" if fused else "0"
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True, counters=counters)
    t0 = time.perf_counter(); p.Trace(True, frames); dt = time.perf_counter() - t0
    img = p.ReadResult().copy()
    s = p.GetStats()
    p.destroy()
    return img, s, dt


for counters in (False, True):
    a, sa, ta = run(False, counters)
    b, sb, tb = run(True, counters)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    keys = ("rays", "nodes_visited", "tris_tested", "hits", "shaded", "bad_materials", "stack_overflows")
    print(json.dumps({"scene": scene, "size": [W, H], "frames": frames, "instrumented": counters, "images_identical": bool(same),
                      "differing_pixels": int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum()),
                      "per_bounce": {k: int(sa[k]) for k in keys}, "fused": {k: int(sb[k]) for k in keys},
                      "trace_ms": [round(sa["trace_ms"], 3), round(sb["trace_ms"], 3)], "shade_ms": [round(sa["shade_ms"], 3), round(sb["shade_ms"], 3)],
                      "launches": [int(sa["trace_launches"]), int(sb["trace_launches"])], "wall_s": [round(ta, 4), round(tb, 4)]}))
    sys.stdout.flush()
    assert same, "fused and per-bounce images differ"
    assert sa["rays"] == sb["rays"], "ray counts differ"
