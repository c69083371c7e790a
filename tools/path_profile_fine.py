import json, os, sys, time
sys.path.insert(0, os.getcwd())
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 5); p.DeviceSynchronize(); p.ResetStats()
p.Trace(True, 20)
s = p.GetStats(); w = list(p.GetWaveProfile().values()); T = w[0]
names = ["total", "exch: remap+hit (pre-lock)", "exch: lock wait", "exch: in lock", "round: park+table+gathers+respond", "round: table write+replacement", "round: wait stores+unpark", "round: publish"]
print(json.dumps({"k_path_ms": round(s["trace_ms"], 2), "shares": {n: round(v / T, 4) for n, v in zip(names[1:], w[1:])}, "sum": round(sum(w[1:]) / T, 4)}))
