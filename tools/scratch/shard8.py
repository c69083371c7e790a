import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=int(os.environ.get("NR", "8")))
p = inst.m_path_tracer
p.SetInstrumentation(timing=bool(int(os.environ.get("TIMING", "1"))))
p.Trace(True, 5); p.DeviceSynchronize(); p.ResetStats()
walls = []
for _ in range(10):
    p.Reset(); p.Trace(True, 5); p.DeviceSynchronize()
    t0 = time.perf_counter(); p.Trace(True, 20); p.DeviceSynchronize(); walls.append(time.perf_counter() - t0)
walls.sort()
print(json.dumps({"wall_ms_batch_median": round(walls[5] * 1e3, 3), "min": round(walls[0] * 1e3, 3)}))
