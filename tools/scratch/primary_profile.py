import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
p.SetInstrumentation(counters=True)
p.Trace(False); p.ResetStats(); p.Trace(False)
s = p.GetStats(); w = p.GetWaveProfile()
waves = 256 * 6 * 4
print(json.dumps({"rays": s["rays"], "nodes": s["nodes_visited"], "tris": s["tris_tested"], "wave_profile": w, "trips_per_wave": w["trips"] / waves, "refills_per_wave": w["refills"] / waves,
                  "lanes_per_trip": w["trip_lanes"] / max(1, w["trips"]), "empty_trips_per_wave": w["empty_trips"] / waves, "nodes_per_ray": s["nodes_visited"] / s["rays"], "tris_per_ray": s["tris_tested"] / s["rays"]}))
for env in ({}, ):
    p.SetInstrumentation(timing=True, counters=False)
    for _ in range(8): p.Trace(False)
    p.ResetStats(); p.DeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(200): p.Trace(False)
    p.DeviceSynchronize(); dt = time.perf_counter() - t0
    s = p.GetStats()
    print(json.dumps({"ms_per_call": dt * 5, "trace_ms": s["trace_ms"] / 200, "other_ms": s["shade_ms"] / 200}))
    p.SetInstrumentation(timing=False, counters=False)
    p.DeviceSynchronize(); t0 = time.perf_counter()
    for _ in range(200): p.Trace(False)
    p.DeviceSynchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"ms_per_call_no_timing": dt * 5}))
