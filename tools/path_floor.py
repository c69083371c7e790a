"""Developer tool (GPU box): the floor of the END of a k_path launch (VERDICT r4 task 2).  At the end of a product launch every workgroup holds a few dozen
paths whose remaining bounces run in sequence; is the 0.65-0.76 ms that takes a property of the chain (dependent fetches, one shading round per bounce) or of
the ~1500 nearly empty workgroups contending with each other?  With the cap variant (adypt_amd/csrc/measure/k_path_init_cap.py: a workgroup holds at most 64
paths) single frames of three sizes are traced, one k_path launch each, 7 bounces after the first:
    lone    8 x 8 = 64 paths           one workgroup on the whole chip
    few     32 x 16 = 512 paths        8 workgroups (two XCDs), each alone on its CU
    full64  384 x 256 = 98 304 paths   1536 workgroups = 6 per CU, 64 paths each: the end-of-launch state everywhere at once
and the kernel time per launch (HIP events) is the chain of 7 bounces of the slowest path.  usage: ADYPT_LIB=adypt_amd/libadypt_cap64.so python tools/path_floor.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes, _native as N
PT = {"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]}
cache = os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache")
scene = os.environ.get("FLOOR_SCENE", "sponza")
out = {"lib": os.path.basename(N.LIB_PATH), "scene": scene, "cases": []}
for tag, w, h in (("lone", 8, 8), ("few", 32, 16), ("full64", 384, 256), ("frame_1080p", 1920, 1080)):
    spec = scenes.make_scene(scene, cache, width=w, height=h, pt=PT)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetFramesInFlight(1)          # one frame per pass: every k_path launch carries w x h paths
    p.SetInstrumentation(timing=True)
    p.Trace(True, 16)               # frame 0 re-traces its primaries; the 15 after it start from the cached hits
    launches = 48
    per = []
    for _ in range(launches):
        p.ResetStats()
        p.Trace(True, 1)
        s = p.GetStats()
        if s["path_launches"] == 1:
            per.append((s["path_ms"], s["path_rays"]))
    per.sort()
    ms = [x[0] for x in per]
    rays = sum(x[1] for x in per) / max(1, len(per))
    med = ms[len(ms) // 2]
    out["cases"].append({"case": tag, "image": [w, h], "paths": w * h, "launches": len(per), "k_path_ms_median": round(med, 4), "k_path_ms_min": round(ms[0], 4), "k_path_ms_max": round(ms[-1], 4),
                         "rays_per_launch": round(rays, 1), "bounces_after_the_first": 7, "us_per_bounce_of_the_slowest_path": round(med * 1e3 / 7, 1)})
    # the instrumented kernel over the same kind of launch: loop trips and shading per wave (a slower kernel: counts only, never times)
    p.SetInstrumentation(timing=False, counters=True)
    p.ResetStats()
    p.Trace(True, 8)
    wp, st = p.GetWaveProfile(), p.GetStats()
    out["cases"][-1].update({"instrumented_8_launches": {"wave_profile": wp, "rays": st["path_rays"], "nodes_per_ray": round(st["path_nodes"] / max(1, st["path_rays"]), 2),
                                                          "tris_per_ray": round(st["path_tris"] / max(1, st["path_rays"]), 2)}})
    p.destroy()
print(json.dumps(out))
