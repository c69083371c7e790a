"""Developer tool (GPU box): where a k_path wave's time goes inside one iteration of its persistent loop — average shader cycles per trip and dependent section.
Needs the stamp variant (adypt_amd/csrc/measure/k_path_timeline.py):
    tools/build_variant.sh timeline --transform adypt_amd/csrc/measure/k_path_timeline.py
    tools/build_variant.sh timeline_cap64 --transform adypt_amd/csrc/measure/k_path_timeline.py --transform adypt_amd/csrc/measure/k_path_init_cap.py
    ADYPT_LIB=adypt_amd/libadypt_timeline.so python tools/path_timeline.py bench          the bench workload: 6 waves per SIMD, every SIMD busy
    ADYPT_LIB=adypt_amd/libadypt_timeline_cap64.so python tools/path_timeline.py lone     ONE workgroup of 64 paths alone on the chip (the end of a launch, isolated)
Every section's figure contains the cost of one stamp (s_memtime + wait), reported beside it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes, _native as N
PT = {"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]}
NAMES = ["before_trip(setup,exchange,shading,sleep)", "A_pop_choose_push", "B_match(table round trip)", "B_pull_and_issue_loads", "triangle_fetch_wait", "C_woop_verdicts_winner", "D_node_wait_slab_E"]
case = sys.argv[1] if len(sys.argv) > 1 else "bench"
scene = os.environ.get("SWEEP_SCENE", "sponza")
w, h, fif, frames = {"bench": (1920, 1080, 0, 20), "lone": (8, 8, 1, 48), "few": (32, 16, 1, 48), "full64": (384, 256, 1, 48)}[case]
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=w, height=h, pt=PT)
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
if fif: p.SetFramesInFlight(fif)
p.SetInstrumentation(timing=True)
p.Trace(True, 16 if fif else 5); p.DeviceSynchronize(); p.ResetStats()
p.Trace(True, frames); p.DeviceSynchronize()
s = p.GetStats(); wp = list(p.GetWaveProfile().values())
trips = wp[7] >> 36; stamp = (wp[7] & ((1 << 36) - 1)) / max(1, trips)
clock = p.GetShaderClockGHz()
sec = {n: round(wp[i] / max(1, trips), 1) for i, n in enumerate(NAMES)}
total = sum(wp[:7]) / max(1, trips)
print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "case": case, "scene": scene, "image": [w, h], "frames": frames, "k_path_launches": s["path_launches"], "k_path_rays": int(s["path_rays"]),
                  "k_path_ms_per_launch": round(s["path_ms"] / max(1, s["path_launches"]), 4), "wave_trips": int(trips), "wave_trips_per_ray": round(trips / max(1, s["path_rays"]), 5),
                  "cycles_per_trip_by_section": sec, "cycles_per_trip_total": round(total, 1), "stamp_cost_cycles": round(stamp, 1), "stamps_per_trip": 7,
                  "cycles_per_trip_without_stamps": round(total - 7 * stamp, 1), "shader_clock": clock,
                  "note": "sums over every wave of the launch(es) / wave-trips; a section's figure includes the one stamp that closes it"}))
