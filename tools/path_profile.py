"""Developer tool (GPU box): where the waves of k_path spend their time.  Needs the profile variant of the library:
    tools/build_variant.sh prof --transform adypt_amd/csrc/measure/k_path_profile.py   and   ADYPT_LIB=adypt_amd/libadypt_prof.so python tools/path_profile.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
scene = os.environ.get("SWEEP_SCENE", "sponza"); fr = int(os.environ.get("SWEEP_FRAMES", "20")); nr = int(os.environ.get("SWEEP_NRANKS", "1"))
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nr)
p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 5); p.DeviceSynchronize(); p.ResetStats()
# (the profile variant keeps three 100 MHz timestamps in otherwise unused counters: start / first wave that finds the queue dry (atomicMin) / last end)
import ctypes
from adypt_amd import _native as N
t0 = time.perf_counter(); p.Trace(True, fr); dt = time.perf_counter() - t0
s = p.GetStats(); w = list(p.GetWaveProfile().values())
total = w[0]
hi = lambda v: v >> 32
lo = lambda v: v & 0xffffffff
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("ADYPT_") and k != "ADYPT_RCCL_LIB"}, "ms_per_frame": round(dt * 1e3 / fr, 4), "k_path_ms": round(s["trace_ms"], 3),
                  "wave_time_shares": {"exchange_incl_shading": round(w[1] / total, 4), "shading_rounds": round(w[2] / total, 4), "idle_sleep": round(w[3] / total, 4),
                                       "lock_wait": round(lo(w[6]) * 256 / total, 4), "trips_and_rest": round(1 - (w[1] + w[3]) / total, 4)},
                  "shading_rounds": hi(w[5]), "paths_per_round": round(lo(w[5]) / max(1, hi(w[5])), 2), "exchanges": hi(w[6]), "trips": hi(w[7]),
                  "lanes_per_trip": round(lo(w[7]) * 64 / max(1, hi(w[7])), 2), "wanted_to_shade_but_busy": w[4] >> 40,
                  "avg_to_shade_backlog_at_exchange": round((w[4] & ((1 << 40) - 1)) / max(1, hi(w[6])), 1),
                  "avg_round_cycles": round(w[2] / max(1, hi(w[5]))), "avg_exchange_cycles_excl_shading": round((w[1] - w[2]) / max(1, hi(w[6]))), "rays": int(s["rays"]),
                  "launch_us": round((s["path_tris"] - s["path_hits"]) / 100.0, 1) if s["path_hits"] else None,
                  "tail_us_after_the_queue_ran_dry": round((s["path_tris"] - s["path_nodes"]) / 100.0, 1) if s["path_nodes"] else None}))
