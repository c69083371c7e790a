"""Developer tool (GPU box): when the workgroups of one k_path launch end, in 100 us buckets after the global queue ran dry.  Needs the variant
    tools/build_variant.sh tailhist --transform adypt_amd/csrc/measure/k_path_tail_hist.py     (ADYPT_LIB=adypt_amd/libadypt_tailhist.so)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
scene = os.environ.get("SWEEP_SCENE", "sponza"); fr = int(os.environ.get("SWEEP_FRAMES", "20")); nr = int(os.environ.get("SWEEP_NRANKS", "1"))
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nr)
p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 5); p.DeviceSynchronize(); p.ResetStats()
t0 = time.perf_counter(); p.Trace(True, fr); dt = time.perf_counter() - t0
s = p.GetStats(); w = list(p.GetWaveProfile().values())
hist = []
for v in w:
    hist += [v & 0xffffffff, v >> 32]
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("ADYPT_") and k not in ("ADYPT_RCCL_LIB", "ADYPT_LIB")}, "nranks": nr, "k_path_ms": round(s["path_ms"], 3),
                  "workgroups_ending_per_100us_after_the_queue_ran_dry": hist, "mean_end_us_after_dry": round(s["path_tris"] / max(1, sum(hist)) / 100.0, 1)}))
