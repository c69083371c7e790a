"""Developer tool (GPU box): ms per adypt_trace_primary call (BASELINE config 2, 1080p bench scene, 300 calls, timing events on) and per k_trace_camera
launch for a list of environment settings, each in its own process, two rounds.   python tools/primary_sweep.py "base" "ADYPT_BITE_PRIMARY=32 ADYPT_REFILL_MIN_PRIMARY=32" ..."""
import json, os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
p.SetInstrumentation(timing=True, counters=False)
for _ in range(8): p.Trace(False)
p.ResetStats(); p.DeviceSynchronize()
t0 = time.perf_counter()
for _ in range(300): p.Trace(False)
p.DeviceSynchronize(); dt = time.perf_counter() - t0
s = p.GetStats()
print(json.dumps({"env": os.environ.get("SW"), "ms_per_call": round(dt / 300 * 1e3, 4), "kernel_ms": round(s["trace_ms"] / 300, 4)}))
''' % ROOT
for rnd in range(2):
    for setting in sys.argv[1:]:
        env = dict(os.environ); env["SW"] = setting
        for kv in setting.split():
            if "=" in kv:
                k, v = kv.split("=", 1); env[k] = v
        subprocess.run([sys.executable, "-c", code], env=env, check=True)
