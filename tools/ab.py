"""Developer tool (GPU box): A/B two builds of the library (ADYPT_LIB) in separate processes, interleaved."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from adypt_amd import api, scenes
spec = scenes.make_scene(os.environ.get("AB_SCENE", "sponza"), os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080, pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24})
nr = int(os.environ.get("AB_NRANKS", "1")); fr = int(os.environ.get("AB_FRAMES", "32"))
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nr)
p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 8); p.Reset(); p.ResetStats()
t0 = time.perf_counter(); p.Trace(True, fr); dt = time.perf_counter() - t0; s = p.GetStats()
print(json.dumps({"lib": os.path.basename(os.environ.get("ADYPT_LIB", "default")), "trace_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "wall_Mrays_s": round(s["rays"] / dt / 1e6, 1), "shade_ms_per_frame": round(s["shade_ms"] / fr, 4), "trace_ms_per_frame": round(s["trace_ms"] / fr, 4), "nranks": nr, "frames": fr, "scene": os.environ.get("AB_SCENE", "sponza")}))
''' % ROOT
libs = sys.argv[1:]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default": env["ADYPT_LIB"] = os.path.join(ROOT, "adypt_amd", lib)
        print(subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE).stdout.decode().strip()); sys.stdout.flush()
