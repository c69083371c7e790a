"""Developer tool (GPU box): the clock the chip holds under the traversal kernel (adypt_get_shader_clock: s_memtime / s_memrealtime sampled inside
the launches) on the bench scene, the 10 M-triangle scene and for a primary-only launch.  python tools/clock_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from adypt_amd import api, scenes
for scene in ("sponza", "sanmiguel"):
    spec = scenes.make_scene(scene, "/tmp/adypt_cache", width=1920, height=1080, pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
    inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer; p.SetInstrumentation(timing=True)
    p.Trace(True, 16); p.ResetStats(); p.Trace(True, 32); s = p.GetStats()
    print(scene, "clock GHz", round(p.GetShaderClockGHz(), 4), "trace Mrays/s", round(s["rays"] / s["trace_ms"] / 1e3, 1))
    p.ResetStats(); p.Trace(False); print("  primary only:", round(p.GetShaderClockGHz(), 4))
    p.destroy()
