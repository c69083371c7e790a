"""Developer tool (GPU box): minimal workload for rocprofv3 counter passes — sponza stand-in 1920x1080, N frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": tl, "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
inst.m_path_tracer.Trace(True, n)
print("rays", inst.m_path_tracer.GetStats()["rays"])
