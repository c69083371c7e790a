"""Developer tool (GPU box): how much would overlapping the kernels of two independent batches buy?  Two contexts (separate
HIP streams) render the same image concurrently, driven asynchronously from one thread; compare with one context
rendering the same number of frames alone.  python tools/overlap_probe.py [frames_per_context]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # render rank 0's shard of an N-way tile split
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})


def make(fif):
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nranks)
    inst.m_path_tracer.SetFramesInFlight(fif)
    inst.m_path_tracer.Trace(True, 16)
    inst.m_path_tracer.Reset(); inst.m_path_tracer.ResetStats()
    return inst


for fif in ((32, 16) if nranks == 1 else (128, 64, 32)):
    one = make(fif)
    t0 = time.perf_counter(); one.m_path_tracer.Trace(True, 2 * frames); dt1 = time.perf_counter() - t0
    rays1 = one.m_path_tracer.GetStats()["rays"]
    del one
    a, b = make(fif), make(fif)
    t0 = time.perf_counter()
    a.m_path_tracer.TraceAsync(frames); b.m_path_tracer.TraceAsync(frames)
    a.m_path_tracer.Wait(); b.m_path_tracer.Wait()
    dt2 = time.perf_counter() - t0
    rays2 = a.m_path_tracer.GetStats()["rays"] + b.m_path_tracer.GetStats()["rays"]
    print(json.dumps({"frames_in_flight": fif, "one_context_Mrays_s": round(rays1 / dt1 / 1e6, 1), "two_contexts_Mrays_s": round(rays2 / dt2 / 1e6, 1),
                      "gain": round((rays2 / dt2) / (rays1 / dt1), 3), "nranks": nranks, "blocks_per_cu": os.environ.get("ADYPT_TRACE_BLOCKS_PER_CU", "5")}))
    del a, b
