"""Developer tool (GPU box): every rank of an N-way pixel-tile split rendered on ONE GPU, one after the other (N = 1, 8, 4): wall time of a 20-frame batch,
rays, and the batch's kernel time by part (k_path / camera rays + k_shade_first / the rest) — is the work balanced over the ranks, and where does a
shard's time go?   python tools/shard_breakdown.py"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adypt_amd import api, scenes
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
for n, rs in ((1, [0]), (8, list(range(8))), (4, [0, 1, 2, 3])):
    for r in rs:
        inst = api.Instance()
        assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=r, tile_nranks=n)
        p = inst.m_path_tracer
        p.SetInstrumentation(timing=True)
        p.Trace(True, 20); p.DeviceSynchronize(); p.ResetStats()
        walls = []
        for _ in range(5):
            t0 = time.perf_counter(); p.Trace(True, 20); walls.append(time.perf_counter() - t0)
        s = p.GetStats(); walls.sort()
        print(json.dumps({"n": n, "rank": r, "wall_ms_batch": round(walls[2] * 1e3, 3), "rays_batch": s["rays"] // 5, "path_ms": round(s["path_ms"] / 5, 3),
                          "first_ms": round((s["trace_ms"] - s["path_ms"]) / 5, 3), "other_ms": round(s["shade_ms"] / 5, 3), "path_rays": s["path_rays"] // 5}), flush=True)
