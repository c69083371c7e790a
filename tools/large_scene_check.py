"""One-off large-scene check (BASELINE.json config 4: ~10 M triangles, 1920x1080, 8 bounces): generate the
'sanmiguel' stand-in, build SBVH -> CWBVH8 on the host, render on the GPU, compare one full frame and a batch of
random rays with the oracle bit for bit, and time a few frames.  Writes one JSON line (also to --out).

    python tools/large_scene_check.py [--scene sanmiguel] [--no-gpu] [--out gpurun_out/large.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sanmiguel")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--cache", default="/tmp/adypt_large")
    ap.add_argument("--no-gpu", action="store_true", help="stop after the host build (no device needed)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from adypt_amd import api, scenes
    os.makedirs(a.cache, exist_ok=True)
    rep = {"scene": a.scene, "width": a.width, "height": a.height}
    t = time.time()
    spec = scenes.make_scene(a.scene, a.cache, width=a.width, height=a.height)
    rep["generate_s"] = round(time.time() - t, 2)
    rep["n_tris"] = spec.n_tris
    cfg = api.InstanceConfig()
    assert cfg.LoadFromFile(spec.config_path)
    t = time.time()
    sc = api.Scene()
    assert sc.LoadFromFile(spec.obj_path), api.InstanceConfig.last_error()
    rep["obj_load_s"] = round(time.time() - t, 2)
    t = time.time()
    bvh = api.WideBVH()
    bvh.Build(sc, cfg.bvh_params())
    rep["bvh_build_s"] = round(time.time() - t, 2)
    from adypt_amd import _native as N
    rep["build_threads"] = int(N.lib.adypt_host_get_threads())
    rep["sbvh_s"] = round(bvh.build_info.sbvh_ms / 1e3, 2)
    rep["wide_s"] = round(bvh.build_info.wide_ms / 1e3, 2)
    rep["wide_nodes"] = int(len(bvh.GetNodes()) // 80)
    rep["refs"] = int(len(bvh.GetTriIndices()))
    if not a.no_gpu:
        from oracle import oracle_py as O
        from tests.helpers import bits, oracle_params_from_config, random_rays
        hs = api.HipScene()
        hs.Initialize(sc, bvh)
        pt = api.HipPathTracer()
        pt.Initialize(cfg.pt_params(4242), hs, a.width, a.height)
        cam = api.Camera()
        cam.Initialize(cfg, a.width, a.height)
        ip, iv = cam.matrices()
        pt.SetCamera(ip, iv, cam.position)
        osc = O.Scene(bvh.nodes, bvh.tri_indices, sc.triangles, sc.materials, textures=sc.textures)
        rays = random_rays(sc.triangles, 500000, 5)
        g = pt.TraceRays(rays, with_stats=True)
        o = O.trace(osc, rays, cfg.c.stack_size)
        rep["random_rays_bit_exact"] = bool(g.tobytes() == o.tobytes())
        rep["random_rays_hit_fraction"] = float((g["tri_id"] >= 0).mean())
        rep["mean_nodes_per_ray"] = float(g["nodes"].mean())
        rep["max_stack_depth"] = int(g["max_depth"].max())
        P = oracle_params_from_config(cfg.c)
        pt.ResetStats()
        pt.Trace(True, 1)
        frame = pt.ReadResult()
        st = O.PathTracerState(a.width, a.height)
        sob = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "sobol_matrices_64x32.u32"), np.uint32)
        t = time.time()
        ost = O.pt_frames(osc, P, O.shift_bytes(4242, a.width, a.height), sob.reshape(64, 32), st, 1).as_dict()
        rep["oracle_frame_s"] = round(time.time() - t, 2)
        rep["frame_bit_exact"] = bool(np.array_equal(bits(frame), bits(st.accum[..., :3])))
        rep["rays_equal"] = bool(pt.GetStats()["rays"] == ost["rays"])
        rep["rays_per_frame"] = int(ost["rays"])
        pt.Trace(True, 16)  # warm-up past the retrace frame
        pt.ReadResult()
        pt.ResetStats()
        t = time.time()
        pt.Trace(True, a.frames)
        pt.ReadResult()
        dt = time.time() - t
        rep["gpu_ms_per_frame"] = round(dt / a.frames * 1e3, 3)
        rep["gpu_mrays_per_s"] = round(pt.GetStats()["rays"] / dt / 1e6, 1)
        rep["frames_in_flight"] = pt.GetFramesInFlight()
        # kernel times (HIP events) and the exact byte census of the same frames -> algorithmic GB/s of the traversal
        pt.Reset(); pt.SetInstrumentation(timing=True); pt.Trace(True, 16); pt.ResetStats()
        pt.Trace(True, a.frames)
        ts = pt.GetStats()
        pt.Reset(); pt.SetInstrumentation(timing=False, counters=True); pt.Trace(True, 16); pt.ResetStats()
        pt.Trace(True, a.frames)
        cs = pt.GetStats()
        alg = 80 * cs["nodes_visited"] + 48 * cs["tris_tested"] + 4 * cs["hits"] + 48 * cs["rays"]
        rep["trace_ms_per_frame"] = round(ts["trace_ms"] / a.frames, 3)
        rep["shade_ms_per_frame"] = round(ts["shade_ms"] / a.frames, 3)
        rep["trace_kernel_mrays_per_s"] = round(ts["rays"] / ts["trace_ms"] / 1e3, 1)
        rep["nodes_per_ray"] = round(cs["nodes_visited"] / cs["rays"], 2)
        rep["tris_per_ray"] = round(cs["tris_tested"] / cs["rays"], 2)
        rep["alg_bytes_per_ray"] = round(alg / cs["rays"], 1)
        rep["trace_alg_GBs"] = round(alg / (ts["trace_ms"] * 1e-3) / 1e9, 1)
        rep["bvh_MB"] = round((len(bvh.nodes) + len(bvh.tri_indices) * 52) / 1e6, 1)  # nodes (bytes) + Woop 48 B + index 4 B per reference
    line = json.dumps(rep)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
