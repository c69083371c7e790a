"""Developer tool (GPU box): broad GPU-vs-oracle parity sweep + quick timing, printing diagnostics instead of
asserting so that one gpurun call yields as much information as possible.  The pytest -m gpu suite is the gate."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CACHE = os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache")
SOBOL = np.fromfile(os.path.join(ROOT, "tests", "golden", "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)


def oracle_scene(inst):
    return O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)


def oracle_params(c):
    ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
    return O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce,
                         subpixel=c.subpixel, tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))


def random_rays(tris, n, seed):
    rs = np.random.RandomState(seed)
    p = np.frombuffer(tris.tobytes(), dtype=O.TRI_DT)["p"].reshape(-1, 3)
    lo, hi = p.min(0), p.max(0)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(lo, hi, size=(n, 3))
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(n, 3))
    k = n // 20
    rays[:k, 4] = 0
    rays[k:2 * k, 5] = 0
    rays[2 * k:3 * k, 4:6] = 0
    rays[3 * k:4 * k, 4:7] *= 1e-30
    return rays


def parity(name, w, h, spp, n_rays, pt=None):
    spec = scenes.make_scene(name, CACHE, width=w, height=h, pt=pt)
    inst = api.Instance()
    t0 = time.time()
    ok = inst.InitializeFromFile(spec.config_path, shift_seed=99)
    assert ok, api.InstanceConfig.last_error()
    t_init = time.time() - t0
    c = inst.m_config.c
    pt_ = inst.m_path_tracer
    osc, P = oracle_scene(inst), oracle_params(c)
    res = {"scene": name, "tris": inst.scene.n_tris, "init_s": round(t_init, 2)}
    # primary frame, every viewer type
    for vt in (0, 1, 2, 4, 5):
        pt_.m_viewer_type = vt
        pt_.Trace(False)
        g = pt_.ReadResult()
        rgba, hits, _ = O.primary_frame(osc, P, vt)
        res["viewer%d_mismatch_px" % vt] = int((rgba[..., :3].view(np.uint32) != g.view(np.uint32)).any(axis=-1).sum())
    tri, uv = pt_.ReadHits()
    res["primary_tri_mismatch"] = int((tri != hits["tri_id"]).sum())
    hit_mask = hits["tri_id"] >= 0
    res["primary_uv_mismatch"] = int(((uv[..., 0].view(np.uint32) != hits["u"].view(np.uint32)) | (uv[..., 1].view(np.uint32) != hits["v"].view(np.uint32)))[hit_mask].sum())
    # ray batch with instrumentation
    rays = random_rays(inst.scene.triangles, n_rays, 3)
    gh = pt_.TraceRays(rays, with_stats=True)
    oh = O.trace(osc, rays, c.stack_size)
    for k in ("ref_idx", "tri_id", "nodes", "tris", "hash", "max_depth"):
        res["rays_%s_mismatch" % k] = int((gh[k] != oh[k]).sum())
    for k in ("u", "v", "t"):
        res["rays_%s_mismatch" % k] = int((gh[k].view(np.uint32) != oh[k].view(np.uint32)).sum())
    gh2 = pt_.TraceRays(rays, with_stats=False)
    res["rays_nostats_vs_stats"] = int((gh2["tri_id"] != gh["tri_id"]).sum() + (gh2["t"].view(np.uint32) != gh["t"].view(np.uint32)).sum())
    # path tracing
    pt_.SetInstrumentation(timing=False, counters=True)
    pt_.ResetStats()
    pt_.Trace(True, spp)
    g = pt_.ReadResult()
    st = O.PathTracerState(c.width, c.height)
    ost = O.pt_frames(osc, P, O.shift_bytes(99, c.width, c.height), SOBOL, st, spp)
    diff = (st.accum[..., :3].view(np.uint32) != g.view(np.uint32)).any(axis=-1)
    res["pt_mismatch_px"] = int(diff.sum())
    if diff.sum():
        rel = np.abs(st.accum[..., :3] - g) / np.maximum(np.abs(st.accum[..., :3]), 1e-6)
        res["pt_max_rel"] = float(rel.max())
        ys, xs = np.where(diff)
        res["pt_first_bad"] = [int(xs[0]), int(ys[0]), st.accum[ys[0], xs[0], :3].tolist(), g[ys[0], xs[0]].tolist()]
    gs = pt_.GetStats()
    res["pt_stats_gpu"] = {k: int(gs[k]) for k in ("rays", "nodes_visited", "tris_tested", "hits", "shaded", "max_stack", "stack_overflows", "bad_materials")}
    res["pt_stats_oracle"] = {k: v for k, v in ost.as_dict().items() if k in ("rays", "nodes", "tris", "hits", "shaded", "max_depth")}
    pt_.SetInstrumentation(False, False)
    print(json.dumps(res))
    sys.stdout.flush()
    return inst


def timing(name, w, h, spp, pt=None):
    spec = scenes.make_scene(name, CACHE, width=w, height=h, pt=pt)
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer
    p.SetInstrumentation(timing=True, counters=False)
    out = {"scene": name, "res": [w, h]}
    p.Trace(False); p.ResetStats()
    t0 = time.time(); p.Trace(False); t1 = time.time()
    s = p.GetStats()
    out["primary"] = {"wall_ms": round((t1 - t0) * 1e3, 3), "trace_ms": round(s["trace_ms"], 3), "Mrays_s_kernel": round(s["rays"] / s["trace_ms"] / 1e3, 1)}
    p.Trace(True, 2); p.Reset(); p.ResetStats()
    t0 = time.time(); p.Trace(True, spp); t1 = time.time()
    s = p.GetStats()
    out["pt"] = {"spp": spp, "wall_ms": round((t1 - t0) * 1e3, 2), "rays": int(s["rays"]), "trace_ms": round(s["trace_ms"], 2), "shade_ms": round(s["shade_ms"], 2),
                 "Mrays_s_wall": round(s["rays"] / (t1 - t0) / 1e6, 1), "Mrays_s_trace_kernel": round(s["rays"] / s["trace_ms"] / 1e3, 1)}
    p.SetInstrumentation(False, True); p.Reset(); p.ResetStats()
    p.Trace(True, 2)
    s = p.GetStats()
    out["per_ray"] = {"nodes": round(s["nodes_visited"] / s["rays"], 2), "tris": round(s["tris_tested"] / s["rays"], 2), "hit_frac": round(s["hits"] / s["rays"], 3),
                      "max_stack": int(s["max_stack"]), "alg_bytes": round((80 * s["nodes_visited"] + 48 * s["tris_tested"] + 4 * s["hits"]) / s["rays"] + 48, 1)}
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    what = sys.argv[1:] or ["parity", "timing"]
    if "parity" in what:
        parity("tiny2", 96, 64, 3, 2000)
        parity("tiny1", 160, 90, 4, 20000)
        parity("tiny0", 160, 90, 20, 20000)
        parity("sibenik", 160, 90, 4, 20000)
        parity("sponza", 192, 108, 4, 20000)
        try:
            parity("tiny0", 100, 75, 3, 1000, pt={"stackSize": 1})  # forces the overflow report
        except Exception as e:  # noqa: BLE001
            print("expected overflow error:", e)
    if "timing" in what:
        timing("sponza", 1920, 1080, 16)
        timing("sibenik", 1920, 1080, 16)
