#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): Mrays/s (primary + secondary) of the wavefront
path tracer on the Sponza-class scene, 1920x1080, 8 bounces, with the % of the HBM-read roofline of the dominant
kernel (the CWBVH8 traversal) and a CPU baseline (the oracle's scalar traversal) timed on the same box.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one path-traced frame (1 spp) of the whole 1920x1080 image = one pass of the hot path over one batch:
camera rays -> [traversal -> shade/scatter] x 8 -> accumulate.  With N > 1 the frame is sharded by 32x32 pixel tile
over the ranks (zero communication while rendering) and the timed region ends with the single gather of the fp32
radiance on rank 0 (RCCL over xGMI).  Total work is fixed as N grows -> "scaling": "strong".

Scene: the real sponza.obj is not available anywhere (no network); a deterministic procedural stand-in of the same
triangle count is generated, written as OBJ/MTL + Adypt .config and loaded through the product's own
loader -> SBVH -> CWBVH8 path (adypt_amd/scenes.py).  `$ADYPT_ASSETS/sponza.obj` is used instead when present.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tmp-lifetime", type=int, default=16, help="reference default 16: primary hits are re-traced every 16th frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache", default=os.environ.get("ADYPT_CACHE", os.path.join(ROOT, ".adypt_cache")))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    dev = local_rank % max(1, torch.cuda.device_count())  # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ADYPT_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse the N>1 flow on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from adypt_amd import api, distributed as D, scenes

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- scene: rank 0 generates + builds the BVH cache, the others load it ------------------------------------
    pt_cfg = {"maxBounce": 8, "subpixel": 8, "tmpLifetime": args.tmp_lifetime, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24}
    t_setup = time.time()
    inst = api.Instance()
    if rank == 0:
        spec = scenes.make_scene(args.scene, args.cache, width=args.width, height=args.height, pt=pt_cfg)
        ok = inst.InitializeFromFile(spec.config_path, shift_seed=12345, device=dev, tile_rank=rank, tile_nranks=world)
        assert ok, api.InstanceConfig.last_error()
    barrier()
    if rank != 0:
        spec = scenes.make_scene(args.scene, args.cache, width=args.width, height=args.height, pt=pt_cfg)
        ok = inst.InitializeFromFile(spec.config_path, shift_seed=12345, device=dev, tile_rank=rank, tile_nranks=world)
        assert ok, api.InstanceConfig.last_error()
    pt = inst.m_path_tracer
    c = inst.m_config.c
    t_setup = time.time() - t_setup
    n_pad = D.max_block_count(c.width, c.height, world) * D.BLOCK_PIXELS * 4
    gather_buf = torch.zeros(n_pad, dtype=torch.float32, device="cuda")

    # ---- warmup ------------------------------------------------------------------------------------------------------
    pt.SetInstrumentation(timing=True, counters=False)
    if args.warmup:
        pt.Trace(True, args.warmup)
    on_device = world == 1 or dist.get_backend() == "nccl"  # gloo rehearsal: host tensors, host un-tiling

    def gather():
        pt.copy_local_radiance(gather_buf.data_ptr(), n_pad // 4)
        if on_device:
            return D.gather_radiance_device(gather_buf, pt, c.width, c.height, rank, world)
        return D.gather_radiance(gather_buf, c.width, c.height, rank, world)

    gather()  # warms the communicator
    pt.ResetStats()

    # ---- timed region: exactly K steps + the one gather ------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    pt.Trace(True, args.steps)
    t_gather = time.perf_counter()
    image = gather()  # rank 0: the assembled W x H x 3 radiance, resident in HBM (as the reference's result texture is)
    barrier()
    elapsed = time.perf_counter() - t0
    gather_ms = (time.perf_counter() - t_gather) * 1e3
    if image is not None and on_device:
        image = image.cpu().numpy()
    st = pt.GetStats()
    red_dev = "cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    rays = torch.tensor([int(st["rays"])], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    total_rays = int(rays.item())

    # ---- census (untimed): the same K frames again through the instrumented traversal -> exact algorithmic bytes ------
    trace_ms, trace_launches, shade_ms = st["trace_ms"], st["trace_launches"], st["shade_ms"]
    pt.Reset()
    pt.SetInstrumentation(timing=False, counters=True)
    if args.warmup:
        pt.Trace(True, args.warmup)
    pt.ResetStats()
    pt.Trace(True, args.steps)
    cs = pt.GetStats()
    assert cs["rays"] == st["rays"], "census pass traced a different number of rays"
    pt.SetInstrumentation(False, False)
    # per ray 80 B x nodes visited + 48 B x triangles tested + 4 B x hit remap + 32 B ray read + 16 B hit write
    alg_bytes = 80 * cs["nodes_visited"] + 48 * cs["tris_tested"] + 4 * cs["hits"] + 48 * cs["rays"]
    achieved = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_trace<false, false>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "launches": int(trace_launches), "avg_launch_ms": round(trace_ms / max(1, trace_launches), 4),
                "alg_bytes_per_launch": round(alg_bytes / max(1, trace_launches)), "alg_bytes_per_ray": round(alg_bytes / max(1, cs["rays"]), 1),
                "nodes_per_ray": round(cs["nodes_visited"] / max(1, cs["rays"]), 2), "tris_per_ray": round(cs["tris_tested"] / max(1, cs["rays"]), 2),
                "trace_kernel_Mrays_s": round(st["rays"] / (trace_ms * 1e3), 1) if trace_ms > 0 else None,
                "note": "algorithmic bytes / HIP-event time of the traversal launches of rank 0.  The BVH (nodes+Woop %.0f MB) is L2/Infinity-Cache "
                        "resident, so most algorithmic bytes never reach HBM (see traffic / traffic_GBs) and frac can exceed 1; the kernel is bound "
                        "by vector-ALU issue (profiles/: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = its duration)" %
                        ((len(inst.bvh.nodes) + len(inst.bvh.tri_indices) * 48) / 1e6)}

    # `traffic`: HBM/fabric bytes per launch of the same kernel from the PMC counters (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024,
    # gfx950 corrections of MI355X_MICROARCH.md).  Counters cannot be read from inside this process; the value comes
    # from the committed rocprofv3 --pmc passes over this very command (tools/collect_profiles.sh -> profiles/r1_pmc_traffic.json).
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
        if world == 1 and (args.scene, args.width, args.height) == ("sponza", 1920, 1080):
            roofline["traffic"] = round(pmc["traffic_bytes_per_ray"] * st["rays"] / max(1, trace_launches))
            # the same traffic as a rate: what the fabric (Infinity Cache + HBM) actually delivered while the kernel ran
            roofline["traffic_GBs"] = round(pmc["traffic_bytes_per_ray"] * st["rays"] / (trace_ms * 1e-3) / 1e9, 1) if trace_ms > 0 else None
            roofline["traffic_source"] = ("profiles/r1_pmc_traffic.json: %.1f fabric bytes per ray (separate rocprofv3 --pmc passes of bench.py --steps 64 --warmup 0) "
                                          "x the rays per launch of this run; L2 hit rate %.2f" % (pmc["traffic_bytes_per_ray"], pmc["TCC_hit_rate"]))
    except (OSError, KeyError, ValueError):
        pass

    # ---- CPU baseline: the oracle's scalar traversal + shading of the same workload, bounded sample ---------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O
        osc = O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)
        ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
        P = O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce, subpixel=c.subpixel,
                          tmp_life=1, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))
        sm = np.fromfile(os.path.join(ROOT, "tests", "golden", "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)
        cores = O.default_threads()
        stc = O.PathTracerState(c.width, c.height)
        shift = O.shift_bytes(12345, c.width, c.height)
        cpu_rays, cpu_t, frames = 0, 0.0, 0
        while cpu_t < 10.0 and frames < 16:
            t1 = time.perf_counter()
            s = O.pt_frames(osc, P, shift, sm, stc, 1, n_threads=cores)
            cpu_t += time.perf_counter() - t1
            cpu_rays += s.rays
            frames += 1
        # one thread (SURVEY.md §8d asks for both): a quarter-height frame keeps it to a few seconds
        P1 = O.make_params(c.width, c.height // 4, list(c.position), *O.camera(c.fov, c.yaw, c.pitch, c.width, c.height // 4), stack_size=c.stack_size,
                           max_bounce=c.max_bounce, subpixel=c.subpixel, tmp_life=1, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))
        st1 = O.PathTracerState(c.width, c.height // 4)
        t1 = time.perf_counter()
        s1 = O.pt_frames(osc, P1, O.shift_bytes(12345, c.width, c.height // 4), sm, st1, 1, n_threads=1)
        t1 = time.perf_counter() - t1
        cpu = {"value": round(cpu_rays / cpu_t / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": "%d full %dx%d frames (every frame traces its primaries), %d rays, %.1f s, oracle/liboracle.so on %d threads"
                         % (frames, c.width, c.height, cpu_rays, cpu_t, cores),
               "value_1_thread": round(s1.rays / t1 / 1e6, 3),
               "sample_1_thread": "one %dx%d frame, %d rays, %.1f s" % (c.width, c.height // 4, s1.rays, t1)}

    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {"metric": "Mrays/sec (primary+secondary) Sponza 1920x1080 8-bounce; % HBM-read roofline",
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed * 1e3 / max(1, args.steps), 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s-like procedural stand-in (%s), %d triangles, %dx%d, full wavefront path trace, maxBounce %d, tmpLifetime %d, 1 spp per step; one radiance gather per run"
                                      % (args.scene, spec.label, inst.scene.n_tris, c.width, c.height, c.max_bounce, c.tmp_lifetime),
                          "rays_per_step": round(total_rays / max(1, args.steps)), "tile_shard": "32x32 blocks, owner (bx+by) mod N", "frames_in_flight": pt.GetFramesInFlight(), "setup_s": round(t_setup, 2)},
               "roofline": roofline, "cpu_baseline": cpu,
               "gather_ms": round(gather_ms, 3), "shade_kernels_ms": round(shade_ms, 2), "trace_kernels_ms": round(trace_ms, 2),
               "image_mean": float(image.mean()) if image is not None else None}
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
