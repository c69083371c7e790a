#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): Mrays/s (primary + secondary) of the wavefront
path tracer on the Sponza-class scene, 1920x1080, 8 bounces, with the roofline of the dominant kernel (the CWBVH8
traversal) and a CPU baseline (the oracle's scalar traversal) timed on the same box.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one path-traced frame (1 spp) of the whole 1920x1080 image = one pass of the hot path over one batch:
camera rays -> [traversal -> shade/scatter] x 8 -> accumulate.  With N > 1 the frame is sharded by 32x32 pixel tile
over the ranks (zero communication while rendering) and the timed region ends with the single gather of the fp32
radiance on rank 0 (RCCL over xGMI, inside the library: adypt_comm_gather_radiance).  Total work is fixed as N grows ->
"scaling": "strong".  The launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT: the product needs no torch
(`--comm torch` keeps the torch.distributed variant: barrier / reductions / gather through torch, backend nccl = RCCL).

What the JSON line carries besides the contract's fields:
  roofline               the traversal kernel on the bench scene.  Its BVH (19 MB) is cache resident, so the HBM roof does not
                         bind; the kernel is bound by vector-ALU issue (with the CU's vector-memory pipeline close behind) ->
                         bound "valu_issue", achieved = vector-ALU issue cycles demanded per second (PMC instructions per ray x
                         measured rays/s of the kernel x the average architectural issue cycles of its instruction mix) against
                         1024 SIMDs x the clock the chip ran the kernel at (PMC); lane_util = fraction of the 64 lanes doing work in an issued instruction.  The algorithmic
                         HBM-read figure of SURVEY.md §8(d) and the measured fabric traffic are reported next to it.
  roofline_hbm_resident  the same kernel on the ~10 M-triangle stand-in of BASELINE config 4 (BVH 0.6 GB > 256 MB Infinity
                         Cache): here HBM binds -> bound "hbm", algorithmic bytes / HIP-event time / 8 TB/s.
  single_frame           one adypt_trace_spp(ctx, 1) per call (what Instance::Update does), with the library's look-ahead.
  cpu_baseline           oracle/liboracle.so on the host cores, bounded sample.

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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec)
N_SIMD = 1024           # 256 CUs x 4 SIMDs
NOMINAL_CLOCK_GHZ = 2.4 # only used when no profile supplies the clock the chip really ran the kernel at
PMC_BENCH, PMC_SANMIGUEL, ISSUE_MODEL = "r3_pmc_bench.json", "r3_pmc_sanmiguel.json", "r3_valu_issue_model.json"
PT_CFG = {"maxBounce": 8, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24}
SEED = 12345


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def counter_figures(name):
    """Per-ray counter figures of the traversal kernel from the committed rocprofv3 --pmc passes (tools/collect_profiles.sh ->
    tools/pmc_profile.py), or None with the reason when the file is missing or was measured on other device sources."""
    from tools.source_hash import device_source_hash
    pmc = load_profile(name)
    if not pmc:
        return None, "profiles/%s missing" % name
    now = device_source_hash()
    if pmc.get("source_hash") != now:
        return None, "profiles/%s was measured on device sources %s, this tree is %s: re-run tools/collect_profiles.sh" % (name, pmc.get("source_hash"), now)
    return pmc, None


def valu_roofline(pmc, model, kernel_rays_s, live_clock_ghz=0.0):
    """The vector-ALU issue roof of the traversal kernel.  Everything is a formula over the committed profiles and ONE live number:
      peak      = 1024 SIMDs x effective clock; effective clock = shader cycles / 100 MHz ticks sampled inside this run's traversal launches
                  (fallback: GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the profile run's kernel trace)
      achieved  = issue cycles the kernel's vector-ALU instructions need per second = SQ_INSTS_VALU per ray (profile) x rays/s (live) x the
                  average ARCHITECTURAL issue time of its instruction mix (2 cycles full-rate fp32 / logic / moves, 4 the other classes and
                  packed fp32, 8 transcendental: profiles/r3_valu_issue_model.json)
      frac      = achieved / peak: a fraction of a roof no instruction stream can exceed.
    Beside it: the same with the issue times measured on one-instruction loops (profiles/r3_valu_calibration.json, true cycles: 2.46 / 4.37 /
    4.33 / 8.24).  That figure exceeds 1 — the kernel's mixed stream issues faster than the weighted sum of single-class loops — so the loops
    are not a roof; it is printed because VERDICT r2 asked for both.  SQ_ACTIVE_INST_VALU is NOT used: the calibration shows it counts 1 per
    instruction (2 per transcendental) whatever the instruction's issue time."""
    # the clock: measured inside this run's own traversal launches (s_memtime / s_memrealtime, adypt_get_shader_clock) when available —
    # boxes of the pool hold 2.1-2.3 GHz under this kernel — otherwise the one of the profile run (GRBM_GUI_ACTIVE / 8 / kernel-trace duration)
    profile_clock = pmc.get("effective_clock_GHz")
    clock = live_clock_ghz if live_clock_ghz and live_clock_ghz > 0.5 else profile_clock
    peak = N_SIMD * (clock or NOMINAL_CLOCK_GHZ)
    inst_rate = pmc["valu_insts_per_ray"] * kernel_rays_s
    arch = model["avg_issue_cycles_per_inst_architectural"] if model else 4.0
    achieved = inst_rate * arch / 1e9
    frac = achieved / peak
    out = {"bound": "valu_issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "Gcycle/s", "frac": round(frac, 4),
           "effective_clock_GHz": round(clock, 3) if clock else None, "clock_source": "live: s_memtime / s_memrealtime inside this run's traversal launches" if clock == live_clock_ghz else "profile run",
           "effective_clock_GHz_in_profile_run": round(profile_clock, 3) if profile_clock else None, "lane_util": round(pmc["lane_util"], 4),
           "useful_frac": round(min(1.0, frac) * pmc["lane_util"], 4),
           "valu_insts_per_ray": round(pmc["valu_insts_per_ray"], 2), "valu_Ginst_s": round(inst_rate / 1e9, 1),
           "issue_cycles_per_inst_architectural": arch, "issue_cycles_available_per_inst": round(peak * 1e9 / inst_rate, 3),
           "pmc_stale": False, "pmc_source_hash": pmc["source_hash"]}
    if model:
        loops = model["avg_issue_cycles_per_inst_single_class_loops"]
        out["issue_cycles_per_inst_single_class_loops"] = loops
        out["frac_at_single_class_loop_rates"] = round(inst_rate * loops / 1e9 / peak, 4)
        if "vmem_rd_insts_per_ray" in pmc:  # the co-limiter: the CU's one vector-memory pipeline (profiles/r2_ablations_k_trace.txt)
            out["vmem_busy_est"] = round(pmc["vmem_rd_insts_per_ray"] * kernel_rays_s * model["vmem_cycles_per_load_inst"] / (256 * (clock or NOMINAL_CLOCK_GHZ) * 1e9), 3)
    return out


def traffic_fields(pmc, rays, launches, kernel_rays_s):
    """Fabric-side bytes from the PMC counters: FETCH_SIZE (KB; x 2 on gfx950 for 16-byte-per-lane loads, MI355X_MICROARCH.md) + WRITE_SIZE,
    per ray (profile) x the rays per launch of this run; also without the x 2 (the correction is calibrated on coalesced streaming)."""
    return {"traffic": round(pmc["traffic_bytes_per_ray"] * rays / max(1, launches)),
            "traffic_GBs": round(pmc["traffic_bytes_per_ray"] * kernel_rays_s / 1e9, 1),
            "traffic_frac_of_hbm_peak": round(pmc["traffic_bytes_per_ray"] * kernel_rays_s / 1e9 / HBM_PEAK_GBS, 4),
            "traffic_uncorrected": round(pmc.get("traffic_bytes_per_ray_uncorrected", 0.0) * rays / max(1, launches)),
            "traffic_uncorrected_frac_of_hbm_peak": round(pmc.get("traffic_bytes_per_ray_uncorrected", 0.0) * kernel_rays_s / 1e9 / HBM_PEAK_GBS, 4),
            "l2_hit_rate": round(pmc.get("TCC_hit_rate", 0.0), 3)}


def census(pt, steps, warmup, expect_rays=None):
    """The same K frames again through the instrumented traversal -> exact node / triangle counts -> algorithmic bytes
    (per ray 80 B x nodes visited + 48 B x triangles tested + 4 B x hit remap + 32 B ray read + 16 B hit write)."""
    pt.Reset()
    pt.SetInstrumentation(timing=False, counters=True)
    if warmup:
        pt.Trace(True, warmup)
    pt.ResetStats()
    pt.Trace(True, steps)
    cs = pt.GetStats()
    pt.SetInstrumentation(False, False)
    if expect_rays is not None:
        assert cs["rays"] == expect_rays, "census pass traced a different number of rays"
    cs["alg_bytes"] = 80 * cs["nodes_visited"] + 48 * cs["tris_tested"] + 4 * cs["hits"] + 48 * cs["rays"]
    return cs


def hbm_resident_block(args, dev):
    """roofline_hbm_resident: the traversal kernel where HBM binds (BASELINE config 4 stand-in, BVH 0.6 GB)."""
    from adypt_amd import api, scenes
    t0 = time.time()
    spec = scenes.make_scene("sanmiguel", args.cache, width=1920, height=1080, pt=dict(PT_CFG, tmpLifetime=16))
    inst = api.Instance()
    assert inst.InitializeFromFile(spec.config_path, shift_seed=SEED, device=dev), api.InstanceConfig.last_error()
    setup_s = time.time() - t0
    pt = inst.m_path_tracer
    steps, warmup = 32, 16
    pt.SetInstrumentation(timing=True, counters=False)
    pt.Trace(True, warmup)
    pt.ResetStats()
    t1 = time.perf_counter()
    pt.Trace(True, steps)
    wall = time.perf_counter() - t1
    st = pt.GetStats()
    live_clock = pt.GetShaderClockGHz()
    cs = census(pt, steps, warmup, st["rays"])
    bvh_mb = (len(inst.bvh.nodes) + len(inst.bvh.tri_indices) * 52) / 1e6
    achieved = cs["alg_bytes"] / (st["trace_ms"] * 1e-3) / 1e9
    rays_s = st["rays"] / (st["trace_ms"] * 1e-3)
    out = {"kernel": "k_trace<false, false>",
           "workload": "sanmiguel-like procedural stand-in (%s), %d triangles, BVH %.0f MB (nodes + Woop + index) > 256 MB Infinity Cache, 1920x1080, 8 bounces, %d frames after %d warm-up"
                       % (spec.label, inst.scene.n_tris, bvh_mb, steps, warmup),
           "alg_GBs": round(achieved, 1), "alg_frac_of_hbm_peak": round(achieved / HBM_PEAK_GBS, 4), "hbm_peak_GBs": HBM_PEAK_GBS,
           "launches": int(st["trace_launches"]), "avg_launch_ms": round(st["trace_ms"] / max(1, st["trace_launches"]), 4),
           "alg_bytes_per_launch": round(cs["alg_bytes"] / max(1, st["trace_launches"])), "alg_bytes_per_ray": round(cs["alg_bytes"] / cs["rays"], 1),
           "nodes_per_ray": round(cs["nodes_visited"] / cs["rays"], 2), "tris_per_ray": round(cs["tris_tested"] / cs["rays"], 2),
           "trace_kernel_Mrays_s": round(rays_s / 1e6, 1), "whole_frame_Mrays_s": round(st["rays"] / wall / 1e6, 1),
           "trace_kernels_ms": round(st["trace_ms"], 2), "shade_kernels_ms": round(st["shade_ms"], 2), "setup_s": round(setup_s, 1)}
    pmc, why = counter_figures(PMC_SANMIGUEL)
    if pmc:
        out.update(traffic_fields(pmc, st["rays"], st["trace_launches"], rays_s))
        valu = valu_roofline(pmc, load_profile(ISSUE_MODEL), rays_s, live_clock)
        # what binds here is neither roof alone: the fraction reported is the counter-measured fabric traffic against the HBM peak — a valid
        # fraction (an upper bound on HBM bytes: Infinity-Cache hits are in it); the algorithmic figure exceeds what reaches the fabric because
        # the L2s catch the top of the tree, and the vector ALUs are busy for valu_issue_frac of the cycles
        out.update({"bound": "between hbm and valu_issue", "achieved": out["traffic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": out["traffic_frac_of_hbm_peak"], "valu_issue_frac": valu["frac"], "lane_util": valu["lane_util"],
                    "effective_clock_GHz": valu["effective_clock_GHz"], "pmc_stale": False,
                    "traffic_over_algorithmic": round(pmc["traffic_bytes_per_ray"] / (cs["alg_bytes"] / cs["rays"]), 3),
                    "note": "achieved / frac = fabric-side bytes of the traversal launches (PMC FETCH_SIZE x 2 + WRITE_SIZE of %s, per ray) x this run's rays/s "
                            "against 8 TB/s; traffic_uncorrected = the same without the x 2.  alg_frac_of_hbm_peak (SURVEY.md 8d bytes / HIP-event time / 8 TB/s) "
                            "is NOT a fraction of HBM traffic: the L2s catch %.0f %% of the requests.  valu_issue_frac: bench.py valu_roofline()"
                            % (pmc.get("command", "?"), 100 * pmc.get("TCC_hit_rate", 0.0))})
    else:
        out.update({"bound": "hbm (algorithmic bytes only)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "pmc_stale": True, "note": "no counter figures: " + why + "; alg_frac_of_hbm_peak is cache assisted and not reported as a fraction"})
    pt.destroy()
    return out


def cpu_baseline(inst, c):
    """The oracle's scalar traversal + shading of the same workload on the host cores, bounded sample (kind "port")."""
    from oracle import oracle_py as O
    osc = O.Scene(inst.bvh.nodes, inst.bvh.tri_indices, inst.scene.triangles, inst.scene.materials, textures=inst.scene.textures)
    ip, iv = O.camera(c.fov, c.yaw, c.pitch, c.width, c.height)
    P = O.make_params(c.width, c.height, list(c.position), ip, iv, stack_size=c.stack_size, max_bounce=c.max_bounce, subpixel=c.subpixel,
                      tmp_life=c.tmp_lifetime, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))  # the bench's own tmpLifetime: same rays per frame
    sm = np.fromfile(os.path.join(ROOT, "tests", "golden", "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)
    cores = O.default_threads()
    stc = O.PathTracerState(c.width, c.height)
    shift = O.shift_bytes(SEED, c.width, c.height)
    cpu_rays, cpu_t, frames = 0, 0.0, 0
    while cpu_t < 10.0 and frames < 16:
        t1 = time.perf_counter()
        s = O.pt_frames(osc, P, shift, sm, stc, 1, n_threads=cores)
        cpu_t += time.perf_counter() - t1
        cpu_rays += s.rays
        frames += 1
    # one thread (SURVEY.md §8d asks for both): a quarter-height frame keeps it to a few seconds
    P1 = O.make_params(c.width, c.height // 4, list(c.position), *O.camera(c.fov, c.yaw, c.pitch, c.width, c.height // 4), stack_size=c.stack_size,
                       max_bounce=c.max_bounce, subpixel=c.subpixel, tmp_life=1, tmin=c.ray_tmin, clamp=c.clamp, sun=list(c.sun))  # one frame: it traces its primaries
    st1 = O.PathTracerState(c.width, c.height // 4)
    t1 = time.perf_counter()
    s1 = O.pt_frames(osc, P1, O.shift_bytes(SEED, c.width, c.height // 4), sm, st1, 1, n_threads=1)
    t1 = time.perf_counter() - t1
    return {"value": round(cpu_rays / cpu_t / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames starting at frame 0 with the bench's tmpLifetime %d (cached primaries are not rays), %d rays, %.1f s, oracle/liboracle.so on %d threads"
                      % (frames, c.width, c.height, c.tmp_lifetime, cpu_rays, cpu_t, cores),
            "value_1_thread": round(s1.rays / t1 / 1e6, 3),
            "sample_1_thread": "one %dx%d frame, %d rays, %.1f s" % (c.width, c.height // 4, s1.rays, t1)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tmp-lifetime", type=int, default=16, help="reference default 16: primary hits are re-traced every 16th frame")
    ap.add_argument("--comm", default=os.environ.get("ADYPT_BENCH_COMM", "native"), choices=["native", "torch"],
                    help="N > 1: 'native' = the library's own RCCL communicator (no torch); 'torch' = torch.distributed (nccl, or gloo with ADYPT_BENCH_BACKEND=gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-block", action="store_true", help="skip roofline_hbm_resident (the 10 M-triangle scene: ~25 s of setup)")
    ap.add_argument("--no-single-frame", action="store_true")
    ap.add_argument("--cache", default=os.environ.get("ADYPT_CACHE", os.path.join(ROOT, ".adypt_cache")))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    use_torch = world > 1 and args.comm == "torch"
    dev = int(os.environ.get("ADYPT_BENCH_DEVICE", local_rank))  # the override only to rehearse N ranks on a 1-GPU box
    dist = torch = None

    def init_torch():
        nonlocal dist, torch, dev
        import torch
        import torch.distributed as dist
        dev = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ADYPT_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse the N>1 flow on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if use_torch:
        init_torch()

    from adypt_amd import api, distributed as D, scenes, _native as N

    # ---- scene: every rank generates + builds its own copy (0.8 s; nothing to wait for, no shared cache to race on) ------------
    pt_cfg = dict(PT_CFG, tmpLifetime=args.tmp_lifetime)
    cache = args.cache if world == 1 else os.path.join(args.cache, "rank%d" % rank)
    t_setup = time.time()
    spec = scenes.make_scene(args.scene, cache, width=args.width, height=args.height, pt=pt_cfg)
    inst = api.Instance()
    try:
        ok = inst.InitializeFromFile(spec.config_path, shift_seed=SEED, device=dev, tile_rank=rank, tile_nranks=world)
    except N.AdyptError as e:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path (%s)" % e)
    assert ok, api.InstanceConfig.last_error()
    pt = inst.m_path_tracer
    c = inst.m_config.c
    fif = pt.GetFramesInFlight()
    t_setup = time.time() - t_setup

    if world > 1 and not use_torch:
        # Two failure classes.  (a) rank 0 cannot even make an RCCL id (library not found, symbol missing): it leaves a marker in the
        # rendezvous file, EVERY rank sees the same RuntimeError, and all of them take the torch.distributed variant of the same gather
        # together.  (b) anything later (ncclCommInitRank on one rank, a timeout) may have happened on this rank alone while the others
        # sit inside RCCL: falling back here would hang the job, so the rank exits non-zero and the launcher tears the job down.
        try:
            uid = D.exchange_unique_id(rank, world)
        except RuntimeError as e:
            sys.stderr.write("bench.py rank %d: no native RCCL id (%s); all ranks fall back to --comm torch\n" % (rank, e))
            use_torch = True
            init_torch()
        else:
            try:
                pt.CommInit(uid)  # the ranks' only exchange outside RCCL was the 128-byte communicator id
                pt.CommBarrier()
            except (N.AdyptError, OSError) as e:
                raise SystemExit("bench.py rank %d: native RCCL communicator failed (%s)" % (rank, e))
            if rank == 0:  # every rank holds the communicator now: a later job must never find this id
                try:
                    os.remove(D.rendezvous_path())
                except OSError:
                    pass

    if use_torch:
        n_pad = D.max_block_count(c.width, c.height, world) * D.BLOCK_PIXELS * 4
        gather_buf = torch.zeros(n_pad, dtype=torch.float32, device="cuda")
        on_device = dist.get_backend() == "nccl"  # gloo rehearsal: host tensors, host un-tiling

        def barrier():
            torch.cuda.synchronize()
            dist.barrier()

        def gather():
            pt.copy_local_radiance(gather_buf.data_ptr(), n_pad // 4)
            if on_device:
                return D.gather_radiance_device(gather_buf, pt, c.width, c.height, rank, world)
            return D.gather_radiance(gather_buf, c.width, c.height, rank, world)
    else:
        def barrier():
            # every rank's GPU drained (hipDeviceSynchronize), then all ranks met (RCCL all-reduce + drain)
            pt.DeviceSynchronize()
            if world > 1:
                pt.CommBarrier()

        def gather():
            return pt.CommGatherDevice()  # the one collective of the data path; rank 0: the assembled image, resident in HBM

    # ---- warmup ------------------------------------------------------------------------------------------------------
    pt.SetInstrumentation(timing=True, counters=False)
    if args.warmup:
        pt.Trace(True, args.warmup)
    gather()  # warms the communicator
    rays_warmup = int(pt.GetStats()["rays"])
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
    st = pt.GetStats()
    live_clock = pt.GetShaderClockGHz()  # the clock the chip held under the traversal launches of the timed region
    # per-rank breakdown, so that a scaling run can be diagnosed from its own line: every rank fills its own slots, one sum all-reduce
    # hands every rank the whole table (5 x N doubles)
    mine = [elapsed * 1e3, float(st["trace_ms"]), float(st["shade_ms"]), gather_ms, float(st["rays"])]
    per_rank = None
    if use_torch:
        if image is not None and on_device:
            image = image.cpu().numpy()
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        rays = torch.tensor([int(st["rays"])], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        elapsed, total_rays = float(tt.item()), int(rays.item())
        table = torch.zeros(5 * world, dtype=torch.float64, device=red_dev)
        table[5 * rank:5 * rank + 5] = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        per_rank = table.cpu().numpy().reshape(world, 5)
    else:
        if world > 1:
            flat = [0.0] * (5 * world)
            flat[5 * rank:5 * rank + 5] = mine
            per_rank = np.array(pt.CommAllReduce(flat, "sum")).reshape(world, 5)
            elapsed = pt.CommAllReduce([elapsed], "max")[0]          # MAX over ranks
            total_rays = int(round(pt.CommAllReduce([float(st["rays"])], "sum")[0]))  # exact: < 2^53
        else:
            total_rays = int(st["rays"])
        image = pt.CommReadResult()  # untimed (collective): the image on the host for the checksum

    # ---- census (untimed): the same K frames again through the instrumented traversal -> exact algorithmic bytes ------
    trace_ms, trace_launches, shade_ms = st["trace_ms"], st["trace_launches"], st["shade_ms"]
    cs = census(pt, args.steps, args.warmup, st["rays"])
    alg_bytes = cs["alg_bytes"]
    alg_gbs = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    kernel_rays_s = st["rays"] / (trace_ms * 1e-3) if trace_ms > 0 else 0.0
    bvh_mb = (len(inst.bvh.nodes) + len(inst.bvh.tri_indices) * 52) / 1e6
    roofline = {"kernel": "k_trace<false, false>", "launches": int(trace_launches), "avg_launch_ms": round(trace_ms / max(1, trace_launches), 4),
                "trace_kernel_Mrays_s": round(kernel_rays_s / 1e6, 1),
                "alg_bytes_per_launch": round(alg_bytes / max(1, trace_launches)), "alg_bytes_per_ray": round(alg_bytes / max(1, cs["rays"]), 1),
                "nodes_per_ray": round(cs["nodes_visited"] / max(1, cs["rays"]), 2), "tris_per_ray": round(cs["tris_tested"] / max(1, cs["rays"]), 2),
                "alg_GBs": round(alg_gbs, 1), "alg_frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "hbm_peak_GBs": HBM_PEAK_GBS,
                "traffic": None, "shader_clock_GHz_live": round(live_clock, 3)}
    # The PMC counters cannot be read from inside this process: per-ray figures come from the committed rocprofv3 --pmc passes over
    # this very command line (tools/collect_profiles.sh -> profiles/r3_pmc_bench.json), hash-checked against the device sources of this
    # tree, x the rays / time measured here.  N > 1: rank 0 traces an interleaved 1/N of the same pixels with the same kernel.
    pmc, why = counter_figures(PMC_BENCH) if (args.scene, args.width, args.height) == ("sponza", 1920, 1080) else (None, "no committed counter passes for this scene / size")
    if pmc:
        roofline.update(valu_roofline(pmc, load_profile(ISSUE_MODEL), kernel_rays_s, live_clock))
        roofline.update(traffic_fields(pmc, st["rays"], trace_launches, kernel_rays_s))
        roofline["note"] = ("BVH (nodes + Woop + index) %.0f MB is L2 / Infinity-Cache resident: alg_frac_of_hbm_peak may exceed 1 and is not a fraction of HBM "
                            "traffic (traffic = what the counters saw on the fabric).  What binds is vector-ALU issue, the CU's vector-memory pipeline close behind "
                            "(vmem_busy_est; profiles/r2_ablations_k_trace.txt).  peak = 1024 SIMDs x effective_clock_GHz (GRBM_GUI_ACTIVE / 8 / launch duration of "
                            "the kernel trace); achieved = SQ_INSTS_VALU per ray (%s) x trace_kernel_Mrays_s of this run x issue_cycles_per_inst_architectural "
                            "(the mix's 2 / 4 / 4 / 8-cycle classes, profiles/%s); frac_at_single_class_loop_rates = the same with the class times measured on "
                            "one-instruction loops (profiles/r3_valu_calibration.json), > 1 because a mixed stream issues faster than those loops: not a roof.  "
                            "lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)%s"
                            % (bvh_mb, pmc.get("command", "?"), ISSUE_MODEL, "; per-ray figures of the 1-GPU passes applied to rank 0's shard" if world > 1 else ""))
    else:
        roofline.update({"bound": "hbm (algorithmic bytes; the scene is cache resident)", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_gbs / HBM_PEAK_GBS, 4), "pmc_stale": True,
                         "note": "no counter figures: " + why + ".  Fallback to SURVEY.md 8d algorithmic bytes / HIP-event time / 8 TB/s, which for this "
                                 "L2 / Infinity-Cache resident BVH (%.0f MB) can exceed 1 and is not a fraction of HBM traffic" % bvh_mb})

    # The HBM-read roof of the metric's name, for the same kernel and launches: a valid fraction needs bytes that really travelled, so
    # achieved = the counters' fabric-side traffic (an upper bound on HBM bytes: Infinity-Cache hits are in it); the algorithmic figure
    # of SURVEY.md 8d is reported next to it and exceeds the peak on this cache-resident scene.
    roofline_hbm = {"bound": "hbm", "kernel": "k_trace<false, false>", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": roofline.get("traffic_GBs"), "frac": roofline.get("traffic_frac_of_hbm_peak"), "traffic": roofline.get("traffic"),
                    "algorithmic_GBs": round(alg_gbs, 1), "algorithmic_over_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": roofline["alg_bytes_per_launch"],
                    "note": "achieved / frac from PMC traffic (None when the committed counter profile is stale); algorithmic_over_peak is not a fraction of HBM traffic: "
                            "the 20 MB BVH is served by the L2s (hit rate in roofline.l2_hit_rate) and the Infinity Cache"}

    # ---- one frame per call (Instance::Update -> Trace(true), src/Instance.cpp:44-57) with the library's look-ahead ----------
    single = None
    if world == 1 and not args.no_single_frame:
        n_calls = 64
        pt.Reset()
        pt.SetInstrumentation(False, False)
        pt.SetLookahead(True)
        for _ in range(pt.GetFramesInFlight()):
            pt.Trace(True, 1)  # warm-up: one whole pass handed out
        pt.ResetStats()
        pt.DeviceSynchronize()
        t1 = time.perf_counter()
        for _ in range(n_calls):
            pt.Trace(True, 1)
        pt.DeviceSynchronize()
        dt = time.perf_counter() - t1
        s1 = pt.GetStats()
        pt.SetLookahead(False)
        pt.SetFramesInFlight(1)  # and without look-ahead, one frame per wavefront pass: what round 1's binding did
        pt.Reset()
        pt.Trace(True, 16)
        pt.ResetStats()
        t1 = time.perf_counter()
        pt.Trace(True, 16)
        dt0 = time.perf_counter() - t1
        s0 = pt.GetStats()
        single = {"calls": n_calls, "ms_per_call": round(dt * 1e3 / n_calls, 4), "Mrays_s": round(s1["rays"] / dt / 1e6, 1),
                  "frac_of_batched": round((s1["rays"] / dt) / (total_rays / elapsed), 3),
                  "without_lookahead_Mrays_s": round(s0["rays"] / dt0 / 1e6, 1),
                  "note": "adypt_trace_spp(ctx, 1) per call, adypt_set_lookahead on: a call that needs untraced frames traces a whole pass of %d, the following calls only apply their running-mean step" % fif}

    # ---- the kernel where HBM binds, and the CPU baseline ----------------------------------------------------------------------
    hbm = None
    if rank == 0 and world == 1 and not args.no_hbm_block and args.scene == "sponza":
        pt.destroy()  # frees the bench scene's queues first
        hbm = hbm_resident_block(args, dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(inst, c)

    if rank == 0:
        value = total_rays / elapsed / 1e6
        out = {"metric": "Mrays/sec (primary+secondary) Sponza 1920x1080 8-bounce; % HBM-read roofline",
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed * 1e3 / max(1, args.steps), 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s-like procedural stand-in (%s), %d triangles, %dx%d, full wavefront path trace, maxBounce %d, tmpLifetime %d, 1 spp per step; one radiance gather per run"
                                      % (args.scene, spec.label, inst.scene.n_tris, c.width, c.height, c.max_bounce, c.tmp_lifetime),
                          "rays_per_step": round(total_rays / max(1, args.steps)), "tile_shard": "32x32 blocks, owner (bx+by) mod N",
                          "frames_in_flight": fif, "rays_warmup": rays_warmup, "comm": ("torch.distributed" if use_torch else ("host-staged TEST transport (ADYPT_COMM_TRANSPORT=host: not a measurement)" if os.environ.get("ADYPT_COMM_TRANSPORT") == "host" else "native RCCL")) if world > 1 else "none",
                          "setup_s": round(t_setup, 2)},
               "roofline": roofline, "roofline_hbm": roofline_hbm, "roofline_hbm_resident": hbm, "cpu_baseline": cpu, "single_frame": single,
               "gather_ms": round(gather_ms, 3), "shade_kernels_ms": round(shade_ms, 2), "trace_kernels_ms": round(trace_ms, 2),
               "per_rank": None if per_rank is None else {
                   "wall_ms": [round(float(v), 3) for v in per_rank[:, 0]], "trace_kernels_ms": [round(float(v), 3) for v in per_rank[:, 1]],
                   "other_kernels_ms": [round(float(v), 3) for v in per_rank[:, 2]], "gather_ms": [round(float(v), 3) for v in per_rank[:, 3]],
                   "rays": [int(v) for v in per_rank[:, 4]],
                   # share of a rank's wall time not inside a tracing kernel or the gather: launch tails show up in the kernels, host gaps here
                   "outside_kernels_frac": [round(float(1.0 - (r[1] + r[2] + r[3]) / max(r[0], 1e-9)), 3) for r in per_rank],
                   "note": "wall = this rank's timed region (K steps + its part of the gather); rank 0's gather includes waiting for the slowest peer"},
               "image_mean": float(image.mean()) if image is not None else None}
        print(json.dumps(out))
        sys.stdout.flush()
    if use_torch:
        dist.barrier()
        dist.destroy_process_group()
    elif world > 1:
        pt.CommBarrier()


if __name__ == "__main__":
    main()
