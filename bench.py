#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): Mrays/s (primary + secondary) of the wavefront path tracer on the
Sponza-class scene, 1920x1080, 8 bounces, with the roofline of the dominant kernel and a CPU baseline (the oracle's scalar traversal)
timed on the same box.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one path-traced frame (1 spp) of the whole 1920x1080 image = one pass of the hot path over one batch: camera rays /
cached primary hits -> bounce 0 (k_shade_first) -> every further bounce in ONE launch (k_path: traversal + shading) -> accumulate.
The timed region (W untimed warm-up steps, then exactly K steps between barriers) is repeated R times from the same start frame;
`value` is the median, the spread is printed beside it.

N > 1: the frame is sharded by 32x32 pixel tile over the GPUs (zero communication while rendering) and the timed region ends with the
single gather of the fp32 radiance on GPU 0 (RCCL over xGMI, inside the library).  Before anything is timed the assembled N-GPU image of two frames
is compared bit for bit with the 1-GPU image (exit 3 on a mismatch; --no-selfcheck skips it).  Total work is fixed as N grows -> "scaling": "strong".
Two ways to get N GPUs, the same tile shard and gather either way:
  * started by a launcher (WORLD_SIZE = N): one process per GPU, adypt_comm_* (ncclCommInitRank);
  * started plainly as `python bench.py --gpus N`: ONE process drives the N devices through adypt_create_multi (ncclCommInitAll) — and
    exits non-zero if the box has fewer than N devices.  It never falls back to fewer GPUs than asked for.

What the JSON line carries besides the contract's fields:
  roofline               the dominant kernel of the timed region (k_path<false>: 98 % of the GPU time): `bound` names what binds it ("valu_issue" on this
                         scene); achieved / frac / traffic are what the metric's name asks for and `frac_of` says so: fabric-side bytes the counters saw
                         per ray (committed rocprofv3 --pmc passes of this very command, hash-checked against the device sources) x this run's rays/s of
                         the kernel (HIP events) / 8 TB/s; the algorithmic-bytes figure of SURVEY.md 8(d) beside it (it exceeds the peak on this
                         cache-resident scene); sub-block `valu_issue`: issue cycles of the kernel's EXECUTED instruction mix against 1024 SIMDs x the clock
                         sampled inside the launches, as a range over the two rate tables.
  roofline_hbm_resident  the same kernel on the ~10 M-triangle stand-in of BASELINE config 4 (BVH 0.6 GB > 256 MB Infinity Cache), priced against the
                         measured roof of its access pattern (gather_roof_GBs: random dependent 80-byte gathers, tools/microbench/gather_roof.hip).
  primary_only           BASELINE config 2: primary rays only (adypt_trace_primary), >= 100 calls.
  tmp_lifetime_1         the K steps with every frame tracing its primary rays (report tmpLifetime 16 and 1).
  sun_visibility         the K steps with the optional occlusion query on: one more ray per escaped path, inside the same launch.
  single_frame           one adypt_trace_spp(ctx, 1) per call (what Instance::Update does), with the library's look-ahead; one_frame_per_pass: single frames
                         in a row (frames_in_flight 1) in one call, in synchronous calls, and in calls with one frame started ahead.
  cpu_baseline           oracle/liboracle.so on the host cores, bounded sample.

Scene: the real sponza.obj is not available anywhere (no network); a deterministic procedural stand-in of the same triangle count is
generated, written as OBJ/MTL + Adypt .config and loaded through the product's own loader -> SBVH -> CWBVH8 path (adypt_amd/scenes.py).
`$ADYPT_ASSETS/sponza.obj` is used instead when present.
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
PMC_BENCH, PMC_SANMIGUEL, PMC_PRIMARY, ISSUE_MODEL, GATHER_ROOF = "r6_pmc_bench.json", "r6_pmc_sanmiguel.json", "r6_pmc_primary.json", "r6_valu_issue_model.json", "r5_gather_roof.json"
PT_CFG = {"maxBounce": 8, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24}
SEED = 12345


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def counter_figures(name):
    """Per-ray counter figures of the dominant kernel from the committed rocprofv3 --pmc passes (tools/collect_profiles.sh ->
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
    """The vector-ALU issue roof of the dominant kernel — what binds it on the cache-resident bench scene.
      peak      = 1024 SIMDs x effective clock: shader cycles / 100 MHz ticks sampled inside THIS run's launches (adypt_get_shader_clock)
      achieved  = issue cycles the kernel's vector-ALU instructions need per second = SQ_INSTS_VALU per ray (committed PMC pass) x rays/s
                  (live) x the average ARCHITECTURAL issue time of its instruction mix (2 cycles full-rate fp32 / logic / moves, 4 the other
                  classes, packed and 64-bit, 8 transcendental: tools/valu_issue_model.py)
      frac      = achieved / peak: a fraction of a roof no instruction stream can exceed.
    `frac_at_single_class_loop_rates`: the same with the class times measured on one-instruction loops (profiles/r3_valu_calibration.json);
    it can exceed 1 — a mixed stream issues faster than the weighted sum of single-class loops — so those are not a roof."""
    profile_clock = pmc.get("effective_clock_GHz")
    clock = live_clock_ghz if live_clock_ghz and live_clock_ghz > 0.5 else profile_clock
    peak = N_SIMD * (clock or NOMINAL_CLOCK_GHZ)
    inst_rate = pmc["valu_insts_per_ray"] * kernel_rays_s
    arch = model["avg_issue_cycles_per_inst_architectural"] if model else 4.0
    achieved = inst_rate * arch / 1e9
    frac = achieved / peak
    out = {"bound": "valu_issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "Gcycle/s", "frac": round(frac, 4),
           "effective_clock_GHz": round(clock, 3) if clock else None,
           "clock_source": "live: s_memtime / s_memrealtime inside this run's launches" if clock == live_clock_ghz else "profile run",
           "effective_clock_GHz_in_profile_run": round(profile_clock, 3) if profile_clock else None, "lane_util": round(pmc["lane_util"], 4),
           "useful_frac": round(min(1.0, frac) * pmc["lane_util"], 4),
           "valu_insts_per_ray": round(pmc["valu_insts_per_ray"], 2), "valu_Ginst_s": round(inst_rate / 1e9, 1),
           "issue_cycles_per_inst_architectural": arch, "issue_cycles_available_per_inst": round(peak * 1e9 / inst_rate, 3),
           "pmc_stale": False, "pmc_source_hash": pmc["source_hash"]}
    if "lds_insts_per_ray" in pmc:
        # Round 6: an LDS or vector-memory instruction holds the SIMD's issue like a "normal" vector-ALU one (operands and results pass through the same register
        # file: 64 vector-ALU + 8 ds_bpermute_b32 cost what 72 vector-ALU do, tools/microbench/lds_xbar.hip) — so the roof the kernel runs into is the one over
        # ALL its vector instructions, and a trip that trades vector-ALU for LDS instructions (the triangle list of round 6) moves `frac` but not this figure
        other = (pmc["lds_insts_per_ray"] + pmc.get("vmem_rd_insts_per_ray", 0.0)) * kernel_rays_s * 4.0 / 1e9
        out["all_vector_instructions"] = {"valu_per_ray": round(pmc["valu_insts_per_ray"], 2), "lds_per_ray": round(pmc["lds_insts_per_ray"], 2), "vmem_rd_per_ray": round(pmc.get("vmem_rd_insts_per_ray", 0.0), 2),
                                          "achieved": round(achieved + other, 1), "frac": round((achieved + other) / peak, 4),
                                          "what": "issue cycles of the vector-ALU instructions (architectural rates) + 4 cycles per LDS and per vector-memory instruction, against 1024 SIMDs x the clock"}
    if model:
        loops = model["avg_issue_cycles_per_inst_single_class_loops"]
        out["issue_cycles_per_inst_single_class_loops"] = loops
        out["frac_at_single_class_loop_rates"] = round(inst_rate * loops / 1e9 / peak, 4)
        # the two rate tables bracket the truth: a mixed stream issues faster than the weighted sum of one-instruction loops, never faster than the architectural rates
        out["frac_range"] = [out["frac"], out["frac_at_single_class_loop_rates"]]
        out["mix_weighted_by"] = "executed instructions (block entries of the counting variant x static counts): profiles/%s" % ISSUE_MODEL
        if "model_over_measured" in model:
            out["mix_reproduces_SQ_INSTS_VALU_within"] = round(abs(model["model_over_measured"] - 1.0), 4)
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


def dominant(st):
    """Which kernel the roofline is about: k_path when the batches ran their bounces in one launch, else the traversal kernel."""
    if st.get("path_launches", 0) > 0:
        return {"kernel": "k_path<false, false>", "ms": st["path_ms"], "launches": st["path_launches"], "rays": st["path_rays"], "fused": True}
    return {"kernel": "k_trace<false, false>", "ms": st["trace_ms"], "launches": st["trace_launches"], "rays": st["rays"], "fused": False}


def census(pt, steps, warmup, expect_rays=None):
    """The same K frames again through the instrumented kernels -> exact node / triangle / hit / shaded counts -> algorithmic bytes of
    SURVEY.md 8(d): per ray 80 B x nodes visited + 48 B x triangles tested + 4 B x hit remap + 32 B ray read + 16 B hit write; full path
    tracing adds 164 B (triangle 100 + material 64) per shaded hit.  For the dominant kernel alone when it is k_path (its own counters)."""
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
    if cs.get("path_rays", 0):
        cs["k_rays"], cs["k_nodes"], cs["k_tris"], cs["k_hits"], cs["k_shaded"] = cs["path_rays"], cs["path_nodes"], cs["path_tris"], cs["path_hits"], cs["path_shaded"]
        cs["k_alg_bytes"] = 80 * cs["k_nodes"] + 48 * cs["k_tris"] + 4 * cs["k_hits"] + 48 * cs["k_rays"] + 164 * cs["k_shaded"]
    else:
        cs["k_rays"], cs["k_nodes"], cs["k_tris"], cs["k_hits"], cs["k_shaded"] = cs["rays"], cs["nodes_visited"], cs["tris_tested"], cs["hits"], 0
        cs["k_alg_bytes"] = cs["alg_bytes"]
    return cs


def gather_roof(resident_mb):
    """Fabric-side GB/s that k_path's own access pattern — every lane chasing dependent, uniformly random 80-byte records, k_path's launch shape
    (6 workgroups of 4 waves per CU) — sustains on a table of about the scene's resident size (profiles/r5_gather_roof.json, tools/microbench/gather_roof.hip)."""
    g = load_profile(GATHER_ROOF)
    if not g:
        return None
    rows = [r for r in g["configs"] if r["record_B"] == 80 and r["wgs_per_cu"] == 6 and r["chains"] == 1 and "fabric_B_per_record" in r]
    if not rows:
        return None
    r = min(rows, key=lambda x: abs(x["table_MiB"] - resident_mb))
    return {"gather_roof_GBs": round(r["Grecords_s"] * r["fabric_B_per_record"], 1), "gather_roof_useful_GBs": r["useful_GBs"], "gather_roof_table_MiB": r["table_MiB"],
            "gather_roof_fabric_B_per_80B_record": r["fabric_B_per_record"], "gather_roof_l2_hit_rate": r.get("l2_hit_rate"),
            "gather_roof_utcl1_miss_per_record": (r.get("translation_per_record") or {}).get("TCP_UTCL1_TRANSLATION_MISS_sum"),
            "gather_roof_fabric_read_latency_cycles": r.get("fabric_read_latency_cycles")}


def roofline_block(dom, cs, pmc_name, live_clock, bvh_mb, note_extra="", binds="valu_issue"):
    """The contract's roofline object for the dominant kernel (`dom`: name, HIP-event ms, launches, rays of the timed region; `cs`: census)."""
    ms, launches, rays = dom["ms"], max(1, dom["launches"]), dom["rays"]
    rays_s = rays / (ms * 1e-3) if ms > 0 else 0.0
    alg_gbs = cs["k_alg_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"kernel": dom["kernel"], "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_rays": int(rays), "kernel_rays_warmup": int(dom.get("rays_warmup", 0)),
           "launches": int(dom["launches"]), "avg_launch_ms": round(ms / launches, 4), "kernel_Mrays_s": round(rays_s / 1e6, 1),
           "alg_bytes_per_launch": round(cs["k_alg_bytes"] / launches), "alg_bytes_per_ray": round(cs["k_alg_bytes"] / max(1, cs["k_rays"]), 1),
           "nodes_per_ray": round(cs["k_nodes"] / max(1, cs["k_rays"]), 2), "tris_per_ray": round(cs["k_tris"] / max(1, cs["k_rays"]), 2),
           "shaded_per_ray": round(cs["k_shaded"] / max(1, cs["k_rays"]), 3),
           "alg_GBs": round(alg_gbs, 1), "alg_frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4), "shader_clock_GHz_live": round(live_clock, 3)}
    pmc, why = counter_figures(pmc_name)
    if pmc and pmc.get("kernel") != dom["kernel"]:
        pmc, why = None, "profiles/%s holds %s, this run's dominant kernel is %s" % (pmc_name, pmc.get("kernel"), dom["kernel"])
    if pmc:
        out.update(traffic_fields(pmc, rays, launches, rays_s))
        out.update({"achieved": out["traffic_GBs"], "frac": out["traffic_frac_of_hbm_peak"], "frac_fabric_traffic_of_hbm_peak": out["traffic_frac_of_hbm_peak"], "pmc_stale": False, "bound": binds,
                    "frac_of": "fabric traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE per ray, committed profile replayed against this run's live rays/s) / 8 TB/s HBM peak; "
                               "`bound` names what binds the kernel, which on this scene is NOT this resource: see valu_issue",
                    "traffic_over_algorithmic": round(pmc["traffic_bytes_per_ray"] / (cs["k_alg_bytes"] / max(1, cs["k_rays"])), 3),
                    "valu_issue": valu_roofline(pmc, load_profile(ISSUE_MODEL), rays_s, live_clock),
                    "note": "achieved / frac / traffic = fabric-side bytes of this kernel's launches (PMC FETCH_SIZE x 2 + WRITE_SIZE per ray, %s) x this run's "
                            "rays/s of the kernel (HIP events on the context's stream) against 8 TB/s: an upper bound on HBM bytes (Infinity-Cache hits are in it).  "
                            "alg_GBs / alg_frac_of_hbm_peak = SURVEY.md 8(d) bytes (exact census of the same frames) / the same time: NOT a fraction of HBM "
                            "traffic when it exceeds traffic — the BVH (%.0f MB) is served by the L2s (l2_hit_rate) and the Infinity Cache.  What binds the kernel "
                            "on this scene is vector-ALU issue: valu_issue (peak = 1024 SIMDs x the clock sampled inside this run's launches; achieved = "
                            "SQ_INSTS_VALU per ray x rays/s x the mix's architectural issue cycles, profiles/%s)%s" % (pmc.get("command", "?"), bvh_mb, ISSUE_MODEL, note_extra)})
    else:
        out.update({"achieved": round(alg_gbs, 1), "frac": round(alg_gbs / HBM_PEAK_GBS, 4), "traffic": None, "pmc_stale": True,
                    "note": "no counter figures: " + why + ".  achieved / frac fall back to SURVEY.md 8(d) algorithmic bytes / HIP-event time / 8 TB/s, which for an "
                            "L2 / Infinity-Cache resident BVH (%.0f MB) can exceed 1 and is not a fraction of HBM traffic" % bvh_mb})
    return out


def hbm_resident_block(args, dev):
    """roofline_hbm_resident: the dominant kernel where memory, not instruction issue, is what it waits for (BASELINE config 4 stand-in)."""
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
    warm = dominant(pt.GetStats())
    pt.ResetStats()
    t1 = time.perf_counter()
    pt.Trace(True, steps)
    wall = time.perf_counter() - t1
    st = pt.GetStats()
    live_clock = pt.GetShaderClockGHz()
    cs = census(pt, steps, warmup, st["rays"])
    bvh_mb = (len(inst.bvh.nodes) + len(inst.bvh.tri_indices) * 52) / 1e6
    out = roofline_block(dict(dominant(st), rays_warmup=warm["rays"]), cs, PMC_SANMIGUEL, live_clock, bvh_mb, binds="memory latency (between hbm and valu_issue)")
    # the roof of THIS access pattern: random dependent gathers of small records from a table of the scene's resident size (BVH + the per-reference triangle copy)
    resident_mb = bvh_mb + len(inst.bvh.tri_indices) * 128 / 1e6
    gr = gather_roof(resident_mb / 1.048576)
    if gr and out.get("traffic_GBs"):
        out.update(gr)
        out["frac_of_gather_roof"] = round(out["traffic_GBs"] / gr["gather_roof_GBs"], 4)
        out["resident_MB"] = round(resident_mb)
    out.update({"workload": "sanmiguel-like procedural stand-in (%s), %d triangles, BVH %.0f MB (nodes + Woop + index) > 256 MB Infinity Cache, 1920x1080, 8 bounces, %d frames after %d warm-up"
                            % (spec.label, inst.scene.n_tris, bvh_mb, steps, warmup),
                "whole_frame_Mrays_s": round(st["rays"] / wall / 1e6, 1), "trace_kernels_ms": round(st["trace_ms"], 2), "other_kernels_ms": round(st["shade_ms"], 2),
                "setup_s": round(setup_s, 1)})
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
    ap.add_argument("--repeats", type=int, default=7, help="the timed region (warm-up + K steps from the same start frame) is run this many times; value = median")
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tmp-lifetime", type=int, default=16, help="reference default 16: primary hits are re-traced every 16th frame")
    ap.add_argument("--comm", default=os.environ.get("ADYPT_BENCH_COMM", "native"), choices=["native", "torch"],
                    help="one process per GPU (launcher): 'native' = the library's own RCCL communicator (no torch); 'torch' = torch.distributed (nccl, or gloo with ADYPT_BENCH_BACKEND=gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-block", action="store_true", help="skip roofline_hbm_resident (the 10 M-triangle scene: ~25 s of setup)")
    ap.add_argument("--no-single-frame", action="store_true")
    ap.add_argument("--no-extra-blocks", action="store_true", help="skip primary_only and tmp_lifetime_1")
    ap.add_argument("--cache", default=os.environ.get("ADYPT_CACHE", os.path.join(ROOT, ".adypt_cache")))
    ap.add_argument("--selfcheck", action="store_true", help="(accepted for older command lines; the check below is the default for N > 1 and this flag changes nothing)")
    ap.add_argument("--no-selfcheck", action="store_true", help="N > 1: skip the self-check — before anything is timed, 2 frames are rendered on the N GPUs and on GPU 0 alone and the two images "
                    "compared bit for bit (exit 3 on a mismatch); it is also skipped, and says so in config.selfcheck, when GPU 0 has no room for a second context of the scene")
    ap.add_argument("--rehearsal", action="store_true", help="NOT a measurement: enables the library's test hooks (adypt_enable_test_hooks) so that ADYPT_MULTI_SHARED_DEVICE / "
                                                             "ADYPT_COMM_TRANSPORT=host can stand in for N GPUs on a box with one")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.repeats < 1:
        raise SystemExit("bench.py: --gpus, --steps, --repeats must be >= 1 and --warmup >= 0")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # no launcher and more than one GPU asked for: ONE process drives all of them (adypt_create_multi, ncclCommInitAll)
    multi = world == 1 and args.gpus > 1
    n_gpus = args.gpus
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
    if args.rehearsal:
        api.enable_test_hooks()  # without this call the library ignores the hook variables, whatever the environment says

    # ---- scene: every rank generates + builds its own copy (0.8 s; nothing to wait for, no shared cache to race on) ------------
    pt_cfg = dict(PT_CFG, tmpLifetime=args.tmp_lifetime)
    cache = args.cache if world == 1 else os.path.join(args.cache, "rank%d" % rank)
    t_setup = time.time()
    spec = scenes.make_scene(args.scene, cache, width=args.width, height=args.height, pt=pt_cfg)
    inst = api.Instance()
    shared_hook = multi and args.rehearsal and os.environ.get("ADYPT_MULTI_SHARED_DEVICE", "0") not in ("", "0")  # TEST HOOK: N tile shards on ONE device (not a measurement)
    host_hook = world > 1 and args.rehearsal and os.environ.get("ADYPT_COMM_TRANSPORT") == "host"                 # TEST HOOK: the RCCL call table served through shared memory
    try:
        if multi:
            ok = inst.InitializeFromFile(spec.config_path, shift_seed=SEED, devices=[0] * n_gpus if shared_hook else list(range(n_gpus)))
        else:
            ok = inst.InitializeFromFile(spec.config_path, shift_seed=SEED, device=dev, tile_rank=rank, tile_nranks=world)
    except N.AdyptError as e:
        raise SystemExit("bench.py --gpus %d: cannot create the tracer on %s (%s).  The product has no CPU path and bench.py never runs on fewer GPUs than asked for."
                         % (n_gpus, "devices 0..%d" % (n_gpus - 1) if multi else "device %d" % dev, e))
    assert ok, api.InstanceConfig.last_error()
    pt = inst.m_path_tracer
    c = inst.m_config.c
    fif = pt.GetFramesInFlight()
    t_setup = time.time() - t_setup
    comm_ranks = 0

    if multi and not shared_hook:
        try:
            pt.CommInit()  # ncclCommInitAll over the N devices, before anything is timed
        except (N.AdyptError, OSError) as e:
            raise SystemExit("bench.py --gpus %d: RCCL communicator over devices 0..%d failed (%s)" % (n_gpus, n_gpus - 1, e))
        comm_ranks = pt.CommRanks()
        if comm_ranks != n_gpus:
            raise SystemExit("bench.py --gpus %d: the RCCL communicator reports %d ranks" % (n_gpus, comm_ranks))
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
            comm_ranks = pt.CommRanks()
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
    elif multi:
        def barrier():
            pt.DeviceSynchronize()  # every device drained (hipDeviceSynchronize each); one process: nothing else to meet

        def gather():
            return pt.GatherDevice()  # grouped ncclSend / ncclRecv into device 0 + un-tiling there; returns after all streams have drained
    else:
        def barrier():
            # every rank's GPU drained (hipDeviceSynchronize), then all ranks met (RCCL all-reduce + drain)
            pt.DeviceSynchronize()
            if world > 1:
                pt.CommBarrier()

        def gather():
            return pt.CommGatherDevice()  # the one collective of the data path; rank 0: the assembled image, resident in HBM

    def reduce_max(x):
        if use_torch:
            t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return pt.CommAllReduce([x], "max")[0] if world > 1 else x

    # ---- --selfcheck: the assembled N-GPU image against the 1-GPU image of the same frames, before anything is timed ------------
    selfcheck = None
    if (multi or world > 1) and not args.no_selfcheck:  # (on by default: a scaling line must never describe a wrong image)
        pt.Reset()
        pt.Trace(True, 2)
        img_n = gather()
        if multi:
            img_n = pt.ReadResult()
        elif use_torch:
            img_n = img_n.cpu().numpy() if (img_n is not None and on_device) else img_n
        else:
            img_n = pt.CommReadResult()
        bad = 0.0
        if rank == 0:
            # the whole image on this rank's device alone: a SECOND context beside the N-GPU one — the scene again (with its per-reference copy of the
            # triangle records: 1.3 GB for the 10 M-triangle stand-in) and queues for two frames.  Not attempted without room for it.
            h = inst.m_hipscene
            # (nodes 80 B, references 52 B + their copy of the 128-byte triangle records, triangles 128 B; queues ~200 B per path of two frames; 1 GB of slack)
            need = len(h.bvh.nodes) * 80 + len(h.bvh.tri_indices) * (52 + 128) + len(h.scene.triangles) * 128 + c.width * c.height * 2 * 200 + (1 << 30)
            free = api.device_free_bytes(0 if multi else dev)
        if rank == 0 and free is not None and free < need:
            selfcheck = {"skipped": "device memory: %.1f GB free, a second context of this scene wants ~%.1f GB" % (free / 1e9, need / 1e9)}
            sys.stderr.write("bench.py: self-check skipped (%s)\n" % selfcheck["skipped"])
        elif rank == 0:
            solo = api.HipPathTracer()
            solo.Initialize(inst.m_config.pt_params(SEED), inst.m_hipscene, c.width, c.height, 0 if multi else dev, 0, 1)
            solo.SetFramesInFlight(2)
            ip, iv = inst.m_camera.matrices()
            solo.SetCamera(ip, iv, inst.m_camera.position)
            solo.Trace(True, 2)
            img_1 = solo.ReadResult()
            solo.destroy()
            differ = int((np.asarray(img_n).reshape(-1).view(np.uint32) != np.asarray(img_1).reshape(-1).view(np.uint32)).sum())
            selfcheck = {"frames": 2, "words_differing": differ, "image_mean": float(np.asarray(img_1).mean())}
            bad = 1.0 if differ else 0.0
            if differ:
                sys.stderr.write("bench.py --selfcheck: the image assembled from %d GPUs differs from the 1-GPU image in %d of %d words\n" % (n_gpus, differ, img_1.size))
        if world > 1:
            bad = reduce_max(bad)  # every rank leaves together
        if bad:
            raise SystemExit(3)
        pt.Reset()

    # ---- the timed region, R times from the same start frame: W untimed warm-up steps, then exactly K steps + the one gather ------------
    pt.SetInstrumentation(timing=True, counters=False)
    repeats = []
    image = None
    for rep in range(args.repeats):
        pt.Reset()
        if args.warmup:
            pt.Trace(True, args.warmup)
        gather()  # warms the communicator; the warm-up frames are done
        if rep == 0:
            warm_stats = pt.GetStats()
            rays_warmup = int(warm_stats["rays"])
        pt.ResetStats()
        barrier()
        t0 = time.perf_counter()
        pt.Trace(True, args.steps)
        t_gather = time.perf_counter()
        image = gather()  # rank 0 / device 0: the assembled W x H x 3 radiance, resident in HBM (as the reference's result texture is)
        barrier()
        elapsed = time.perf_counter() - t0
        gather_ms = (time.perf_counter() - t_gather) * 1e3
        st = pt.GetStats()
        repeats.append({"elapsed_all_ranks": reduce_max(elapsed), "elapsed": elapsed, "gather_ms": gather_ms, "st": st, "clock": pt.GetShaderClockGHz()})
    order = sorted(range(len(repeats)), key=lambda i: repeats[i]["elapsed_all_ranks"])
    med = repeats[order[len(order) // 2]]  # the median repeat: its kernel timings are the ones reported
    elapsed, gather_ms, st, live_clock = med["elapsed_all_ranks"], med["gather_ms"], med["st"], med["clock"]

    # per-device breakdown, so that a scaling run can be diagnosed from its own line
    mine = [med["elapsed"] * 1e3, float(st["trace_ms"]), float(st["shade_ms"]), gather_ms, float(st["rays"])]
    per_rank = None
    if multi:
        cstats = [pt.ContextStats(i) for i in range(n_gpus)]  # the counters / kernel timings of the last repeat, per device
        per_rank = np.array([[med["elapsed"] * 1e3, float(s["trace_ms"]), float(s["shade_ms"]), gather_ms, float(s["rays"])] for s in cstats])
        total_rays = int(st["rays"])
        image = pt.ReadResult()
    elif use_torch:
        if image is not None and on_device:
            image = image.cpu().numpy()
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        rays = torch.tensor([int(st["rays"])], dtype=torch.int64, device=red_dev)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        total_rays = int(rays.item())
        table = torch.zeros(5 * world, dtype=torch.float64, device=red_dev)
        table[5 * rank:5 * rank + 5] = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
        per_rank = table.cpu().numpy().reshape(world, 5)
    else:
        if world > 1:
            flat = [0.0] * (5 * world)
            flat[5 * rank:5 * rank + 5] = mine
            per_rank = np.array(pt.CommAllReduce(flat, "sum")).reshape(world, 5)
            total_rays = int(round(pt.CommAllReduce([float(st["rays"])], "sum")[0]))  # exact: < 2^53
        else:
            total_rays = int(st["rays"])
        image = pt.CommReadResult()  # untimed (collective): the image on the host for the checksum

    single_gpu = world == 1 and not multi
    extras = single_gpu and rank == 0
    roofline = primary_only = life1 = single = hbm = cpu = sun_vis = None
    if extras:
        # ---- census (untimed): the same K frames again through the instrumented kernels -> exact algorithmic bytes ------
        dom = dict(dominant(st), rays_warmup=dominant(warm_stats)["rays"])
        cs = census(pt, args.steps, args.warmup, st["rays"])
        bvh_mb = (len(inst.bvh.nodes) + len(inst.bvh.tri_indices) * 52) / 1e6
        on_bench_scene = (args.scene, args.width, args.height) == ("sponza", 1920, 1080)
        roofline = roofline_block(dom, cs, PMC_BENCH if on_bench_scene else "(none for this scene / size)", live_clock, bvh_mb)

        if not args.no_extra_blocks:
            # ---- BASELINE config 2: primary rays only (primaryray.glsl:46-94 -> adypt_trace_primary), one 2.07 M-ray launch per call ----
            n_calls = 128
            pt.SetInstrumentation(timing=True, counters=False)
            for _ in range(8):
                pt.Trace(False)
            pt.ResetStats()
            pt.DeviceSynchronize()
            t1 = time.perf_counter()
            for _ in range(n_calls):
                pt.Trace(False)
            pt.DeviceSynchronize()
            dt = time.perf_counter() - t1
            s2 = pt.GetStats()
            primary_only = {"workload": "BASELINE config 2: %dx%d, primary rays only, viewer type 0, %d calls of adypt_trace_primary (one traversal launch each)" % (c.width, c.height, n_calls),
                            "rays_per_call": int(s2["rays"] // n_calls), "Mrays_s_per_call": round(s2["rays"] / dt / 1e6, 1), "ms_per_call": round(dt * 1e3 / n_calls, 4),
                            "kernel_Mrays_s": round(s2["rays"] / s2["trace_ms"] / 1e3, 1), "kernel_ms_per_call": round(s2["trace_ms"] / n_calls, 4),
                            "kernel": "k_trace_camera<false, true>", "kernel_rays": int(s2["rays"]), "kernel_rays_warmup": int(s2["rays"] // n_calls) * 8}
            # its roofline: SURVEY.md 8(d) bytes from one instrumented call (exact node / triangle census of the same rays), counter figures from the committed
            # rocprofv3 --pmc passes of this very command (tools/collect_profiles.sh primary: every launch of this kernel name belongs to this block)
            pt.SetInstrumentation(timing=False, counters=True)
            pt.ResetStats()
            pt.Trace(False)
            s2c = pt.GetStats()
            pt.SetInstrumentation(timing=True, counters=False)
            k_rays_s = s2["rays"] / (s2["trace_ms"] * 1e-3)
            alg = (80.0 * s2c["nodes_visited"] + 48.0 * s2c["tris_tested"] + 4.0 * s2c["hits"]) / max(1, s2c["rays"]) + 32.0  # + the hit and the colour written per pixel (2 x float4)
            primary_only.update({"nodes_per_ray": round(s2c["nodes_visited"] / max(1, s2c["rays"]), 2), "tris_per_ray": round(s2c["tris_tested"] / max(1, s2c["rays"]), 2),
                                 "hits_per_ray": round(s2c["hits"] / max(1, s2c["rays"]), 3), "alg_bytes_per_ray": round(alg, 1),
                                 "alg_GBs": round(alg * k_rays_s / 1e9, 1), "alg_frac_of_hbm_peak": round(alg * k_rays_s / 1e9 / HBM_PEAK_GBS, 4)})
            pmc2, why2 = counter_figures(PMC_PRIMARY)
            if pmc2 and pmc2.get("kernel") == primary_only["kernel"]:
                clock = live_clock if live_clock and live_clock > 0.5 else NOMINAL_CLOCK_GHZ
                primary_only.update(traffic_fields(pmc2, s2["rays"], n_calls, k_rays_s))
                primary_only.update({"valu_insts_per_ray": round(pmc2["valu_insts_per_ray"], 2), "lane_util": round(pmc2["lane_util"], 4), "vmem_rd_insts_per_ray": round(pmc2.get("vmem_rd_insts_per_ray", 0.0), 3),
                                     "valu_issue_cycles_available_per_inst": round(N_SIMD * clock * 1e9 / (pmc2["valu_insts_per_ray"] * k_rays_s), 3),
                                     "traffic_over_algorithmic": round(pmc2["traffic_bytes_per_ray"] / alg, 3), "pmc_stale": False,
                                     "bound": "the end of the launch (5.3 rays per lane: the launch is ramping down from its first reservation on) on top of vector-ALU issue",
                                     "note": "coherent rays: fewer distinct lines per fetch than k_path's, the same trip; in-kernel rate against k_path's = what the ramp-down of a 2 M-ray launch costs"})
            else:
                primary_only.update({"pmc_stale": True, "note": "no counter figures: " + (why2 or "profiles/%s holds another kernel" % PMC_PRIMARY)})
            # ---- the same K steps with tmpLifetime 1: every frame traces its primary rays ----
            params = inst.m_config.pt_params(SEED)
            params.tmp_lifetime = 1
            pt.SetConfig(params)
            pt.Reset()
            if args.warmup:
                pt.Trace(True, args.warmup)
            pt.ResetStats()
            pt.DeviceSynchronize()
            t1 = time.perf_counter()
            pt.Trace(True, args.steps)
            pt.DeviceSynchronize()
            dt = time.perf_counter() - t1
            s3 = pt.GetStats()
            life1 = {"workload": "the timed region's %d steps after %d warm-up with tmpLifetime 1 (no primary hit is cached)" % (args.steps, args.warmup),
                     "Mrays_s": round(s3["rays"] / dt / 1e6, 1), "ms_per_step": round(dt * 1e3 / args.steps, 4), "rays_per_step": int(s3["rays"] // args.steps),
                     "trace_kernels_ms": round(s3["trace_ms"], 2), "other_kernels_ms": round(s3["shade_ms"], 2)}
            pt.SetConfig(inst.m_config.pt_params(SEED))
            pt.Reset()
            # ---- the same K steps with the sun-visibility query on (pathtracer.glsl:132, commented out in the reference): every escaped path
            # sends one more ray towards the sun, inside the same k_path launch (k_path<., SUN>); the queries are rays of the census ----
            pt.SetSunVisibility(True)
            if args.warmup:
                pt.Trace(True, args.warmup)
            pt.ResetStats()
            pt.DeviceSynchronize()
            t1 = time.perf_counter()
            pt.Trace(True, args.steps)
            pt.DeviceSynchronize()
            dt = time.perf_counter() - t1
            s4 = pt.GetStats()
            sun_vis = {"workload": "the timed region's %d steps after %d warm-up with adypt_set_sun_visibility on: one any-hit query per escaped path, traced inside the same launch" % (args.steps, args.warmup),
                       "Mrays_s": round(s4["rays"] / dt / 1e6, 1), "ms_per_step": round(dt * 1e3 / args.steps, 4), "rays_per_step": int(s4["rays"] // args.steps),
                       "one_launch_pipeline": bool(pt.GetFusedBounces()), "rate_vs_option_off": round((s4["rays"] / dt) / (total_rays / elapsed), 4)}
            pt.SetSunVisibility(False)
            pt.Reset()

        # ---- one frame per call (Instance::Update -> Trace(true), src/Instance.cpp:44-57) with the library's look-ahead ----------
        if not args.no_single_frame:
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
            pt.Trace(True, 16)       # one call, 16 frames, one frame per pass: consecutive frames overlap on two streams (no frame is traced that was not asked for)
            dt0 = time.perf_counter() - t1
            s0 = pt.GetStats()
            pt.ResetStats()
            t1 = time.perf_counter()
            for _ in range(32):
                pt.Trace(True, 1)    # one SYNCHRONOUS call per frame, nothing traced ahead: each frame's launch end is paid in full
            dt2 = time.perf_counter() - t1
            s2 = pt.GetStats()
            pt.SetLookahead(True)    # one call per frame, one frame started ahead on the second stream (dropped if the camera moves)
            pt.Trace(True, 2)
            pt.ResetStats()
            pt.DeviceSynchronize()
            t1 = time.perf_counter()
            for _ in range(32):
                pt.Trace(True, 1)
            pt.DeviceSynchronize()
            dt3 = time.perf_counter() - t1
            s3 = pt.GetStats()
            pt.SetLookahead(False)
            single = {"calls": n_calls, "ms_per_call": round(dt * 1e3 / n_calls, 4), "Mrays_s": round(s1["rays"] / dt / 1e6, 1),
                      "frac_of_batched": round((s1["rays"] / dt) / (total_rays / elapsed), 3),
                      "without_lookahead_Mrays_s": round(s0["rays"] / dt0 / 1e6, 1),
                      "one_frame_per_pass": {"one_call_of_16_frames_Mrays_s": round(s0["rays"] / dt0 / 1e6, 1),
                                             "one_synchronous_call_per_frame_Mrays_s": round(s2["rays"] / dt2 / 1e6, 1),
                                             "one_call_per_frame_one_frame_started_ahead_Mrays_s": round(s3["rays"] / dt3 / 1e6, 1),
                                             "note": "frames_in_flight 1.  Consecutive single frames run on two streams: frame k + 1's bounce 0 and k_path are enqueued under the end of frame k's "
                                                     "k_path — within a call that asks for several frames, and across calls with adypt_set_lookahead (one frame ahead, same tmpLifetime group only)"},
                      "note": "adypt_trace_spp(ctx, 1) per call, adypt_set_lookahead on: a call that needs untraced frames traces a whole pass of %d, the following calls only apply their running-mean step" % fif}

        # ---- the kernel where memory binds, and the CPU baseline ----------------------------------------------------------------------
        if not args.no_hbm_block and args.scene == "sponza":
            pt.destroy()  # frees the bench scene's queues first
            hbm = hbm_resident_block(args, dev)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(inst, c)

    if rank == 0:
        values = sorted(total_rays / r["elapsed_all_ranks"] / 1e6 for r in repeats)
        value = total_rays / elapsed / 1e6
        if multi:
            comm = ("TEST HOOK ADYPT_MULTI_SHARED_DEVICE: %d tile shards on ONE device, device copies instead of RCCL — not a measurement" % n_gpus) if shared_hook \
                else "native RCCL, ncclCommInitAll, %d ranks in one process" % comm_ranks
        elif world > 1:
            comm = "torch.distributed" if use_torch else ("host-staged TEST transport (ADYPT_COMM_TRANSPORT=host: not a measurement)" if host_hook
                                                         else "native RCCL, ncclCommInitRank, %d ranks, one process per GPU" % comm_ranks)
        else:
            comm = "none"
        out = {"metric": "Mrays/sec (primary+secondary) Sponza 1920x1080 8-bounce; % HBM-read roofline",
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "repeats": args.repeats, "value_min": round(values[0], 2), "value_max": round(values[-1], 2),
               "ms_per_step_all": [round(r["elapsed_all_ranks"] * 1e3 / args.steps, 4) for r in repeats],
               "config": {"workload": "%s-like procedural stand-in (%s), %d triangles, %dx%d, full wavefront path trace, maxBounce %d, tmpLifetime %d, 1 spp per step; one radiance gather per timed region"
                                      % (args.scene, spec.label, inst.scene.n_tris, c.width, c.height, c.max_bounce, c.tmp_lifetime),
                          "rays_per_step": round(total_rays / args.steps), "tile_shard": "32x32 blocks, owner (bx+by) mod N",
                          "frames_in_flight": fif, "rays_warmup": rays_warmup, "comm": comm, "comm_ranks": comm_ranks,
                          "devices": ("one process, devices %s" % ([0] * n_gpus if shared_hook else list(range(n_gpus)))) if multi else ("one process per device" if world > 1 else "device %d" % dev),
                          "bounces_in_one_launch": bool(st.get("path_launches", 0) > 0), "timing": "median of %d repeats of [reset, %d warm-up steps, barrier, %d timed steps + gather, barrier]" % (args.repeats, args.warmup, args.steps),
                          "setup_s": round(t_setup, 2),
                          "setup_s_per_device": [round(pt.SetupSeconds(i), 3) for i in range(n_gpus)] if multi else None,
                          "selfcheck": selfcheck, "rehearsal": bool(args.rehearsal)},
               "roofline": roofline, "roofline_hbm_resident": hbm, "cpu_baseline": cpu, "primary_only": primary_only, "tmp_lifetime_1": life1, "sun_visibility": sun_vis, "single_frame": single,
               "gather_ms": round(gather_ms, 3), "other_kernels_ms": round(st["shade_ms"], 2), "trace_kernels_ms": round(st["trace_ms"], 2),
               "per_rank": None if per_rank is None else {
                   "wall_ms": [round(float(v), 3) for v in per_rank[:, 0]], "trace_kernels_ms": [round(float(v), 3) for v in per_rank[:, 1]],
                   "other_kernels_ms": [round(float(v), 3) for v in per_rank[:, 2]], "gather_ms": [round(float(v), 3) for v in per_rank[:, 3]],
                   "rays": [int(v) for v in per_rank[:, 4]],
                   # share of a rank's wall time not inside a tracing kernel or the gather: launch tails show up in the kernels, host gaps here
                   "outside_kernels_frac": [round(float(1.0 - (r[1] + r[2] + r[3]) / max(r[0], 1e-9)), 3) for r in per_rank],
                   "note": "wall = the timed region of the median repeat (K steps + the gather); one process per device: each rank's own wall, rank 0's gather includes waiting "
                           "for the slowest peer; one process for all devices: the process's wall for every device, kernel times per device"},
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
