"""GPU box: per-ray PMC figures of the dominant kernel (bench.py: roofline.kernel — k_path<false> when a batch runs its bounces in one
launch) from the separate rocprofv3 --pmc passes written by tools/collect_profiles.sh — the file bench.py reads (profiles/r5_pmc_*.json).

HBM / fabric traffic, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is reported in KB and
counts the 128-byte fabric reads of 16-byte-per-lane loads as 64 bytes -> x 1024 x 2; WRITE_SIZE is exact -> x 1024.
Issue: SQ_INSTS_VALU (wave-instructions) per ray; lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).

Usage: pmc_profile.py <profile dir> <bench json of one of the passes> "<command>" [kernel_trace_phases.json of the same command]
Every pass runs the same command (with --repeats 1), so the kernel's launches (warm-up + timed frames) trace config.kernel_rays_warmup +
roofline.kernel_rays rays in each pass."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import device_source_hash

root, bench_json, command = sys.argv[1], sys.argv[2], sys.argv[3]
phases = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None


def per_kernel(counter):
    vals = []  # (KERNEL is set below, before the first call)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


bench = json.loads([l for l in open(bench_json).read().strip().splitlines() if l.startswith("{")][-1])
block = bench.get(os.environ.get("PMC_BLOCK", "roofline")) or bench["roofline"]
KERNEL = block["kernel"]
rays = block["kernel_rays"] + block["kernel_rays_warmup"]
fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
hit, req = per_kernel("TCC_HIT_sum"), per_kernel("TCC_REQ_sum")
insts, tcyc = per_kernel("SQ_INSTS_VALU"), per_kernel("SQ_THREAD_CYCLES_VALU")
vmem = per_kernel("SQ_INSTS_VMEM_RD")
lds, salu = per_kernel("SQ_INSTS_LDS"), per_kernel("SQ_INSTS_SALU")
busy, wcyc, gui = per_kernel("SQ_BUSY_CYCLES"), per_kernel("SQ_WAVE_CYCLES"), per_kernel("GRBM_GUI_ACTIVE")
active_valu = per_kernel("SQ_ACTIVE_INST_VALU")
N_XCD, N_SIMD = 8, 1024
traffic = sum(fetch) * 1024 * 2 + sum(write) * 1024
out = {
    "kernel": KERNEL,
    "command": command + " (one rocprofv3 --pmc pass per counter group, tools/collect_profiles.sh)",
    "launches": len(fetch) or len(insts),
    "rays": rays,
    "FETCH_SIZE_KB_per_launch": sum(fetch) / max(1, len(fetch)),
    "WRITE_SIZE_KB_per_launch": sum(write) / max(1, len(write)),
    "correction": "gfx950: FETCH_SIZE counts 128-B fabric reads as 64 B for 16-B-per-lane loads -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact; unit KB -> x1024",
    "traffic_bytes_per_launch": traffic / max(1, len(fetch)),
    "traffic_bytes_per_ray": traffic / max(1, rays),
    "alg_bytes_per_ray": block["alg_bytes_per_ray"],
    "TCC_hit_rate": sum(hit) / max(1.0, sum(req)),
    "SQ_INSTS_VALU_per_launch": sum(insts) / max(1, len(insts)),
    "valu_insts_per_ray": sum(insts) / max(1, rays),
    "lane_util": sum(tcyc) / max(1.0, 64.0 * sum(insts)),
    "vmem_rd_insts_per_ray": sum(vmem) / max(1, rays),
    "lds_insts_per_ray": sum(lds) / max(1, rays), "salu_insts_per_ray": sum(salu) / max(1, rays),  # (an LDS instruction costs the SIMD an issue slot like a vector-ALU one: profiles/r6_ablations.txt 3)
    "GRBM_GUI_ACTIVE_per_launch": sum(gui) / max(1, len(gui)),
    "traffic_bytes_per_ray_uncorrected": (sum(fetch) * 1024 + sum(write) * 1024) / max(1, rays),
    "SQ_ACTIVE_INST_VALU_per_launch": sum(active_valu) / max(1, len(active_valu)),
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the launch's duration in shader-clock cycles
    "cycles_per_launch": sum(gui) / max(1, len(gui)) / N_XCD,
    # NOT a busy fraction: profiles/r3_valu_calibration.json shows SQ_ACTIVE_INST_VALU = 1 per instruction (2 per transcendental) whatever
    # its issue time, so x 4 over-counts full-rate instructions (kept for comparison with rounds 1-2, unused by bench.py)
    "valu_busy_frac": (sum(active_valu) * 4.0) / max(1.0, N_SIMD * sum(gui) / N_XCD),
    "cycles_per_valu_inst_per_simd": (N_SIMD * sum(gui) / N_XCD) / max(1.0, sum(insts)),
    "source_hash": device_source_hash(),
    "note": "fabric-side bytes (Infinity-Cache hits are counted, MI355X_MICROARCH.md)",
}
# translation and fabric-side latency (VERDICT r4 task 1): per vector-memory read instruction / per fabric read
utcl_req, utcl_miss, utcl_mum = per_kernel("TCP_UTCL1_REQUEST_sum"), per_kernel("TCP_UTCL1_TRANSLATION_MISS_sum"), per_kernel("TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum")
ea_level, ea_req, ea_dram = per_kernel("TCC_EA0_RDREQ_LEVEL_sum"), per_kernel("TCC_EA0_RDREQ_sum"), per_kernel("TCC_EA0_RDREQ_DRAM_sum")
l1_lat, l1_req = per_kernel("TCP_TCC_READ_REQ_LATENCY_sum"), per_kernel("TCP_TCC_READ_REQ_sum")
if utcl_req:
    out["utcl1_requests_per_ray"] = sum(utcl_req) / max(1, rays)
    out["utcl1_translation_miss_per_request"] = sum(utcl_miss) / max(1.0, sum(utcl_req))
    out["utcl1_translation_miss_under_miss_per_request"] = sum(utcl_mum) / max(1.0, sum(utcl_req))
if ea_level and ea_req:
    out["fabric_read_latency_cycles"] = sum(ea_level) / max(1.0, sum(ea_req))  # requests in flight summed per cycle / requests
    out["fabric_reads_from_dram_frac"] = sum(ea_dram) / max(1.0, sum(ea_req)) if ea_dram else None
if l1_lat and l1_req:
    out["l1_to_l2_read_latency_cycles"] = sum(l1_lat) / max(1.0, sum(l1_req))
if phases:
    if "timed_avg_launch_ms" in phases:
        out["avg_launch_ms_kernel_trace"] = phases["timed_avg_launch_ms"]
    if phases.get("warmup_plus_timed_launches") == len(gui) and phases.get("warmup_plus_timed_total_ms"):
        # the same launches in both runs (warm-up + timed frames): cycles of all of them / duration of all of them
        out["kernel_trace_total_ms_same_launches"] = phases["warmup_plus_timed_total_ms"]
        out["effective_clock_GHz"] = (sum(gui) / N_XCD) / (phases["warmup_plus_timed_total_ms"] * 1e-3) / 1e9
print(json.dumps(out, indent=1))
