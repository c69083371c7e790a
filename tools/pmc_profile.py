"""GPU box: per-ray PMC figures of the traversal kernel from the separate rocprofv3 --pmc passes written by
tools/collect_profiles.sh — the file bench.py reads (profiles/r2_pmc_*.json).

HBM / fabric traffic, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is reported in KB and
counts the 128-byte fabric reads of 16-byte-per-lane loads as 64 bytes -> x 1024 x 2; WRITE_SIZE is exact -> x 1024.
Issue: SQ_INSTS_VALU (wave-instructions) per ray; lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).

Usage: pmc_profile.py <profile dir> <bench json of one of the passes> "<command>"
Every pass runs the same command, so the k_trace<false, false> launches (warm-up + timed frames) trace config.rays_warmup +
rays_per_step x steps rays in each pass."""
import csv, glob, json, sys

root, bench_json, command = sys.argv[1], sys.argv[2], sys.argv[3]
KERNEL = "k_trace<false, false>"


def per_kernel(counter):
    vals = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


bench = json.loads([l for l in open(bench_json).read().strip().splitlines() if l.startswith("{")][-1])
rays = bench["config"]["rays_per_step"] * bench["steps"] + bench["config"]["rays_warmup"]
fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
hit, req = per_kernel("TCC_HIT_sum"), per_kernel("TCC_REQ_sum")
insts, tcyc = per_kernel("SQ_INSTS_VALU"), per_kernel("SQ_THREAD_CYCLES_VALU")
vmem = per_kernel("SQ_INSTS_VMEM_RD")
busy, wcyc, gui = per_kernel("SQ_BUSY_CYCLES"), per_kernel("SQ_WAVE_CYCLES"), per_kernel("GRBM_GUI_ACTIVE")
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
    "alg_bytes_per_ray": bench["roofline"]["alg_bytes_per_ray"],
    "TCC_hit_rate": sum(hit) / max(1.0, sum(req)),
    "SQ_INSTS_VALU_per_launch": sum(insts) / max(1, len(insts)),
    "valu_insts_per_ray": sum(insts) / max(1, rays),
    "lane_util": sum(tcyc) / max(1.0, 64.0 * sum(insts)),
    "vmem_rd_insts_per_ray": sum(vmem) / max(1, rays),
    "GRBM_GUI_ACTIVE_per_launch": sum(gui) / max(1, len(gui)),
    "note": "fabric-side bytes (Infinity-Cache hits are counted, MI355X_MICROARCH.md)",
}
print(json.dumps(out, indent=1))
