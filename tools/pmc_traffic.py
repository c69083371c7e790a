"""GPU box: HBM/fabric traffic of the traversal kernel from the separate rocprofv3 --pmc passes written by
tools/collect_profiles.sh.  Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is
reported in KB and counts the 128-byte fabric reads of 16-byte-per-lane loads as 64 bytes -> x 1024 x 2; WRITE_SIZE is
exact -> x 1024.  Usage: pmc_traffic.py <profile dir> <bench json of the kernel-trace pass>"""
import csv, glob, json, sys

root, bench_json = sys.argv[1], sys.argv[2]
KERNEL = "k_trace<false, false>"


def per_kernel(counter):
    vals = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
hit, req = per_kernel("TCC_HIT_sum"), per_kernel("TCC_REQ_sum")
rays = bench["config"]["rays_per_step"] * bench["steps"]  # --warmup 0: every k_trace<false,false> launch is a timed one
traffic = sum(fetch) * 1024 * 2 + sum(write) * 1024
out = {
    "kernel": KERNEL,
    "command": "rocprofv3 --pmc <group> -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline (tools/collect_profiles.sh; one run per counter group)" % (bench["steps"], bench["warmup"]),
    "launches": len(fetch),
    "rays": rays,
    "FETCH_SIZE_KB_per_launch": sum(fetch) / max(1, len(fetch)),
    "WRITE_SIZE_KB_per_launch": sum(write) / max(1, len(write)),
    "correction": "gfx950: FETCH_SIZE counts 128-B fabric reads as 64 B for 16-B-per-lane loads -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact; unit KB -> x1024",
    "traffic_bytes_per_launch": traffic / max(1, len(fetch)),
    "traffic_bytes_per_ray": traffic / max(1, rays),
    "alg_bytes_per_ray": bench["roofline"]["alg_bytes_per_ray"],
    "TCC_hit_rate": sum(hit) / max(1.0, sum(req)),
    "note": "fabric-side bytes (Infinity-Cache hits are counted, MI355X_MICROARCH.md); the BVH is cache resident, so this is far below the algorithmic bytes: no wasted re-reads",
}
print(json.dumps(out, indent=1))
