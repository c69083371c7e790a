"""GPU box: split the dominant kernel's launches of a `rocprofv3 --kernel-trace` run of bench.py into the phases of the run and print the
average launch duration of the TIMED regions — the figure bench.py reports as roofline.avg_launch_ms (the --stats summary averages over
every launch of the process: warm-up, the R repeats, the tmpLifetime-1 block, the one-frame-per-call section, the 10 M-triangle scene).

bench.py launches the dominant kernel (roofline.kernel) in this order: R x [warm-up batches, timed batches], [tmpLifetime-1 block],
[single-frame section], [10 M-triangle scene: warm-up, timed].  With k_path a batch of m <= frames_in_flight frames is ONE launch; with the
launch-per-bounce pipeline it is (maxBounce - 1) traversal launches, + 1 when it contains a re-tracing frame.

usage: kernel_trace_phases.py <dir with *_kernel_trace.csv> <bench json line file>"""
import csv, glob, json, sys

root, bench_json = sys.argv[1], sys.argv[2]
bench = json.loads([l for l in open(bench_json).read().strip().splitlines() if l.startswith("{")][-1])
kernel = bench["roofline"]["kernel"]
fused = kernel.startswith("k_path")
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
dur = [(e - s) / 1e6 for s, e in rows]
fif = bench["config"]["frames_in_flight"]


def launches(first, n, life=16, bounces=8):
    """launches of the dominant kernel for n frames starting at frame `first`, traced in batches of at most fif frames"""
    total = 0
    while n > 0:
        m = min(n, fif)
        if fused:
            total += 1
        else:
            total += bounces - 1 + (1 if any((first + k) % life == 0 for k in range(m)) else 0)
        first += m
        n -= m
    return total


n_warm = launches(0, bench["warmup"]) if bench["warmup"] else 0
n_timed = launches(bench["warmup"], bench["steps"])
R = bench.get("repeats", 1)
timed, per_repeat, base = [], [], 0
for r in range(R):
    t = dur[base + n_warm:base + n_warm + n_timed]
    timed += t
    per_repeat.append(round(sum(t), 4))
    base += n_warm + n_timed
out = {"kernel": kernel, "launches_in_process": len(dur), "repeats": R, "warmup_launches_per_repeat": n_warm, "timed_launches_per_repeat": n_timed,
       # the launches a --pmc pass of the same command with --repeats 1 and the extra sections switched off sees: warm-up + timed
       "warmup_plus_timed_launches": n_warm + n_timed, "warmup_plus_timed_total_ms": sum(dur[:n_warm + n_timed]),
       "timed_avg_launch_ms": sum(timed) / max(1, len(timed)), "timed_ms_per_repeat": per_repeat, "timed_launch_ms": [round(x, 4) for x in timed],
       "bench_roofline_avg_launch_ms": bench["roofline"]["avg_launch_ms"], "bench_roofline_launches": bench["roofline"]["launches"]}
hb = bench.get("roofline_hbm_resident")
if hb and hb["kernel"] == kernel:
    tail = dur[-hb["launches"]:]
    out["hbm_resident_timed_avg_launch_ms"] = sum(tail) / max(1, len(tail))
    out["hbm_resident_bench_avg_launch_ms"] = hb["avg_launch_ms"]
print(json.dumps(out, indent=1))
