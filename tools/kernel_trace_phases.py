"""GPU box: split the traversal launches of a `rocprofv3 --kernel-trace` run of bench.py into the phases of the run and print
the average launch duration of each — in particular of the TIMED region, the figure bench.py reports as roofline.avg_launch_ms
(the --stats summary averages over every launch of the process: warm-up, timed frames, the one-frame-per-call section and the
10 M-triangle scene).

bench.py launches k_trace<false, false> in this order: warm-up batch, timed batch(es), [single-frame section], [10 M-triangle
scene: warm-up, timed].  A batch of m frames starting at frame s issues (1 if it contains a re-tracing frame) + (maxBounce - 1)
launches, + 1 more when its first frame re-traces; the split below only needs the counts.

usage: kernel_trace_phases.py <dir with *_kernel_trace.csv> <bench json line file>"""
import csv, glob, json, sys

root, bench_json = sys.argv[1], sys.argv[2]
bench = json.loads([l for l in open(bench_json).read().strip().splitlines() if l.startswith("{")][-1])
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<false, false>" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
dur = [(e - s) / 1e6 for s, e in rows]


def launches(first, m, life=16, bounces=8):
    """traversal launches of one batch of m frames starting at frame `first`"""
    retrace = any((first + k) % life == 0 for k in range(m))
    return bounces - 1 + (1 if retrace else 0)  # bounce 0 starts from cached primary hits unless the batch re-traces them (one primary-only launch)


n_warm = launches(0, bench["warmup"]) if bench["warmup"] else 0
n_timed, first, left, fif = 0, bench["warmup"], bench["steps"], bench["config"]["frames_in_flight"]
while left > 0:
    m = min(left, fif)
    n_timed += launches(first, m)
    first += m
    left -= m
timed = dur[n_warm:n_warm + n_timed]
out = {"k_trace<false,false>_launches_in_process": len(dur), "warmup_launches": n_warm, "timed_launches": len(timed),
       # the launches a --pmc pass of the same command sees when the extra sections are switched off: warm-up + timed
       "warmup_plus_timed_launches": n_warm + n_timed, "warmup_plus_timed_total_ms": sum(dur[:n_warm + n_timed]),
       "timed_avg_launch_ms": sum(timed) / max(1, len(timed)), "timed_launch_ms": [round(x, 4) for x in timed],
       "bench_roofline_avg_launch_ms": bench["roofline"]["avg_launch_ms"], "bench_roofline_launches": bench["roofline"]["launches"]}
hb = bench.get("roofline_hbm_resident")
if hb:
    tail = dur[-hb["launches"]:]
    out["hbm_resident_timed_avg_launch_ms"] = sum(tail) / max(1, len(tail))
    out["hbm_resident_bench_avg_launch_ms"] = hb["avg_launch_ms"]
print(json.dumps(out, indent=1))
