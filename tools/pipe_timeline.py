"""Developer tool (GPU box): run under `rocprofv3 --kernel-trace` to see how the kernels of a (pipelined) batch lie in time.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/pipe_timeline.py <nranks> <frames> ;  python3 tools/pipe_timeline.py --analyse gpurun_out/tl
Renders rank 0's shard of an <nranks>-way split, <frames> frames per call, 4 calls (the last one is analysed)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--analyse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("adypt::", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    # the last k_resolve ends the last batch; the batch starts after the previous k_resolve
    res = [i for i, r in enumerate(rows) if r[2].startswith("k_resolve")]
    a, b = res[-2] + 1, res[-1] + 1
    batch = rows[a:b]
    t0, t1 = batch[0][0], max(r[1] for r in batch)
    print("last batch: %d kernels, %.3f ms from first start to last end" % (len(batch), (t1 - t0) / 1e6))
    ev = sorted([(s, 1) for s, e, *_ in batch] + [(e, -1) for s, e, *_ in batch])
    busy = over = 0; depth = 0; last = t0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    print("time with >= 1 kernel running: %.3f ms, with >= 2: %.3f ms, with none: %.3f ms" % (busy / 1e6, over / 1e6, (t1 - t0 - busy) / 1e6))
    by = {}
    for s, e, n, q, st in batch:
        by.setdefault(n[:24], []).append((e - s) / 1e3)
    for n, v in sorted(by.items()):
        print("  %-26s x%3d  sum %8.1f us  avg %7.1f us" % (n, len(v), sum(v), sum(v) / len(v)))
    print("first 40 kernels of the batch (start us, duration us, queue, name):")
    for s, e, n, q, st in batch[:40]:
        print("  %9.1f %8.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
    sys.exit(0)
from adypt_amd import api, scenes
nranks, frames = int(sys.argv[1]), int(sys.argv[2])
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nranks)
p = inst.m_path_tracer
import time
for k in range(4):
    t0 = time.perf_counter(); p.Trace(True, frames); dt = time.perf_counter() - t0
    print(json.dumps({"call": k, "pipeline": p.GetPipeline(), "ms_per_frame": round(dt * 1e3 / frames, 4)}))
