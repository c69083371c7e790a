"""Developer tool: copy the bench lines and the shard / primary-rate measurements of a GPU run (gpurun_out/r6/final/, written by the command in
this file's last lines) into profiles/r6_bench_json_*.json and profiles/r6_primary_rate_and_shard_emulation_1gpu.txt, and print the numbers the
documents quote.    python tools/finalize_bench_profiles.py [dir]
GPU side (tools/full_cycle.sh does all of it):  O=gpurun_out/r6/final; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json; python bench.py > $O/default.json;
  ADYPT_FUSED_BOUNCES=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-hbm-block --no-cpu-baseline > $O/launch_per_bounce.json;
  python bench.py --scene salle --width 4096 --height 4096 --steps 8 --warmup 2 --no-hbm-block --no-cpu-baseline > $O/salle.json;
  for n in 1 2 4 8; do SWEEP_NRANKS=$n python tools/path_sweep.py 1 "-:" "-:ADYPT_FUSED_BOUNCES=0"; done > $O/shard.log;
  python tools/primary_rate.py 200 > $O/primary.log; python tools/shard_breakdown.py > $O/shard_breakdown.log"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = (sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6", "final")) + "/"
P = os.path.join(ROOT, "profiles") + "/"
lines = {}
for src, dst in (("driver_command", "driver_command"), ("default", "default"), ("launch_per_bounce", "launch_per_bounce"), ("salle", "salle_4096x4096"), ("no_deferral", "no_deferral")):
    if not os.path.exists(O + src + ".json"):
        continue  # (no_deferral: ADYPT_RARE_MIN=0, written by tools/full_cycle.sh only)
    line = open(O + src + ".json").read().strip().splitlines()[-1]
    lines[src] = json.loads(line)
    open(P + "r6_bench_json_%s.json" % dst, "w").write(line + "\n")
rows = {}
for l in open(O + "shard.log"):
    if l.startswith("{"):
        d = json.loads(l); rows.setdefault(d["setting"], []).append(d)
out = ["Round 6, one MI355X (same box, same gpurun call as profiles/r6_bench_json_*.json).  ms per frame of 20-frame batches (tools/path_sweep.py, median of 5), rank 0 of an N-way pixel-tile shard;",
       "'-:ADYPT_FUSED_BOUNCES=0' = the launch-per-bounce pipeline (round 3), '-:' = k_path.  eff = t(1) / (N x t(N)) within the same pipeline: predictions, not measurements."]
for i, n in enumerate((1, 2, 4, 8)):
    for s in ("-:ADYPT_FUSED_BOUNCES=0", "-:"):
        d, b = rows[s][i], rows[s][0]
        out.append("N=%d %-26s ms/frame %.4f  kernels %.4f + %.4f  eff %.3f" % (n, s, d["ms_per_frame_median"], d["trace_ms_per_frame"], d["other_ms_per_frame"], b["ms_per_frame_median"] / (n * d["ms_per_frame_median"])))
out += ["", "tools/shard_breakdown.py: EVERY rank of the split rendered on the one GPU, one after the other (wall of a 20-frame batch, median of 5; k_path / camera rays / k_shade_first and the rest):"]
sb = [json.loads(l) for l in open(O + "shard_breakdown.log") if l.startswith("{")]
out += [json.dumps(d) for d in sb]
w1 = [d for d in sb if d["n"] == 1][0]["wall_ms_batch"]
for n in (4, 8):
    mx = max(d["wall_ms_batch"] for d in sb if d["n"] == n)
    out.append("N=%d: slowest rank %.3f ms -> predicted efficiency %.3f (the ranks' rays differ by < 1.5 %%: the split is balanced; what is lost is the end of each rank's one k_path launch)" % (n, mx, w1 / (n * mx)))
out += ["", "tools/primary_rate.py 200 (BASELINE config 2, timing events on; one launch per call: k_trace_camera<false, true>):", [l for l in open(O + "primary.log") if l.startswith("{")][-1].strip()]
open(P + "r6_primary_rate_and_shard_emulation_1gpu.txt", "w").write("\n".join(out) + "\n")
for k, d in lines.items():
    r = d.get("roofline") or {}
    print(k, d["value"], d.get("value_min"), d.get("value_max"), d["ms_per_step"], "kernel", r.get("kernel"), r.get("kernel_Mrays_s"), r.get("avg_launch_ms"), "frac", r.get("frac"), "valu", (r.get("valu_issue") or {}).get("frac"))
    for b in ("primary_only", "tmp_lifetime_1", "single_frame", "roofline_hbm_resident", "cpu_baseline"):
        if d.get(b):
            print("   ", b, json.dumps({x: d[b][x] for x in d[b] if x not in ("note", "workload", "sample", "sample_1_thread")})[:300])
print("\n".join(out[2:10] + out[-6:-3]))
