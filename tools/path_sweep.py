"""Developer tool (GPU box): time 20-frame batches of the bench workload for a list of "lib:ENV=.. ENV=.." settings, each in its own
process, interleaved over `rounds` rounds.   python tools/path_sweep.py [rounds] setting ...     (lib "-" = the product library)
SWEEP_SCENE / SWEEP_FRAMES / SWEEP_NRANKS (rank 0's shard of an N-way split) choose the workload."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from adypt_amd import api, scenes
scene = os.environ.get("SWEEP_SCENE", "sponza"); fr = int(os.environ.get("SWEEP_FRAMES", "20")); nr = int(os.environ.get("SWEEP_NRANKS", "1"))
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nr)
p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 5); p.DeviceSynchronize(); p.ResetStats()
reps = 5; walls = []
for _ in range(reps):
    t0 = time.perf_counter(); p.Trace(True, fr); walls.append(time.perf_counter() - t0)
s = p.GetStats(); walls.sort()
print(json.dumps({"setting": os.environ.get("SWEEP_SETTING"), "wall_Mrays_s_median": round(s["rays"] / reps / walls[reps // 2] / 1e6, 1), "ms_per_frame_median": round(walls[reps // 2] * 1e3 / fr, 4),
                  "ms_per_frame_min": round(walls[0] * 1e3 / fr, 4), "trace_ms_per_frame": round(s["trace_ms"] / fr / reps, 4), "other_ms_per_frame": round(s["shade_ms"] / fr / reps, 4),
                  "launches_per_batch": s["trace_launches"] // reps, "image_sum": float(p.ReadResult().sum())}))
''' % ROOT
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 1
for rnd in range(rounds):
    for setting in args:
        lib, _, envs = setting.partition(":")
        env = dict(os.environ)
        env["SWEEP_SETTING"] = setting
        if lib not in ("", "-"):
            env["ADYPT_LIB"] = os.path.join(ROOT, "adypt_amd", "libadypt_%s.so" % lib)
        env.update(dict(kv.split("=", 1) for kv in envs.split()) if envs else {})
        out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print(out.stdout.decode().strip() or ("FAILED %s: %s" % (setting, out.stderr.decode()[-400:])))
        sys.stdout.flush()
