"""Developer tool (GPU box): random scheduling tunables of k_path (deferred-ring threshold, shading-batch size, refill threshold, workgroups per CU, LDS stack depth,
frames in flight) on small scenes — every run in its own process under a timeout (a deadlock shows as a timeout, not as a hung box), the image, the primary-hit
cache and the counters compared bit for bit with the launch-per-bounce pipeline.    python tools/stress_path_tunables.py [n_cases] [seed]      (STRESS_BIG=1: larger frames of the bench scene)"""
import json, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
from tests.test_gpu_fused_bounces import _render, _same
case = json.loads(sys.argv[1])
cache = os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache")
ref = _render(cache, case["scene"], case["w"], case["h"], case["pt"], case["spp"], fused=False, env={k: v for k, v in case["env"].items() if k == "ADYPT_FRAMES_IN_FLIGHT"}, sun=case.get("sun"))
one = _render(cache, case["scene"], case["w"], case["h"], case["pt"], case["spp"], fused=True, env=case["env"], sun=case.get("sun"))
_same(ref, one, json.dumps(case))
print("ok", one["stats"]["path_rays"])
''' % ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    env = {"ADYPT_REF_TRIANGLES_MAX_MB": rnd.choice(["", "", "0"]), "ADYPT_RARE_MIN": rnd.choice([0, 1, 2, 5, 16, 33, 48, 64]), "ADYPT_SHADE_MIN": rnd.choice([1, 3, 9, 24, 47, 64]), "ADYPT_REFILL_MIN": rnd.choice([1, 4, 7, 16, 31, 64]),
           "ADYPT_PATH_BLOCKS_PER_CU": rnd.choice([1, 1, 2, 4, 6]), "ADYPT_DEFER_MAX": rnd.choice([0, 1, 8, 24, 64])}
    if rnd.random() < 0.3: env["ADYPT_PATH_LDS_DEPTH"] = rnd.choice([1, 2, 3])
    if rnd.random() < 0.3: env["ADYPT_FRAMES_IN_FLIGHT"] = rnd.choice([2, 3, 5])
    big = os.environ.get("STRESS_BIG", "0") != "0"   # the bench scene at up to 960 x 540: workgroups that hold full tables, rounds that defer
    scene = rnd.choice(["sponza", "sponza", "sibenik"] if big else ["tiny0", "tiny0", "sibenik"])
    w, h = rnd.choice([(480, 270), (640, 360), (960, 540)] if big else [(64, 40), (120, 68), (200, 120), (320, 200)])
    case = {"scene": scene, "w": w, "h": h, "spp": rnd.choice([5, 8, 12]), "env": env, "sun": rnd.choice([None, None, [0.6, 1.0, 0.2], [-0.3, 0.8, 0.5]]),
            "pt": {"tmpLifetime": rnd.choice([1, 3, 4, 16]), "maxBounce": rnd.choice([2, 5, 8, 13]), "subpixel": rnd.choice([1, 2, 3]), "stackSize": 24}}
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, json.dumps(case)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, cwd=ROOT)
        ok = r.returncode == 0
        msg = r.stdout.decode().strip().splitlines()[-1:] if ok else r.stderr.decode().strip().splitlines()[-3:]
    except subprocess.TimeoutExpired:
        ok, msg = False, ["TIMEOUT (deadlock?)"]
    bad += 0 if ok else 1
    print(json.dumps({"case": i, "ok": ok, "msg": msg, **case})); sys.stdout.flush()
    if not ok and "TIMEOUT" in msg[0]:
        break  # (nothing further on a box whose GPU may be busy with a hung kernel)
print(json.dumps({"cases": i + 1, "failed": bad}))
sys.exit(1 if bad else 0)
