"""Developer tool (GPU box): the deferred ring of k_path's shading rounds (ADYPT_RARE_MIN) on scenes at the two ends — a closed room in which nearly every surface is
glossy or glass, and the same room all diffuse — 1920 x 1080, 8 bounces, batches of 16 frames; each setting in its own process.
    python tools/ab_rare_scenes.py ["ADYPT_RARE_MIN=0" "" "ADYPT_DEFER_MAX=32" ...]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from adypt_amd import api
obj, w, h = sys.argv[1], 1920, 1080
sc = api.Scene(); assert sc.LoadFromFile(obj)
cfg = api.InstanceConfig(); b = api.WideBVH(); b.Build(sc, cfg.bvh_params())
hs = api.HipScene(); hs.Initialize(sc, b)
p = cfg.pt_params(77); p.stack_size, p.max_bounce, p.subpixel, p.tmp_lifetime = 24, 8, 2, 16
pt = api.HipPathTracer(); pt.Initialize(p, hs, w, h)
ip, iv = api.camera_matrices(60.0, 200.0, -5.0, w, h); pt.SetCamera(ip, iv, [4.5, 2.0, 4.0])
pt.SetInstrumentation(timing=True); pt.Trace(True, 16); pt.ResetStats()
pt.Trace(True, 32); s = pt.GetStats()
print(json.dumps({"Mrays_s_in_kernels": round(s["rays"] / s["trace_ms"] / 1e3, 1), "rays_per_frame": int(s["rays"] / 32), "image_sum": float(pt.ReadResult().sum())}))
''' % ROOT
def room(path, wall, side):
    v, f = [], []
    def quad(a, b, c, d, mat):
        i = len(v) + 1; v.extend([a, b, c, d]); f.append("usemtl %s\nf %d %d %d\nf %d %d %d\n" % (mat, i, i + 1, i + 2, i, i + 2, i + 3))
    X, Y, Z = 6.0, 4.0, 5.0
    quad((0, 0, 0), (X, 0, 0), (X, 0, Z), (0, 0, Z), wall); quad((0, Y, 0), (0, Y, Z), (X, Y, Z), (X, Y, 0), wall)
    quad((0, 0, 0), (0, Y, 0), (X, Y, 0), (X, 0, 0), wall); quad((0, 0, Z), (X, 0, Z), (X, Y, Z), (0, Y, Z), wall)
    quad((0, 0, 0), (0, 0, Z), (0, Y, Z), (0, Y, 0), side); quad((X, 0, 0), (X, Y, 0), (X, Y, Z), (X, 0, Z), wall)
    quad((2, Y - 0.01, 2), (4, Y - 0.01, 2), (4, Y - 0.01, 3), (2, Y - 0.01, 3), "lamp"); quad((2.5, 0.0, 1.5), (3.5, 0.0, 1.5), (3.5, 1.2, 2.0), (2.5, 1.2, 2.0), side)
    open(path + ".mtl", "w").write("newmtl gl\nKd 0.5 0.4 0.3\nKs 0.4 0.4 0.4\nNs 120\nillum 2\nnewmtl glass\nKd 1 1 1\nNi 1.45\nillum 7\nnewmtl lamp\nKd 0 0 0\nKe 7 6 5\nillum 1\n"
                            "newmtl matte\nKd 0.6 0.55 0.5\nillum 1\n")
    open(path + ".obj", "w").write("mtllib %s.mtl\n" % os.path.basename(path) + "".join("v %g %g %g\n" % p for p in v) + "".join(f))
    return path + ".obj"
d = tempfile.mkdtemp()
for name, wall, side in (("room of glossy and glass", "gl", "glass"), ("the same room, matte", "matte", "matte"), ("matte room, one glossy wall", "matte", "gl")):
    obj = room(os.path.join(d, name.split()[0] + wall + side), wall, side)
    for setting in (sys.argv[1:] or ["ADYPT_RARE_MIN=0", ""]):
        env = dict(kv.split("=", 1) for kv in setting.split())
        r = subprocess.run([sys.executable, "-c", CHILD, obj], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=200)
        print(json.dumps({"scene": name, "env": setting, **json.loads(r.stdout.decode().strip().splitlines()[-1])})); sys.stdout.flush()
