"""Developer tool (GPU box): the octant-affinity question of tools/xcd_locality_probe.py asked with REALISTIC ray sets instead of uniformly
random ones: camera rays in the product's queue order (block-major local pixels), then bounce 1..4 rays leaving the hit points in a random
direction of the outer hemisphere, the survivors KEPT IN QUEUE ORDER — what k_shade's append produces — against the same rays ordered by
the octant / Morton code of their origin.  adypt_trace_rays puts consecutive eighths of a batch into the 8 XCD-affine segments.
    python tools/xcd_locality_probe2.py [scene]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "sanmiguel"
W, H = 1920, 1080
spec = scenes.make_scene(scene, os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=W, height=H,
                         pt={"maxBounce": 8, "subpixel": 8, "tmpLifetime": 16, "clamp": 4.0, "sun": [12.0, 11.0, 10.0], "stackSize": 24})
inst = api.Instance()
assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
p = inst.m_path_tracer
p.SetInstrumentation(timing=True)
tris = np.frombuffer(np.ascontiguousarray(inst.scene.triangles).tobytes(), dtype=np.float32).reshape(-1, 25)
lo, hi = inst.scene.GetAABB()
rs = np.random.RandomState(11)

# camera rays in queue order: 32x32 blocks row-major, inside a block 16 wave tiles of 8x8
ip, iv = inst.m_camera.matrices()
ip = np.asarray(ip, np.float64).reshape(4, 4).T; iv = np.asarray(iv, np.float64).reshape(4, 4).T   # column-major in the API
L = np.arange(((W + 31) // 32) * ((H + 31) // 32) * 1024)
blk, inb = L >> 10, L & 1023
wt, ln = inb >> 6, inb & 63
x = (blk % ((W + 31) // 32)) * 32 + (wt & 3) * 8 + (ln & 7); y = (blk // ((W + 31) // 32)) * 32 + (wt >> 2) * 8 + (ln >> 3)
ok = (x < W) & (y < H); x, y = x[ok], y[ok]
sx = 2.0 * (x + 0.5) / W - 1.0; sy = -(2.0 * (y + 0.5) / H - 1.0)
t4 = (ip @ np.stack([sx, sy, np.ones_like(sx), np.ones_like(sx)]))[:3]
d = (iv[:3, :3] @ t4).T; d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(np.asarray(inst.m_camera.position, np.float64)[None, :], len(d), 0)


def time_batch(o, d, order):
    rays = np.zeros((len(o), 8), np.float32)
    rays[:, 0:3] = o[order]; rays[:, 3] = 1e-4; rays[:, 4:7] = d[order]
    p.TraceRays(rays[: 1 << 18], with_stats=False)
    best, h = 1e9, None
    for _ in range(3):
        p.ResetStats(); h = p.TraceRays(rays, with_stats=False); best = min(best, p.GetStats()["trace_ms"])
    back = np.empty_like(h); back[order] = h
    return best, back


def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
    return v


for bounce in range(0, 5):
    n = len(o)
    u = (o - lo) / np.maximum(hi - lo, 1e-9)
    octant = (u[:, 0] > 0.5).astype(np.int64) | ((u[:, 1] > 0.5).astype(np.int64) << 1) | ((u[:, 2] > 0.5).astype(np.int64) << 2)
    q = np.clip((u * 1024).astype(np.int64), 0, 1023)
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    orders = {"queue order": np.arange(n)}
    if bounce > 0:
        orders["by octant of origin (stable)"] = np.argsort(octant, kind="stable")
        orders["by morton code of origin"] = np.argsort(morton, kind="stable")
        orders["random"] = rs.permutation(n)
    res, hits = {}, None
    for name, order in orders.items():
        ms, h = time_batch(o, d, order)
        if hits is None: hits = h
        res[name] = round(n / ms / 1e3, 1)
    print(json.dumps({"scene": scene, "bounce": bounce, "rays": n, "Mrays_s": res, "octant_histogram": np.bincount(octant, minlength=8).tolist()}))
    sys.stdout.flush()
    # next bounce: survivors in queue order
    hit = hits["tri_id"] >= 0
    tri = tris[hits["tri_id"][hit]]
    o2 = o[hit] + d[hit] * hits["t"][hit][:, None].astype(np.float64)
    ng = np.cross(tri[:, 3:6] - tri[:, 0:3], tri[:, 6:9] - tri[:, 0:3]).astype(np.float64)
    ng /= np.maximum(np.linalg.norm(ng, axis=1, keepdims=True), 1e-30)
    ng[np.sum(ng * d[hit], axis=1) > 0] *= -1.0
    nd = rs.randn(len(o2), 3); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd[np.sum(nd * ng, axis=1) < 0] *= -1.0
    o, d = o2 + ng * 1e-3, nd
