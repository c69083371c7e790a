"""Developer tool (GPU box): what workgroups 0..7 of one k_path launch do after the global queue ran dry — every shading round with its time, size and
the state of the workgroup's lists.  Needs   tools/build_variant.sh drain --transform adypt_amd/csrc/measure/k_path_drain_trace.py   (ADYPT_LIB=...)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes, _native as N
nr = int(os.environ.get("SWEEP_NRANKS", "1")); fr = int(os.environ.get("SWEEP_FRAMES", "20"))
spec = scenes.make_scene("sponza", os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache"), width=1920, height=1080,
                         pt={"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24, "subpixel": 8, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]})
inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345, tile_rank=0, tile_nranks=nr)
p = inst.m_path_tracer; p.Trace(True, 5); p.DeviceSynchronize()
lib = C.CDLL(N.LIB_PATH)
W, E = 8, 8192
class Ev(C.Structure):
    _fields_ = [("t0", C.c_uint64), ("t1", C.c_uint64), ("take", C.c_uint32), ("live", C.c_uint32), ("left_shade", C.c_uint32), ("left_trace", C.c_uint32), ("own_rays", C.c_uint32), ("pad", C.c_uint32)]
ev = ((Ev * E) * W)(); cnt = (C.c_uint32 * W)(); dry = C.c_uint64()
lib.adypt_debug_read_drain(None, None, None, 1)
p.Trace(True, fr); p.DeviceSynchronize()
assert lib.adypt_debug_read_drain(ev, cnt, C.byref(dry), 0) == 0
out = []
for w in range(W):
    rows = [ev[w][i] for i in range(min(cnt[w], E))]
    after = [r for r in rows if r.t0 >= dry.value]
    before = [r for r in rows if r.t0 < dry.value]
    def stat(rs):
        if not rs: return None
        dur = [(r.t1 - r.t0) / 100.0 for r in rs]
        gap = [(rs[i + 1].t0 - rs[i].t1) / 100.0 for i in range(len(rs) - 1)]
        return {"rounds": len(rs), "avg_take": round(sum(r.take for r in rs) / len(rs), 1), "avg_round_us": round(sum(dur) / len(dur), 1),
                "avg_gap_between_rounds_us": round(sum(gap) / max(1, len(gap)), 1), "avg_own_rays_of_shader": round(sum(r.own_rays for r in rs) / len(rs), 1)}
    last = (rows[-1].t1 - dry.value) / 100.0 if rows else None
    tl = [(round((r.t0 - dry.value) / 100.0), r.take, r.live, r.left_shade, r.left_trace, r.own_rays, round((r.t1 - r.t0) / 100.0, 1)) for r in after]
    out.append({"wg": w, "events": cnt[w], "steady_state_before_dry(last 200)": stat(before[-200:]), "after_dry": stat(after), "last_round_ends_us_after_dry": round(last, 1) if last else None,
                "after_dry_timeline(us_after_dry, take, live, left_shade, left_trace, own_rays, round_us)": tl[::max(1, len(tl) // 40)]})
print(json.dumps({"nranks": nr, "workgroups": out}))
