"""Developer tool (GPU box): what the shading round's material divergence costs k_path — the bench scene as it is against the same geometry with every
glossy / dielectric material made a plain diffuse one (an UPPER bound on what class-coherent shading rounds could buy: the rays differ, it is not an A/B of code)."""
import json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adypt_amd import api, scenes
cache = os.environ.get("ADYPT_CACHE", "/tmp/adypt_cache")
pt = {"maxBounce": 8, "tmpLifetime": 16, "stackSize": 24}
for mode in ("as is", "no glossy / dielectric", "no glossy / dielectric / texture"):
    out = os.path.join(cache, "mat_" + re.sub(r"\W+", "_", mode))
    spec = scenes.make_scene("sponza", out, width=1920, height=1080, pt=pt)
    mtl = spec.obj_path[:-4] + ".mtl"
    t = open(mtl).read()
    if mode != "as is":
        t = re.sub(r"illum [267]\b", "illum 1", t)
        if "texture" in mode: t = re.sub(r"map_Kd .*\n", "", t)
        open(mtl, "w").write(t)
        if os.path.exists(spec.config["bvh_file"]): pass  # (geometry unchanged: the cached BVH stays valid)
    inst = api.Instance(); assert inst.InitializeFromFile(spec.config_path, shift_seed=12345)
    p = inst.m_path_tracer; p.SetInstrumentation(timing=True); p.Trace(True, 8); p.Reset(); p.ResetStats()
    p.Trace(True, 32); s = p.GetStats()
    print(json.dumps({"materials": mode, "trace_Mrays_s": round(s["rays"] / s["trace_ms"] / 1e3, 1), "rays_per_frame": int(s["rays"] / 32), "trace_ms_per_frame": round(s["trace_ms"] / 32, 4)})); sys.stdout.flush()
