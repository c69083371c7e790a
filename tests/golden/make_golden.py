"""Generates the golden vectors under tests/golden/ (build container only: needs oracle/_ref/adypt_ref, i.e. the
reference's own CPU code compiled from /root/reference by oracle/Makefile).

Fixtures are DATA: inputs (tiny OBJ/MTL scenes written by adypt_amd/scenes.py, .config text) and the outputs the
REFERENCE code produced for them.  No reference source text is stored.

  G1  <scene>.tris/.mats/.bvh      Scene::LoadFromFile, OglScene::init_materials, SBVHBuilder+WideBVHBuilder+SaveToFile
      bvh_sha256.json              SHA-256 of the reference-built .bvh for the large stand-ins (sibenik, sponza)
  G2  config_*.json/.config        InstanceConfig::LoadFromFile -> GetJson text; reject cases
  G3  sobol_points_1000x16.f32     Sobol::Next, first 1000 frames x 16 dims; sobol_matrices_64x32.u32 (table rows)
  G4  <scene>.woop                 OglScene::init_triangles
      camera_cases.json/.f32       glm view/projection inverses for a few cameras
  G5  <scene>_kat.npz              traversal known answers from the oracle restatement (checked here against fp64 brute force)
  G6  tiny0_frame_*.npy            tiny end-to-end frames from the oracle (32x18, 4 spp)
      exr_*.rgbaf32                tinyexr SaveEXR -> LoadEXR round trip of a test image (fp16 and fp32)
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from adypt_amd import scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

REF = O.REF_BIN
BVH_ARGS = ["48", "0.3", "1.0"]


def run(*a, **kw):
    return subprocess.run([REF, *map(str, a)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def kat_rays(tris, n, seed):
    rs = np.random.RandomState(seed)
    p = tris["p"].reshape(-1, 3)
    lo, hi = p.min(0) - 0.5, p.max(0) + 0.5
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rs.uniform(lo, hi, size=(n, 3))
    rays[:, 3] = 1e-4
    rays[:, 4:7] = rs.normal(size=(n, 3))
    k = n // 16
    rays[0 * k:1 * k, 4] = 0.0                      # zero x component (2^-64 clamp path)
    rays[1 * k:2 * k, 5] = 0.0
    rays[2 * k:3 * k, 6] = -0.0                     # negative zero: clamped to -2^-64? (dir >= 0 is true for -0 -> +eps)
    rays[3 * k:4 * k, 4:6] = 0.0                    # axis parallel
    rays[4 * k:5 * k, 4:7] *= 1e-30                 # tiny but non-zero directions
    rays[5 * k:6 * k, :3] = p[rs.randint(0, len(p), size=k)]  # origins exactly on vertices
    rays[6 * k:7 * k, 3] = 0.5                      # large tmin
    return rays


def main():
    assert os.path.exists(REF), "build oracle/_ref first (make -C oracle _ref)"
    tmp = tempfile.mkdtemp()
    # ---- G1 / G4 / G5: tiny scenes ----------------------------------------------------------------------------
    for name in ("tiny0", "tiny1", "tiny2"):
        spec = scenes.make_scene(name, tmp, width=64, height=36, force=True)
        tag = os.path.splitext(os.path.basename(spec.obj_path))[0]
        for ext in (".obj", ".mtl"):
            shutil.copy(os.path.join(tmp, tag + ext), os.path.join(HERE, name + ext))
        # the copied OBJ refers to "<tag>.mtl": rewrite the mtllib line to the fixture name
        txt = open(os.path.join(HERE, name + ".obj")).read().replace("mtllib %s.mtl" % tag, "mtllib %s.mtl" % name)
        open(os.path.join(HERE, name + ".obj"), "w").write(txt)
        obj = os.path.join(HERE, name + ".obj")
        run("build", obj, os.path.join(HERE, name + ".bvh"), *BVH_ARGS)
        run("scene", obj, os.path.join(HERE, name + ".tris"), os.path.join(HERE, name + ".mats"))
        run("woop", obj, os.path.join(HERE, name + ".bvh"), *BVH_ARGS, os.path.join(HERE, name + ".woop"))
        _, idx, nodes = O.load_bvh_file(os.path.join(HERE, name + ".bvh"))
        tris = np.fromfile(os.path.join(HERE, name + ".tris"), dtype=O.TRI_DT)
        mats = np.fromfile(os.path.join(HERE, name + ".mats"), dtype=O.MAT_DT)
        woop = np.fromfile(os.path.join(HERE, name + ".woop"), dtype=np.float32).reshape(-1, 12)
        sc = O.Scene(nodes, idx, tris, mats, woop=woop)
        rays = kat_rays(tris, 4096, 17)
        hits = O.trace(sc, rays, 32)
        # the reference clamps |dir_k| to >= 2^-64 *before* normalising (traversal.glsl:16-20), which redirects rays whose
        # un-normalised direction is tiny; give the brute force the same clamped direction
        rays_bf = rays.copy()
        eps = np.float32(2.0 ** -64)
        d = rays_bf[:, 4:7]
        rays_bf[:, 4:7] = np.where(np.abs(d) > eps, d, np.where(d >= 0, eps, -eps))
        bi, bt = O.brute_force(tris, rays_bf)
        mism = hits["tri_id"] != bi
        bad = mism & ~(np.abs(hits["t"].astype(np.float64) - bt) <= 1e-4 * np.maximum(1.0, np.abs(bt)))
        assert bad.sum() == 0, "oracle traversal disagrees with fp64 brute force"
        cam = scenes._SCENE_TABLE[name][3]
        ip, iv = O.camera(cam["fov"], cam["yaw"], cam["pitch"], 64, 36)
        P = O.make_params(64, 36, cam["position"], ip, iv, stack_size=32)
        rgba, phits, _ = O.primary_frame(sc, P, 0)
        np.savez_compressed(os.path.join(HERE, name + "_kat.npz"), rays=rays, hits=hits, primary_hits=phits,
                            primary_rgba=rgba, cam=np.array([cam["fov"], cam["yaw"], cam["pitch"]] + cam["position"], dtype=np.float32))
        print(name, "tris", len(tris), "refs", len(idx), "nodes", len(nodes), "kat mismatches vs brute force (coplanar ties):", int(mism.sum()))
    # ---- G1: SHA-256 of big stand-ins ------------------------------------------------------------------------
    shas = {}
    for name in ("sibenik", "sponza"):
        spec = scenes.make_scene(name, tmp, force=True)
        out = os.path.join(tmp, name + "_ref.bvh")
        r = run("build", spec.obj_path, out, *BVH_ARGS)
        shas[name] = {"bvh_sha256": sha(out), "obj_sha256": sha(spec.obj_path), "n_tris": spec.n_tris,
                      "ref_log": r.stderr.decode().strip().splitlines()[-1]}
        print(name, shas[name])
    json.dump(shas, open(os.path.join(HERE, "bvh_sha256.json"), "w"), indent=1)
    # ---- G2: config --------------------------------------------------------------------------------------------
    good = scenes.config_json(1920, 1080, "scenes/sponza.obj", "scenes/sponza.bvh",
                              dict(scenes._DEFAULT_PT, rayTMin=0.0001, clamp=4.0, sun=[12.0, 11.5, 1e-7]), scenes._DEFAULT_BVH,
                              {"fov": 45.0, "yaw": 270.0, "pitch": -3.25, "position": [-13.0, 2.2, 0.3], "speed": 1.5e21, "mouseSensitive": 0.3})
    cases = {"good": good,
             "int_literal_float": good.replace('"clamp": 4.0', '"clamp": 4'),
             "negative_int": good.replace('"width": 1920', '"width": -1920'),
             "float_for_int": good.replace('"maxBounce": 8', '"maxBounce": 8.0'),
             "sun_len_2": good.replace("            12.0,\n", ""),
             "not_object": "[1, 2, 3]",
             "trailing_garbage": good + " x",
             "exponent_float": good.replace('"fov": 45.0', '"fov": 4.5e1'),
             "string_escape": good.replace("scenes/sponza.obj", "sc\\u00e9nes/a\\\\b \\\"q\\\".obj")}
    results = {}
    for k, text in cases.items():
        p = os.path.join(tmp, k + ".config")
        open(p, "w").write(text)
        if k == "missing":
            continue
        r = subprocess.run([REF, "config", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        results[k] = {"input": text, "accepted": r.returncode == 0, "json": r.stdout.decode() if r.returncode == 0 else None}
        print("config", k, "accepted" if r.returncode == 0 else "rejected(rc=%d)" % r.returncode)
    json.dump(results, open(os.path.join(HERE, "config_cases.json"), "w"), indent=1)
    # ---- G3: Sobol ---------------------------------------------------------------------------------------------
    run("sobol", 16, 1000, os.path.join(HERE, "sobol_points_1000x16.f32"))
    run("sobolmat", 64, os.path.join(HERE, "sobol_matrices_64x32.u32"))
    # ---- G4: cameras -------------------------------------------------------------------------------------------
    cams = [(45.0, 270.0, 0.0, 1920, 1080), (60.0, 200.0, -25.0, 1280, 720), (33.3, 12.5, 89.0, 640, 480), (90.0, 0.0, 0.0, 4096, 4096)]
    blob = []
    for c in cams:
        out = os.path.join(tmp, "cam.bin")
        run("camera", *c, out)
        blob.append(np.fromfile(out, dtype=np.float32))
    np.stack(blob).tofile(os.path.join(HERE, "camera_cases.f32"))
    json.dump(cams, open(os.path.join(HERE, "camera_cases.json"), "w"))
    # ---- G6: tiny end-to-end frames from the oracle ------------------------------------------------------------------
    name = "tiny0"
    _, idx, nodes = O.load_bvh_file(os.path.join(HERE, name + ".bvh"))
    tris = np.fromfile(os.path.join(HERE, name + ".tris"), dtype=O.TRI_DT)
    mats = np.fromfile(os.path.join(HERE, name + ".mats"), dtype=O.MAT_DT)
    sc = O.Scene(nodes, idx, tris, mats)
    cam = scenes._SCENE_TABLE[name][3]
    ip, iv = O.camera(cam["fov"], cam["yaw"], cam["pitch"], 32, 18)
    P = O.make_params(32, 18, cam["position"], ip, iv, stack_size=16, max_bounce=5, subpixel=2, tmp_life=2, tmin=1e-4, clamp=4.0, sun=(12.0, 11.0, 10.0))
    st = O.PathTracerState(32, 18)
    sm = np.fromfile(os.path.join(HERE, "sobol_matrices_64x32.u32"), dtype=np.uint32).reshape(64, 32)
    stats = O.pt_frames(sc, P, O.shift_bytes(4242, 32, 18), sm, st, 4, n_threads=1)
    np.save(os.path.join(HERE, "tiny0_frame_32x18_4spp.npy"), st.accum)
    json.dump(stats.as_dict(), open(os.path.join(HERE, "tiny0_frame_32x18_4spp.json"), "w"))
    # shift bytes of std::mt19937 (first 64 bytes for two seeds)
    np.stack([O.shift_bytes(s, 8, 4).reshape(-1) for s in (0, 4242)]).tofile(os.path.join(HERE, "shift_bytes_seed0_seed4242.u8"))
    # ---- EXR round trip through the reference's tinyexr --------------------------------------------------------------
    rs = np.random.RandomState(5)
    img = (rs.uniform(0, 4, size=(40, 52, 3)) ** 2).astype(np.float32)
    img[0, 0] = [0.0, 1e-8, 65504.0]
    img[1, 1] = [1e6, 6.1e-5, 0.333333]
    img.tofile(os.path.join(HERE, "exr_input_52x40.rgbf32"))
    for fp16 in (0, 1):
        e = os.path.join(tmp, "t%d.exr" % fp16)
        run("exrsave", os.path.join(HERE, "exr_input_52x40.rgbf32"), 52, 40, fp16, e)
        run("exrload", e, os.path.join(HERE, "exr_ref_decoded_fp%d.rgbaf32" % (16 if fp16 else 32)))
    shutil.rmtree(tmp)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
