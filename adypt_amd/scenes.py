"""Deterministic procedural stand-in scenes (OBJ + MTL + PPM textures + Adypt .config).

The assets BASELINE.json names (sibenik / sponza / San Miguel / salle_de_bain) do not exist in the
reference repository, in the build container or on the GPU box, and there is no network
(SURVEY.md §7 "Assets absent", §8d "Concrete inputs").  These generators emit *labelled stand-ins*
with the same triangle counts and the same character (few huge wall/floor triangles next to finely
tessellated columns, arches and cloth, so the SBVH builder's spatial splits trigger), written as
plain OBJ/MTL so that they travel through the very same loader -> builder -> tracer path a real asset
would (reference: src/Util/Scene.cpp:9-136, src/InstanceConfig.cpp:10-101).

If ``$ADYPT_ASSETS/<name>.obj`` exists it is used instead and the scene is labelled ``real``.

All coordinates are float32 and written with ``%.9g`` so that every OBJ parser that is accurate to
1e-9 relative (tinyobj's hand-written one included) reads back identical float32 bits.
Random numbers come from ``numpy.random.RandomState(seed)`` (MT19937; stream is stable across numpy
versions).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

F = np.float32


# --------------------------------------------------------------------------------------------
# mesh accumulation
# --------------------------------------------------------------------------------------------
class Mesh:
    """Triangle soup with per-corner position / normal / uv indices, grouped by material."""

    def __init__(self) -> None:
        self.v: List[np.ndarray] = []      # (n,3) float32 blocks
        self.vn: List[np.ndarray] = []
        self.vt: List[np.ndarray] = []
        self.nv = 0
        self.nvn = 0
        self.nvt = 0
        # list of (material name, faces (m,3) of v idx, vn idx or None, vt idx or None)
        self.groups: List[Tuple[str, np.ndarray, Optional[np.ndarray], Optional[np.ndarray]]] = []

    def add(self, mtl: str, v: np.ndarray, f: np.ndarray, vn: Optional[np.ndarray] = None,
            vt: Optional[np.ndarray] = None) -> None:
        """v (n,3); f (m,3) indexing v; vn/vt optional per-vertex (n,3)/(n,2) sharing f."""
        v = np.ascontiguousarray(v, dtype=F)
        f = np.ascontiguousarray(f, dtype=np.int64)
        self.v.append(v)
        fn = ft = None
        if vn is not None:
            self.vn.append(np.ascontiguousarray(vn, dtype=F))
            fn = f + self.nvn
            self.nvn += len(vn)
        if vt is not None:
            self.vt.append(np.ascontiguousarray(vt, dtype=F))
            ft = f + self.nvt
            self.nvt += len(vt)
        self.groups.append((mtl, f + self.nv, fn, ft))
        self.nv += len(v)

    @property
    def n_tris(self) -> int:
        return int(sum(len(g[1]) for g in self.groups))

    # -- OBJ ---------------------------------------------------------------------------------
    def write_obj(self, path: str, mtllib: str) -> None:
        def fmt_rows(prefix: str, a: np.ndarray) -> str:
            cols = a.shape[1]
            fmt = prefix + " " + " ".join(["%.9g"] * cols)
            return "\n".join(fmt % tuple(r) for r in a.tolist())

        with open(path, "w") as out:
            out.write("# adypt_amd procedural stand-in scene\nmtllib %s\n" % mtllib)
            for blk in self.v:
                out.write(fmt_rows("v", blk.astype(np.float64)))
                out.write("\n")
            for blk in self.vn:
                out.write(fmt_rows("vn", blk.astype(np.float64)))
                out.write("\n")
            for blk in self.vt:
                out.write(fmt_rows("vt", blk.astype(np.float64)))
                out.write("\n")
            for gi, (mtl, f, fn, ft) in enumerate(self.groups):
                out.write("g part%d\nusemtl %s\n" % (gi, mtl))
                f1 = f + 1
                if fn is not None and ft is not None:
                    a = np.stack([f1, ft + 1, fn + 1], axis=2).reshape(len(f), 9)
                    fmt = "f %d/%d/%d %d/%d/%d %d/%d/%d"
                elif fn is not None:
                    a = np.stack([f1, fn + 1], axis=2).reshape(len(f), 6)
                    fmt = "f %d//%d %d//%d %d//%d"
                elif ft is not None:
                    a = np.stack([f1, ft + 1], axis=2).reshape(len(f), 6)
                    fmt = "f %d/%d %d/%d %d/%d"
                else:
                    a = f1
                    fmt = "f %d %d %d"
                out.write("\n".join(fmt % tuple(r) for r in a.tolist()))
                out.write("\n")


@dataclass
class Material:
    name: str
    Kd: Tuple[float, float, float] = (0.7, 0.7, 0.7)
    Ks: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    Ke: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    illum: int = 1
    Ns: float = 1.0
    Ni: float = 1.0
    d: float = 1.0
    map_Kd: Optional[str] = None

    def mtl_text(self) -> str:
        s = ["newmtl %s" % self.name,
             "Kd %.9g %.9g %.9g" % self.Kd,
             "Ks %.9g %.9g %.9g" % self.Ks,
             "Ke %.9g %.9g %.9g" % self.Ke,
             "Ns %.9g" % self.Ns, "Ni %.9g" % self.Ni, "d %.9g" % self.d,
             "illum %d" % self.illum]
        if self.map_Kd:
            s.append("map_Kd %s" % self.map_Kd)
        return "\n".join(s) + "\n\n"


# --------------------------------------------------------------------------------------------
# primitive builders (all return v, f[, vn][, vt])
# --------------------------------------------------------------------------------------------
def grid_patch(p00, pu, pv, nu: int, nv: int, flip: bool = False, uv_scale=(1.0, 1.0)):
    """Planar patch p00 + s*pu + t*pv, s,t in [0,1], nu x nv quads."""
    p00, pu, pv = (np.asarray(x, dtype=np.float64) for x in (p00, pu, pv))
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    v = p00[None, None, :] + S[..., None] * pu + T[..., None] * pv
    n = np.cross(pu, pv)
    n = n / np.linalg.norm(n)
    if flip:
        n = -n
    vn = np.broadcast_to(n, v.shape).reshape(-1, 3)
    vt = np.stack([S * uv_scale[0], T * uv_scale[1]], axis=-1).reshape(-1, 2)
    f = _grid_faces(nu, nv, flip)
    return v.reshape(-1, 3), f, vn, vt


def _grid_faces(nu: int, nv: int, flip: bool = False, wrap_u: bool = False) -> np.ndarray:
    cols = nv + 1
    rows = nu if wrap_u else nu + 1
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1 = (i + 1) % rows if wrap_u else i + 1
    a = i * cols + j
    b = i1 * cols + j
    c = i1 * cols + j + 1
    d = i * cols + j + 1
    t1 = np.stack([a, b, c], axis=-1).reshape(-1, 3)
    t2 = np.stack([a, c, d], axis=-1).reshape(-1, 3)
    f = np.concatenate([t1, t2], axis=0)
    # interleave the two triangles of each quad (better locality in file order)
    f = np.stack([t1, t2], axis=1).reshape(-1, 3)
    if flip:
        f = f[:, ::-1]
    return f


def revolve(profile_r, profile_y, center_xz, nseg: int, smooth: bool = True):
    """Surface of revolution around the y axis through center_xz."""
    r = np.asarray(profile_r, dtype=np.float64)
    y = np.asarray(profile_y, dtype=np.float64)
    ang = np.arange(nseg) * (2.0 * np.pi / nseg)
    ca, sa = np.cos(ang), np.sin(ang)
    v = np.stack([center_xz[0] + ca[:, None] * r[None, :],
                  np.broadcast_to(y, (nseg, len(y))),
                  center_xz[1] + sa[:, None] * r[None, :]], axis=-1)
    # normals from the profile tangent
    dr = np.gradient(r)
    dy = np.gradient(y)
    nr, ny = dy, -dr
    ln = np.sqrt(nr * nr + ny * ny)
    ln[ln == 0] = 1.0
    nr, ny = nr / ln, ny / ln
    vn = np.stack([ca[:, None] * nr[None, :], np.broadcast_to(ny, (nseg, len(y))),
                   sa[:, None] * nr[None, :]], axis=-1)
    f = _grid_faces(nseg, len(y) - 1, flip=True, wrap_u=True)
    return v.reshape(-1, 3), f, (vn.reshape(-1, 3) if smooth else None)


def uv_sphere(center, radius: float, nseg: int, nring: int):
    th = np.linspace(0.0, np.pi, nring + 1)
    r = radius * np.sin(th)
    y = center[1] - radius * np.cos(th)
    v, f, _ = revolve(r, y, (center[0], center[2]), nseg)
    vn = (v - np.asarray(center)[None, :]) / radius
    return v, f, vn


def arch(p0, p1, radius: float, thick: float, depth: float, nseg: int, nd: int):
    """Half-ring arch spanning p0 -> p1 (same height), extruded along the horizontal normal."""
    p0 = np.asarray(p0, dtype=np.float64)
    p1 = np.asarray(p1, dtype=np.float64)
    mid = 0.5 * (p0 + p1)
    ax = (p1 - p0)
    L = np.linalg.norm(ax)
    ax = ax / L
    up = np.array([0.0, 1.0, 0.0])
    nrm = np.cross(ax, up)
    th = np.linspace(0.0, np.pi, nseg + 1)
    out_v, out_f, out_n = [], [], []
    base = 0
    for rr, flip in ((radius, True), (radius + thick, False)):
        d = np.linspace(-0.5 * depth, 0.5 * depth, nd + 1)
        TH, D = np.meshgrid(th, d, indexing="ij")
        pts = (mid[None, None, :] + (-np.cos(TH))[..., None] * ax * rr + np.sin(TH)[..., None] * up * rr
               + D[..., None] * nrm)
        n = (-np.cos(TH))[..., None] * ax + np.sin(TH)[..., None] * up
        if flip:
            n = -n
        out_v.append(pts.reshape(-1, 3))
        out_n.append(n.reshape(-1, 3))
        out_f.append(_grid_faces(nseg, nd, flip=flip) + base)
        base += pts.shape[0] * pts.shape[1]
    # front/back faces of the ring
    for side, flip in ((-0.5 * depth, False), (0.5 * depth, True)):
        rr = np.array([radius, radius + thick])
        TH, R = np.meshgrid(th, rr, indexing="ij")
        pts = (mid[None, None, :] + (-np.cos(TH))[..., None] * ax * R[..., None]
               + np.sin(TH)[..., None] * up * R[..., None] + side * nrm)
        n = np.broadcast_to(nrm * (1.0 if side > 0 else -1.0), pts.shape)
        out_v.append(pts.reshape(-1, 3))
        out_n.append(n.reshape(-1, 3))
        out_f.append(_grid_faces(nseg, 1, flip=flip) + base)
        base += pts.shape[0] * pts.shape[1]
    return np.concatenate(out_v), np.concatenate(out_f), np.concatenate(out_n)


def cloth(p00, pu, pv, nu: int, nv: int, amp: float, waves: float, rs: np.random.RandomState):
    """Wavy hanging cloth: a displaced grid patch with smooth normals."""
    v, f, vn, vt = grid_patch(p00, pu, pv, nu, nv)
    n0 = vn[0].astype(np.float64)
    s = vt[:, 0].astype(np.float64)
    t = vt[:, 1].astype(np.float64)
    ph = rs.uniform(0, 2 * np.pi, size=3)
    disp = amp * (np.sin(2 * np.pi * waves * s + ph[0]) * (0.3 + 0.7 * t)
                  + 0.35 * np.sin(2 * np.pi * (2.3 * waves) * s + 5.0 * t + ph[1])
                  + 0.15 * np.sin(2 * np.pi * 7.0 * t + ph[2]))
    v = v + disp[:, None] * n0[None, :]
    # smooth normals from the displaced grid
    V = v.reshape(nu + 1, nv + 1, 3)
    du = np.gradient(V, axis=0)
    dv = np.gradient(V, axis=1)
    n = np.cross(du, dv)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return v, f, n.reshape(-1, 3), vt


def blob(center, radius: float, nseg: int, nring: int, rs: np.random.RandomState, rough: float = 0.15):
    """Bumpy sphere ('lion head' / vase stand-in): dense small triangles."""
    v, f, vn = uv_sphere(center, radius, nseg, nring)
    d = (v - np.asarray(center)[None, :]) / radius
    k = rs.uniform(1.5, 6.0, size=(6, 3))
    ph = rs.uniform(0, 2 * np.pi, size=6)
    bump = sum(np.sin(d @ k[i] * 2.0 + ph[i]) for i in range(6)) / 6.0
    v = np.asarray(center)[None, :] + d * (radius * (1.0 + rough * bump))[:, None]
    return v, f, d


def box(lo, hi, inward: bool = False):
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5)]
    v, f, vn = [], [], []
    for q in quads:
        p = c[list(q)]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n = n / np.linalg.norm(n)
        b = len(v)
        v.extend(p)
        vn.extend([n] * 4)
        f.append((b, b + 1, b + 2))
        f.append((b, b + 2, b + 3))
    v = np.array(v)
    f = np.array(f)
    vn = np.array(vn)
    if inward:
        f = f[:, ::-1]
        vn = -vn
    return v, f, vn


# --------------------------------------------------------------------------------------------
# textures
# --------------------------------------------------------------------------------------------
def write_ppm_checker(path: str, size: int, cells: int, c0, c1, rs: np.random.RandomState) -> None:
    y, x = np.mgrid[0:size, 0:size]
    chk = ((x * cells // size) + (y * cells // size)) & 1
    img = np.where(chk[..., None] == 0, np.array(c0)[None, None, :], np.array(c1)[None, None, :]).astype(np.int32)
    img = np.clip(img + rs.randint(-12, 13, size=img.shape), 0, 255).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (size, size))
        f.write(img.tobytes())


# --------------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------------
@dataclass
class SceneSpec:
    name: str
    label: str            # "stand-in" or "real"
    obj_path: str
    config_path: str
    n_tris: int
    config: Dict = field(default_factory=dict)


_DEFAULT_PT = {"invocationSize": 8, "stackSize": 24, "maxBounce": 8, "subpixel": 8, "tmpLifetime": 16,
               "rayTMin": 0.0001, "clamp": 4.0, "sun": [12.0, 11.0, 10.0]}
_DEFAULT_BVH = {"maxSpatialDepth": 48, "triangleSAH": 0.3, "nodeSAH": 1.0}


def _fl(x: float) -> float:
    return float(np.float32(x))


def config_json(width: int, height: int, obj: str, bvh: str, pt: Dict, bvhp: Dict, cam: Dict) -> str:
    """Adypt .config text.  Floats must carry a decimal point (rapidjson IsFloat)."""
    def jf(x) -> str:
        s = repr(float(x))
        if "e" in s or "E" in s:
            s = "%.10f" % float(x)
        if "." not in s:
            s += ".0"
        return s

    lines = ["{",
             '    "width": %d,' % width,
             '    "height": %d,' % height,
             '    "scene": {',
             '        "filename": %s' % json.dumps(obj),
             "    },",
             '    "pathTracer": {',
             '        "invocationSize": %d,' % pt["invocationSize"],
             '        "stackSize": %d,' % pt["stackSize"],
             '        "maxBounce": %d,' % pt["maxBounce"],
             '        "subpixel": %d,' % pt["subpixel"],
             '        "tmpLifetime": %d,' % pt["tmpLifetime"],
             '        "rayTMin": %s,' % jf(pt["rayTMin"]),
             '        "clamp": %s,' % jf(pt["clamp"]),
             '        "sun": [',
             "            %s," % jf(pt["sun"][0]),
             "            %s," % jf(pt["sun"][1]),
             "            %s" % jf(pt["sun"][2]),
             "        ]",
             "    },",
             '    "bvh": {',
             '        "filename": %s,' % json.dumps(bvh),
             '        "maxSpatialDepth": %d,' % bvhp["maxSpatialDepth"],
             '        "triangleSAH": %s,' % jf(bvhp["triangleSAH"]),
             '        "nodeSAH": %s' % jf(bvhp["nodeSAH"]),
             "    },",
             '    "camera": {',
             '        "speed": %s,' % jf(cam.get("speed", 1.0)),
             '        "mouseSensitive": %s,' % jf(cam.get("mouseSensitive", 0.3)),
             '        "fov": %s,' % jf(cam["fov"]),
             '        "yaw": %s,' % jf(cam["yaw"]),
             '        "pitch": %s,' % jf(cam["pitch"]),
             '        "position": [',
             "            %s," % jf(cam["position"][0]),
             "            %s," % jf(cam["position"][1]),
             "            %s" % jf(cam["position"][2]),
             "        ]",
             "    }",
             "}"]
    return "\n".join(lines)


def _atrium(mesh: Mesh, rs: np.random.RandomState, detail: float, mats: Dict[str, Material], out_dir: str,
            tex_name: str) -> None:
    """Two-storey colonnaded atrium, open to the sky ('sponza-like')."""
    L, Wd, H = 30.0, 12.0, 14.0                 # x in [-15,15], z in [-6,6], y in [0,14]
    x0, x1, z0, z1 = -L / 2, L / 2, -Wd / 2, Wd / 2

    mats["floor"] = Material("floor", Kd=(0.8, 0.8, 0.8), illum=1, map_Kd=tex_name)
    mats["wall"] = Material("wall", Kd=(0.72, 0.66, 0.58), illum=1)
    mats["stone"] = Material("stone", Kd=(0.62, 0.6, 0.56), illum=1)
    mats["arch"] = Material("arch", Kd=(0.55, 0.5, 0.45), Ks=(0.2, 0.2, 0.2), illum=2, Ns=60.0)
    mats["cloth_r"] = Material("cloth_r", Kd=(0.7, 0.12, 0.1), illum=1)
    mats["cloth_g"] = Material("cloth_g", Kd=(0.12, 0.55, 0.2), illum=1)
    mats["cloth_b"] = Material("cloth_b", Kd=(0.12, 0.2, 0.65), illum=1)
    mats["mirror"] = Material("mirror", Kd=(0.0, 0.0, 0.0), Ks=(0.92, 0.92, 0.92), illum=3)
    mats["glass"] = Material("glass", Kd=(0.0, 0.0, 0.0), Ks=(1.0, 1.0, 1.0), illum=7, Ni=1.5)
    mats["gloss"] = Material("gloss", Kd=(0.25, 0.2, 0.1), Ks=(0.6, 0.55, 0.4), illum=2, Ns=400.0)
    mats["lamp"] = Material("lamp", Kd=(0.0, 0.0, 0.0), Ke=(18.0, 15.0, 11.0), illum=1)
    mats["bronze"] = Material("bronze", Kd=(0.45, 0.3, 0.15), Ks=(0.3, 0.25, 0.2), illum=2, Ns=25.0)

    # -- floor + ground-floor walls: few huge triangles -----------------------------------
    mesh.add("floor", *_pick(grid_patch((x0, 0, z0), (0, 0, Wd), (L, 0, 0), 2, 3, uv_scale=(6.0, 15.0)), True, True))
    mesh.add("wall", *_pick(grid_patch((x0, 0, z0), (L, 0, 0), (0, H, 0), 3, 2), True, False))          # z = z0, faces +z
    mesh.add("wall", *_pick(grid_patch((x0, 0, z1), (0, H, 0), (L, 0, 0), 2, 3), True, False))          # z = z1, faces -z
    mesh.add("wall", *_pick(grid_patch((x0, 0, z0), (0, H, 0), (0, 0, Wd), 2, 2), True, False))         # x = x0, faces +x
    mesh.add("wall", *_pick(grid_patch((x1, 0, z0), (0, 0, Wd), (0, H, 0), 2, 2), True, False))         # x = x1, faces -x
    # upper gallery floors (slabs) along both long sides
    gal = 2.6
    for zz0, zz1 in ((z0, z0 + gal), (z1 - gal, z1)):
        v, f, vn = box((x0, 6.0, zz0), (x1, 6.35, zz1))
        mesh.add("stone", v, f, vn)

    # -- columns ----------------------------------------------------------------------------
    nseg = max(8, int(round(40 * detail)))
    nprof = max(6, int(round(22 * detail)))
    col_x = np.linspace(x0 + 2.0, x1 - 2.0, 10)
    def column(cx, cz, y_lo, y_hi, r):
        t = np.linspace(0.0, 1.0, nprof + 1)
        rr = r * (1.0 - 0.12 * t + 0.22 * np.exp(-((t - 0.0) / 0.06) ** 2) + 0.26 * np.exp(-((t - 1.0) / 0.05) ** 2)
                  + 0.015 * np.cos(t * 40.0))
        yy = y_lo + (y_hi - y_lo) * t
        v, f, vn = revolve(rr, yy, (cx, cz), nseg)
        mesh.add("stone", v, f, vn)
    for cz in (z0 + gal, z1 - gal):
        for cx in col_x:
            column(cx, cz, 0.0, 5.2, 0.34)
            column(cx, cz, 6.35, 10.6, 0.26)
    # -- arches between columns (both storeys) --------------------------------------------------
    aseg = max(6, int(round(28 * detail)))
    adep = max(1, int(round(5 * detail)))
    span = col_x[1] - col_x[0]
    for cz in (z0 + gal, z1 - gal):
        for i in range(len(col_x) - 1):
            for yb, th in ((5.2 - 0.0, 0.45), (10.6, 0.35)):
                r = 0.5 * span - 0.36
                v, f, vn = arch((col_x[i] + 0.36, yb - r * 0.0, cz), (col_x[i + 1] - 0.36, yb, cz), r * 0.999, th, 0.7,
                                aseg, adep)
                v[:, 1] -= r * 0.55  # spring the arch below the capital line
                mesh.add("arch", v, f, vn)
    # -- cloth banners hanging across the nave -----------------------------------------------
    cu = max(8, int(round(72 * detail)))
    cv = max(8, int(round(72 * detail)))
    cm = ["cloth_r", "cloth_g", "cloth_b"]
    for k, cx in enumerate(np.linspace(x0 + 5.0, x1 - 5.0, 6)):
        v, f, vn, vt = cloth((cx, 11.6, z0 + gal + 0.5), (0.0, 0.0, Wd - 2 * gal - 1.0), (0.25, -5.0, 0.0), cu, cv, 0.22, 2.0 + 0.5 * k, rs)
        mesh.add(cm[k % 3], v, f, vn, vt)
    # long side drapes on the upper gallery
    for k, cz in enumerate((z0 + 0.15, z1 - 0.15)):
        v, f, vn, vt = cloth((x0 + 3.0, 10.0, cz), (L - 6.0, 0.0, 0.0), (0.0, -3.2, 0.0), cu * 2, cv // 2, 0.1, 9.0, rs)
        mesh.add(cm[(k + 1) % 3], v, f, vn, vt)
    # -- furniture: mirror / glass / glossy spheres, bronze blobs, a lamp ----------------------------
    sseg = max(12, int(round(72 * detail)))
    sring = max(8, int(round(40 * detail)))
    v, f, vn = uv_sphere((-6.0, 1.0, -0.8), 1.0, sseg, sring); mesh.add("mirror", v, f, vn)
    v, f, vn = uv_sphere((-2.5, 0.8, 1.2), 0.8, sseg, sring); mesh.add("glass", v, f, vn)
    v, f, vn = uv_sphere((1.5, 0.7, -1.0), 0.7, sseg, sring); mesh.add("gloss", v, f, vn)
    bseg = max(16, int(round(150 * detail)))
    bring = max(10, int(round(90 * detail)))
    v, f, vn = blob((6.0, 1.3, 0.6), 1.1, bseg, bring, rs); mesh.add("bronze", v, f, vn)
    v, f, vn = blob((10.5, 0.9, -1.4), 0.8, bseg, bring, rs, rough=0.25); mesh.add("bronze", v, f, vn)
    v, f, vn = box((-1.5, 9.2, -0.6), (1.5, 9.3, 0.6)); mesh.add("lamp", v, f, vn)
    # pedestal boxes with long thin slabs crossing the nave (forces spatial splits)
    for k in range(5):
        xx = -12.0 + 6.0 * k
        v, f, vn = box((xx - 0.05, 5.6, z0 + gal), (xx + 0.05, 5.75, z1 - gal)); mesh.add("stone", v, f, vn)
    write_ppm_checker(os.path.join(out_dir, tex_name), 256, 16, (190, 180, 160), (90, 70, 60), rs)


def _pick(t, with_n: bool, with_t: bool):
    v, f, vn, vt = t
    return v, f, (vn if with_n else None), (vt if with_t else None)


def _cathedral(mesh: Mesh, rs: np.random.RandomState, detail: float, mats: Dict[str, Material]) -> None:
    """Closed nave with two rows of columns and a vaulted ceiling grid ('sibenik-like')."""
    L, Wd, H = 40.0, 16.0, 15.0
    x0, x1, z0, z1 = -L / 2, L / 2, -Wd / 2, Wd / 2
    mats["floor"] = Material("floor", Kd=(0.6, 0.58, 0.55), Ks=(0.15, 0.15, 0.15), illum=2, Ns=45.0)
    mats["wall"] = Material("wall", Kd=(0.7, 0.66, 0.6), illum=1)
    mats["stone"] = Material("stone", Kd=(0.62, 0.6, 0.56), illum=1)
    mats["window"] = Material("window", Kd=(0.0, 0.0, 0.0), Ke=(9.0, 8.0, 6.5), illum=1)
    mats["gold"] = Material("gold", Kd=(0.1, 0.08, 0.02), Ks=(0.9, 0.75, 0.3), illum=3)
    mesh.add("floor", *_pick(grid_patch((x0, 0, z0), (0, 0, Wd), (L, 0, 0), 2, 4), True, False))
    mesh.add("wall", *_pick(grid_patch((x0, 0, z0), (L, 0, 0), (0, H, 0), 4, 2), True, False))
    mesh.add("wall", *_pick(grid_patch((x0, 0, z1), (0, H, 0), (L, 0, 0), 2, 4), True, False))
    mesh.add("wall", *_pick(grid_patch((x0, 0, z0), (0, H, 0), (0, 0, Wd), 2, 2), True, False))
    mesh.add("wall", *_pick(grid_patch((x1, 0, z0), (0, 0, Wd), (0, H, 0), 2, 2), True, False))
    # vaulted ceiling: barrel vault as a fine grid
    nu = max(8, int(round(120 * detail)))
    nv = max(8, int(round(60 * detail)))
    s = np.linspace(0, 1, nu + 1)
    t = np.linspace(0, 1, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    ang = np.pi * T
    v = np.stack([x0 + L * S, H - 0.01 + 3.0 * np.sin(ang) + 0.25 * np.abs(np.sin(S * np.pi * 10)), z0 + Wd * (0.5 - 0.5 * np.cos(ang))], axis=-1)
    du = np.gradient(v, axis=0)
    dv = np.gradient(v, axis=1)
    n = np.cross(du, dv)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    mesh.add("stone", v.reshape(-1, 3), _grid_faces(nu, nv, flip=False), n.reshape(-1, 3))
    nseg = max(8, int(round(36 * detail)))
    nprof = max(6, int(round(30 * detail)))
    for cz in (z0 + 4.0, z1 - 4.0):
        for cx in np.linspace(x0 + 3.0, x1 - 3.0, 9):
            tt = np.linspace(0, 1, nprof + 1)
            rr = 0.55 * (1.0 - 0.1 * tt + 0.3 * np.exp(-(tt / 0.05) ** 2) + 0.35 * np.exp(-((tt - 1) / 0.04) ** 2) + 0.02 * np.cos(tt * 60))
            v, f, vn = revolve(rr, 13.5 * tt, (cx, cz), nseg)
            mesh.add("stone", v, f, vn)
    # emissive windows high on the long walls
    for cx in np.linspace(x0 + 5.0, x1 - 5.0, 7):
        v, f, vn = box((cx - 0.9, 8.0, z0 + 0.01), (cx + 0.9, 12.0, z0 + 0.06)); mesh.add("window", v, f, vn)
        v, f, vn = box((cx - 0.9, 8.0, z1 - 0.06), (cx + 0.9, 12.0, z1 - 0.01)); mesh.add("window", v, f, vn)
    v, f, vn = uv_sphere((14.0, 1.6, 0.0), 1.6, max(12, int(64 * detail)), max(8, int(36 * detail))); mesh.add("gold", v, f, vn)
    v, f, vn = blob((-12.0, 1.5, 1.0), 1.3, max(16, int(110 * detail)), max(10, int(70 * detail)), rs); mesh.add("stone", v, f, vn)


def _tiny(mesh: Mesh, rs: np.random.RandomState, variant: int, mats: Dict[str, Material], out_dir: str) -> None:
    """Few-hundred-triangle fixtures for golden vectors (tests/golden)."""
    mats["white"] = Material("white", Kd=(0.75, 0.75, 0.75), illum=1)
    mats["red"] = Material("red", Kd=(0.75, 0.15, 0.15), illum=1)
    mats["green"] = Material("green", Kd=(0.15, 0.75, 0.15), illum=1)
    mats["mirror"] = Material("mirror", Ks=(0.9, 0.9, 0.9), Kd=(0, 0, 0), illum=3)
    mats["glass"] = Material("glass", Ks=(1, 1, 1), Kd=(0, 0, 0), illum=7, Ni=1.5)
    mats["gloss"] = Material("gloss", Kd=(0.3, 0.3, 0.1), Ks=(0.5, 0.5, 0.5), illum=2, Ns=200.0)
    mats["matte2"] = Material("matte2", Kd=(0.3, 0.3, 0.6), Ks=(0.5, 0.5, 0.5), illum=2, Ns=10.0)
    mats["lamp"] = Material("lamp", Kd=(0, 0, 0), Ke=(15.0, 15.0, 12.0), illum=1)
    mats["pass"] = Material("pass", Kd=(0.5, 0.5, 0.5), illum=0)
    if variant == 0:
        # Cornell-like open box (open top -> sun), mixed materials
        mesh.add("white", *_pick(grid_patch((-2, 0, -2), (0, 0, 4), (4, 0, 0), 2, 2), True, False))
        mesh.add("red", *_pick(grid_patch((-2, 0, -2), (0, 4, 0), (0, 0, 4), 1, 1), True, False))
        mesh.add("green", *_pick(grid_patch((2, 0, -2), (0, 0, 4), (0, 4, 0), 1, 1), True, False))
        mesh.add("white", *_pick(grid_patch((-2, 0, -2), (4, 0, 0), (0, 4, 0), 1, 1), True, False))
        v, f, vn = uv_sphere((-0.9, 0.7, -0.5), 0.7, 12, 8); mesh.add("mirror", v, f, vn)
        v, f, vn = uv_sphere((0.9, 0.6, 0.4), 0.6, 12, 8); mesh.add("glass", v, f, vn)
        v, f, vn = uv_sphere((0.0, 0.4, 1.2), 0.4, 10, 6); mesh.add("gloss", v, f, vn)
        v, f, vn = box((-0.5, 3.6, -0.5), (0.5, 3.7, 0.5)); mesh.add("lamp", v, f, vn)
        v, f, vn = box((-1.8, 0.0, 1.0), (-1.2, 1.2, 1.6)); mesh.add("matte2", v, f, vn)
        v, f, _ = box((1.2, 0.0, -1.7), (1.7, 0.9, -1.2)); mesh.add("pass", v, f, None)   # no normals -> generated
    elif variant == 1:
        # long thin crossing slabs + scattered small triangles: forces spatial splits
        for k in range(6):
            a = k * np.pi / 6
            d = np.array([np.cos(a), 0.0, np.sin(a)]) * 6.0
            p = np.array([-d[0] / 2, 0.2 * k, -d[2] / 2])
            v = np.array([p, p + d, p + d + np.array([0, 0.05, 0]), p + np.array([0, 0.05, 0])])
            mesh.add("white" if k % 2 else "red", v, np.array([[0, 1, 2], [0, 2, 3]]))
        c = rs.uniform(-3, 3, size=(120, 3))
        c[:, 1] = rs.uniform(0, 2, size=120)
        for i in range(120):
            tri = c[i][None, :] + rs.uniform(-0.15, 0.15, size=(3, 3))
            mesh.add("green" if i % 3 else "gloss", tri, np.array([[0, 1, 2]]))
        mesh.add("white", *_pick(grid_patch((-4, -0.2, -4), (0, 0, 8), (8, 0, 0), 1, 1), True, False))
    else:
        # root with fewer than 8 children: 5 well separated triangles (+ degenerate axis-aligned quad)
        for i in range(5):
            o = np.array([i * 3.0, 0.0, 0.0])
            mesh.add("white", np.array([o, o + [1, 0, 0], o + [0, 1, 0]]), np.array([[0, 1, 2]]))


_SCENE_TABLE = {
    # name: (builder, seed, default detail, camera, W, H)
    "sponza": ("atrium", 2, 1.0, {"fov": 45.0, "yaw": 270.0, "pitch": 0.0, "position": [-13.0, 2.2, 0.3]}),
    "sibenik": ("cathedral", 1, 1.0, {"fov": 50.0, "yaw": 270.0, "pitch": 5.0, "position": [-18.0, 3.0, 0.0]}),
    # "salle-de-bain-like" (~1.2 M triangles) and "san-miguel-like" (~10 M): the same atrium generator at higher
    # tessellation (triangle count grows with detail^2); stand-ins for BASELINE.json configs 5 and 4
    "salle": ("atrium", 4, 2.2, {"fov": 50.0, "yaw": 250.0, "pitch": -8.0, "position": [-11.0, 3.5, 2.5]}),
    "sanmiguel": ("atrium", 3, 6.35, {"fov": 55.0, "yaw": 285.0, "pitch": 4.0, "position": [-12.5, 7.5, -1.0]}),
    "tiny0": ("tiny0", 10, 1.0, {"fov": 45.0, "yaw": 0.0, "pitch": -10.0, "position": [0.0, 2.0, 6.5]}),
    "tiny1": ("tiny1", 11, 1.0, {"fov": 60.0, "yaw": 200.0, "pitch": -25.0, "position": [1.5, 4.0, -6.0]}),
    "tiny2": ("tiny2", 12, 1.0, {"fov": 60.0, "yaw": 180.0, "pitch": 0.0, "position": [6.0, 0.5, -10.0]}),
}


def make_scene(name: str, out_dir: str, width: int = 1920, height: int = 1080, detail: Optional[float] = None,
               pt: Optional[Dict] = None, bvh: Optional[Dict] = None, camera: Optional[Dict] = None,
               force: bool = False) -> SceneSpec:
    """Generate (or reuse) scene `name` under out_dir; returns paths + the config dict."""
    os.makedirs(out_dir, exist_ok=True)
    kind, seed, det0, cam0 = _SCENE_TABLE[name]
    detail = det0 if detail is None else detail
    tag = "%s_d%03d" % (name, int(round(detail * 100)))
    pt_cfg = dict(_DEFAULT_PT)
    pt_cfg.update(pt or {})
    bvh_cfg = dict(_DEFAULT_BVH)
    bvh_cfg.update(bvh or {})
    cam = dict(cam0)
    cam.update(camera or {})

    assets = os.environ.get("ADYPT_ASSETS")
    real = os.path.join(assets, name + ".obj") if assets else None
    if real and os.path.exists(real):
        obj_path, label = real, "real"
        n_tris = -1
    else:
        label = "stand-in"
        obj_path = os.path.join(out_dir, tag + ".obj")
        meta_path = os.path.join(out_dir, tag + ".meta.json")
        if force or not (os.path.exists(obj_path) and os.path.exists(meta_path)):
            rs = np.random.RandomState(seed)
            mesh = Mesh()
            mats: Dict[str, Material] = {}
            if kind == "atrium":
                _atrium(mesh, rs, detail, mats, out_dir, tag + "_floor.ppm")
            elif kind == "cathedral":
                _cathedral(mesh, rs, detail, mats)
            else:
                _tiny(mesh, rs, int(kind[-1]), mats, out_dir)
            with open(os.path.join(out_dir, tag + ".mtl"), "w") as f:
                for m in mats.values():
                    f.write(m.mtl_text())
            mesh.write_obj(obj_path + ".tmp", tag + ".mtl")
            os.replace(obj_path + ".tmp", obj_path)
            with open(meta_path, "w") as f:
                json.dump({"n_tris": mesh.n_tris, "seed": seed, "detail": detail}, f)
        with open(meta_path) as f:
            n_tris = json.load(f)["n_tris"]

    bvh_path = os.path.join(out_dir, tag + ".bvh")
    cfg_path = os.path.join(out_dir, "%s_%dx%d.config" % (tag, width, height))
    with open(cfg_path, "w") as f:
        f.write(config_json(width, height, obj_path, bvh_path, pt_cfg, bvh_cfg, cam))
    cfg = {"width": width, "height": height, "pathTracer": pt_cfg, "bvh": bvh_cfg, "camera": cam,
           "scene": obj_path, "bvh_file": bvh_path}
    return SceneSpec(name=name, label=label, obj_path=obj_path, config_path=cfg_path, n_tris=n_tris, config=cfg)
