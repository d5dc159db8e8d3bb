"""Host-side scene preparation for tests and bench.py (numpy + libnova_scene.so).

Produces the inputs of the hot path in the layouts of ``include/nova_hip.h``: what Nova's own host
code hands to ``Accelerator::create_buffer`` / ``create_image2D_*`` in
``Raytracer::set_scene`` (reference ``src/core/raytracer.cpp:47-167``).  Asset import (assimp) and
the BVH builder are OUT of the hot-path scope; this module only has to feed it.

* numpy dtypes mirroring ``src/kernel_types/*.hpp`` in the HIP/CUDA host layout,
* ``Camera.eye_coords`` = ``Camera::get_eye_coords`` (``src/camera/camera.cpp:89-105``),
* a small OBJ/MTL reader with assimp's ``PTV_NORMALIZE`` ([-1,1] normalisation, used by
  ``src/model/model.cpp:22-28``), face normals where the file has none,
* seeded procedural stand-ins for the assets that are not in the reference tree
  (``.MISSING_LARGE_BLOBS``): dragon, fireplace_room, sky HDR, and the 10 M-triangle soup.
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# ------------------------------------------------------------------------------------------------
# POD layouts (include/nova_hip.h)
# ------------------------------------------------------------------------------------------------
F3 = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]

BVH_NODE = np.dtype([("top_offset_left", "<f4", 4), ("bottom_num_right", "<f4", 4)])
TRIANGLE_DATA = np.dtype([("x", "<f4", 4), ("y", "<f4", 4), ("z", "<f4", 4)])
TRIANGLE_META = np.dtype(
    {
        "names": [
            "normal1", "normal2", "normal3", "tangent1", "tangent2", "tangent3",
            "bitangent1", "bitangent2", "bitangent3",
            "texture_coord1", "texture_coord2", "texture_coord3",
            "kD", "metallic", "roughness",
            "diffuse_index", "metallic_index", "roughness_index", "normal_index", "light_index",
        ],
        "formats": [("<f4", 3)] * 9 + [("<f4", 2)] * 3 + [("<f4", 3), "<f4", "<f4"] + ["<i4"] * 5,
        "offsets": [0, 12, 24, 36, 48, 60, 72, 84, 96, 112, 120, 128, 136, 148, 152, 156, 160, 164,
                    168, 172],
        "itemsize": 176,
    }
)
AREA_LIGHT = np.dtype(
    {
        "names": ["intensity", "a", "b", "c", "type"],  # rect: position/normal/dims, tri: v1/v2/v3
        "formats": [("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), "u1"],
        "offsets": [0, 12, 24, 36, 48],
        "itemsize": 52,
    }
)
SCENE_PARAMS = np.dtype(
    {
        "names": ["coord_scale", "coord_dims", "eye_pos", "frame_x", "frame_y", "frame_z",
                  "shading_diffuse", "shading_metallic", "shading_roughness", "path_tracing",
                  "num_samples", "exposure", "anti_aliasing"],
        "formats": [("<f4", 2), ("<f4", 2), ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3),
                    ("<f4", 3), "<f4", "<f4", "i1", "<i4", "<f4", "i1"],
        "offsets": [0, 8, 16, 28, 40, 52, 64, 76, 80, 84, 88, 92, 96],
        "itemsize": 104,
    }
)
RECT_LIGHT = np.dtype([("intensity", "<f4", 3), ("position", "<f4", 3), ("normal", "<f4", 3),
                       ("dims", "<f4", 2)])
GROUND_PLANE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("dims", "<f4", 2),
                         ("diffuse", "<f4", 3), ("metallic", "<f4"), ("roughness", "<f4")])

RAY_EPSILON = np.float32(1e-2)  # kernels/constants.hpp:10
FLT_MAX = np.float32(3.4028234663852886e38)


def _load_scene_lib() -> ctypes.CDLL:
    path = os.path.join(_HERE, "libnova_scene.so")
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    lib = ctypes.CDLL(path)
    lib.nova_scene_build.restype = ctypes.c_void_p
    lib.nova_scene_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32,
                                     ctypes.c_void_p]
    lib.nova_scene_sizes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.nova_scene_copy.argtypes = [ctypes.c_void_p] * 6
    lib.nova_scene_free.argtypes = [ctypes.c_void_p]
    lib.nova_scene_struct_sizes.argtypes = [ctypes.c_void_p]
    sizes = np.zeros(6, np.int32)
    lib.nova_scene_struct_sizes(sizes.ctypes.data)
    expect = [BVH_NODE.itemsize, TRIANGLE_DATA.itemsize, TRIANGLE_META.itemsize,
              AREA_LIGHT.itemsize, SCENE_PARAMS.itemsize, 64]
    if list(sizes) != expect:
        raise RuntimeError(f"struct layout mismatch: C {list(sizes)} vs numpy {expect}")
    return lib


_lib: Optional[ctypes.CDLL] = None


def scene_lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = _load_scene_lib()
    return _lib


# ------------------------------------------------------------------------------------------------
# Camera (src/camera/camera.cpp:89-105), evaluated in float32 like glm does
# ------------------------------------------------------------------------------------------------
def _normalize32(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.float32)
    return (v * (np.float32(1.0) / np.sqrt(np.dot(v, v), dtype=np.float32))).astype(np.float32)


@dataclasses.dataclass
class Camera:
    position: Sequence[float]
    target: Sequence[float]
    up: Sequence[float]
    fovy: float
    width: int
    height: int

    def eye_coords(self):
        w_, h_ = np.float32(self.width), np.float32(self.height)
        fovy = np.float32(self.fovy)
        half_fov = np.array([fovy * w_ / h_, fovy], np.float32) * np.float32(0.5)
        coord_dims = np.array([w_, h_], np.float32) * np.float32(0.5)
        coord_scale = (np.tan(np.deg2rad(half_fov), dtype=np.float32) / coord_dims).astype(np.float32)
        pos = np.asarray(self.position, np.float32)
        w = -_normalize32(np.asarray(self.target, np.float32) - pos)
        u = _normalize32(np.cross(np.asarray(self.up, np.float32), w).astype(np.float32))
        v = np.cross(w, u).astype(np.float32)
        return coord_scale, coord_dims, pos, (u, v, w)


def make_scene_params(camera: Camera, shading_diffuse=(1.0, 1.0, 1.0), shading_metallic=0.5,
                      shading_roughness=0.1, path_tracing=True, num_samples=1, exposure=1.0,
                      anti_aliasing=False) -> np.ndarray:
    """SceneParams as Raytracer::set_scene fills it (src/core/raytracer.cpp:53-65)."""
    p = np.zeros((), SCENE_PARAMS)
    cs, cd, pos, (u, v, w) = camera.eye_coords()
    p["coord_scale"], p["coord_dims"], p["eye_pos"] = cs, cd, pos
    p["frame_x"], p["frame_y"], p["frame_z"] = u, v, w
    p["shading_diffuse"] = shading_diffuse
    p["shading_metallic"] = shading_metallic
    p["shading_roughness"] = shading_roughness
    p["path_tracing"] = 1 if path_tracing else 0
    p["num_samples"] = num_samples
    p["exposure"] = exposure
    p["anti_aliasing"] = 1 if anti_aliasing else 0
    return p


# ------------------------------------------------------------------------------------------------
# Built scene
# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Scene:
    bvh: np.ndarray          # BVH_NODE[n_nodes]
    triangles: np.ndarray    # TRIANGLE_DATA[n]
    meta: np.ndarray         # TRIANGLE_META[n]
    lights: np.ndarray       # AREA_LIGHT[n_lights]
    tri_order: np.ndarray    # uint32[n]: output triangle i = input triangle tri_order[i]
    verts: np.ndarray        # float32[n, 9] in OUTPUT (BVH leaf) order
    max_depth: int
    materials: np.ndarray    # float32[layers, h, w, 4]
    sky: np.ndarray          # float32[h, w, 4]

    @property
    def num_lights(self) -> int:
        return int(self.lights.shape[0])


def default_meta(n: int, kD=(1.0, 1.0, 1.0)) -> np.ndarray:
    m = np.zeros(n, TRIANGLE_META)
    m["kD"] = kD
    m["metallic"] = -1.0
    m["roughness"] = -1.0
    for k in ("diffuse_index", "metallic_index", "roughness_index", "normal_index", "light_index"):
        m[k] = -1
    return m


def face_normals(verts: np.ndarray) -> np.ndarray:
    v = verts.reshape(-1, 3, 3).astype(np.float32)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]).astype(np.float32)
    ln = np.sqrt((n * n).sum(-1, dtype=np.float32), dtype=np.float32)
    ln[ln == 0] = 1
    return (n / ln[:, None]).astype(np.float32)


def constant_sky(rgb=(0.5, 0.7, 1.0), w=8, h=4) -> np.ndarray:
    s = np.empty((h, w, 4), np.float32)
    s[..., :3] = rgb
    s[..., 3] = 1.0
    return s


def analytic_sky(w=512, h=256, seed=6) -> np.ndarray:
    """Seeded sun + gradient equirect HDR: stand-in for assets/sky/lakeside_4k.hdr (absent)."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    elev = (v - 0.5) * np.pi
    azim = (u - 0.5) * 2 * np.pi
    d = np.stack([np.cos(elev) * np.cos(azim), np.sin(elev), np.cos(elev) * np.sin(azim)], -1)
    sun = np.array([0.3, 0.8, -0.52])
    sun /= np.linalg.norm(sun)
    c = np.clip((d * sun).sum(-1), 0, 1)
    up = np.clip(d[..., 1], 0, 1)[..., None]
    sky = (1 - up) * np.array([0.9, 0.85, 0.8]) + up * np.array([0.25, 0.45, 0.9])
    ground = np.array([0.25, 0.22, 0.2])
    col = np.where(d[..., 1:2] > 0, sky, ground)
    col = col + (c ** 600)[..., None] * 400.0 + (c ** 8)[..., None] * 0.6
    col *= 1.0 + 0.02 * rng.standard_normal((h, w, 1))
    out = np.empty((h, w, 4), np.float32)
    out[..., :3] = np.clip(col, 0, None)
    out[..., 3] = 1.0
    return out


def build_scene(verts: np.ndarray, meta: Optional[np.ndarray] = None,
                emissive: Optional[np.ndarray] = None,
                rect_lights: Optional[np.ndarray] = None,
                ground: Optional[np.ndarray] = None,
                materials: Optional[np.ndarray] = None,
                sky: Optional[np.ndarray] = None) -> Scene:
    """IntersectableManager::build (src/intersectables/intersectable_manager.cpp:102-186)."""
    lib = scene_lib()
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    n = verts.shape[0]
    if meta is None:
        meta = default_meta(n)
        fn = face_normals(verts)
        meta["normal1"] = meta["normal2"] = meta["normal3"] = fn
    meta = np.ascontiguousarray(meta)
    assert meta.dtype == TRIANGLE_META and meta.shape[0] == n
    em = None if emissive is None else np.ascontiguousarray(emissive, np.float32).reshape(n, 3)
    rl = None if rect_lights is None else np.ascontiguousarray(rect_lights).view(RECT_LIGHT)
    gp = None if ground is None else np.ascontiguousarray(ground).view(GROUND_PLANE)
    h = lib.nova_scene_build(
        verts.ctypes.data, meta.ctypes.data, None if em is None else em.ctypes.data, n,
        None if rl is None else rl.ctypes.data, 0 if rl is None else rl.shape[0],
        None if gp is None else gp.ctypes.data)
    if not h:
        raise ValueError("empty scene")
    try:
        sizes = np.zeros(4, np.uint64)
        lib.nova_scene_sizes(h, sizes.ctypes.data)
        n_nodes, n_tris, n_lights, depth = (int(x) for x in sizes)
        bvh = np.zeros(n_nodes, BVH_NODE)
        tris = np.zeros(n_tris, TRIANGLE_DATA)
        out_meta = np.zeros(n_tris, TRIANGLE_META)
        lights = np.zeros(n_lights, AREA_LIGHT)
        order = np.zeros(n_tris, np.uint32)
        lib.nova_scene_copy(h, bvh.ctypes.data, tris.ctypes.data, out_meta.ctypes.data,
                            lights.ctypes.data if n_lights else None, order.ctypes.data)
    finally:
        lib.nova_scene_free(h)
    if materials is None:
        materials = np.zeros((1, 1, 1, 4), np.float32)  # dummy array (raytracer.cpp:135-142)
    if sky is None:
        sky = constant_sky()
    # vertices in output order; triangles appended by the builder (lights, ground) are not in
    # `verts`, so mark them with NaN
    all_verts = np.full((n_tris, 9), np.nan, np.float32)
    in_model = order < n
    all_verts[in_model] = verts[order[in_model]]
    return Scene(bvh, tris, out_meta, lights, order, all_verts, depth,
                 np.ascontiguousarray(materials, np.float32), np.ascontiguousarray(sky, np.float32))


# ------------------------------------------------------------------------------------------------
# OBJ / MTL
# ------------------------------------------------------------------------------------------------
def _parse_mtl(path: str) -> dict:
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split("#")[0].split()
            if not t:
                continue
            if t[0] == "newmtl":
                cur = {"Kd": (0.6, 0.6, 0.6), "Ke": (0.0, 0.0, 0.0)}
                mats[" ".join(t[1:])] = cur
            elif cur is not None and t[0] in ("Kd", "Ke") and len(t) >= 4:
                cur[t[0]] = tuple(float(x) for x in t[1:4])
            elif cur is not None and t[0].startswith("map_"):
                cur[t[0]] = " ".join(t[1:])
    return mats


def read_obj_indexed(path: str):
    """The data of an OBJ file as arrays: positions, normals, texture coordinates, per-corner indices of the
    fan-triangulated faces (-1 = absent), material number per triangle, material names, MTL dictionary."""
    pos, nrm, tex = [], [], []
    tri_v, tri_t, tri_n, tri_m = [], [], [], []
    names, mats = [], {}
    cur = -1
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split("#")[0].split()
            if not t:
                continue
            if t[0] == "v":
                pos.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                nrm.append([float(x) for x in t[1:4]])
            elif t[0] == "vt":
                tex.append([float(t[1]), float(t[2]) if len(t) > 2 else 0.0])
            elif t[0] == "mtllib":
                mats.update(_parse_mtl(os.path.join(os.path.dirname(path), " ".join(t[1:]))))
            elif t[0] == "usemtl":
                name = " ".join(t[1:])
                if name not in names:
                    names.append(name)
                cur = names.index(name)
            elif t[0] == "f":
                idx = []
                for w in t[1:]:
                    p = (w.split("/") + ["", ""])[:3]
                    vi = int(p[0])
                    ti = int(p[1]) if p[1] else 0
                    ni = int(p[2]) if p[2] else 0
                    idx.append((vi + len(pos) if vi < 0 else vi - 1,
                                (ti + len(tex) if ti < 0 else ti - 1) if ti else -1,
                                (ni + len(nrm) if ni < 0 else ni - 1) if ni else -1))
                for k in range(1, len(idx) - 1):
                    tri = (idx[0], idx[k], idx[k + 1])
                    tri_v.append([c[0] for c in tri])
                    tri_t.append([c[1] for c in tri])
                    tri_n.append([c[2] for c in tri])
                    tri_m.append(cur)
    return dict(pos=np.asarray(pos, np.float32).reshape(-1, 3), nrm=np.asarray(nrm, np.float32).reshape(-1, 3),
                tex=np.asarray(tex, np.float32).reshape(-1, 2), tri_v=np.asarray(tri_v, np.int32).reshape(-1, 3),
                tri_t=np.asarray(tri_t, np.int32).reshape(-1, 3), tri_n=np.asarray(tri_n, np.int32).reshape(-1, 3),
                tri_m=np.asarray(tri_m, np.int32)), names, mats


def assemble_mesh(d, tangents=False):
    """Indexed OBJ data -> (verts[n,9], meta[n], kept): geometry normalised to [-1,1] like assimp's
    AI_CONFIG_PP_PTV_NORMALIZE (src/model/model.cpp:22-28); missing normals = face normals; degenerate
    triangles dropped (model.cpp:128-134).  tangents=True adds a per-face tangent frame from the texture
    coordinates, the bitangent fixed as model.cpp:151-163 does (cross(n, t), sign of the computed one);
    assimp's CalcTangentSpace smooths per vertex instead, so this is same-shaped input, not assimp's output.
    `kept` = indices into the input triangles."""
    P = d["pos"]
    lo, hi = P.min(0), P.max(0)
    ext = hi - lo
    P = ((P - (lo + ext * np.float32(0.5))) / np.float32(max(ext) * 0.5)).astype(np.float32)
    v = P[d["tri_v"]]  # [n, 3, 3]
    e01 = np.linalg.norm(v[:, 0] - v[:, 1], axis=1)
    e12 = np.linalg.norm(v[:, 1] - v[:, 2], axis=1)
    e02 = np.linalg.norm(v[:, 0] - v[:, 2], axis=1)
    L = np.sort(np.stack([e01, e12, e02], 1), axis=1)
    kept = np.flatnonzero(~(L[:, 0] + L[:, 1] <= L[:, 2]))
    v = v[kept]
    verts = np.ascontiguousarray(v.reshape(-1, 9), np.float32)
    n = verts.shape[0]
    meta = default_meta(n)
    fn = face_normals(verts)
    tn = d["tri_n"][kept]
    N = d["nrm"] if d["nrm"].shape[0] else np.zeros((1, 3), np.float32)
    ns = np.where((tn >= 0).all(1)[:, None, None], N[np.maximum(tn, 0)], fn[:, None, :]).astype(np.float32)
    ns = (ns / np.maximum(np.linalg.norm(ns, axis=2, keepdims=True), 1e-30)).astype(np.float32)
    meta["normal1"], meta["normal2"], meta["normal3"] = ns[:, 0], ns[:, 1], ns[:, 2]
    tt = d["tri_t"][kept]
    T = d["tex"] if d["tex"].shape[0] else np.zeros((1, 2), np.float32)
    uv = np.where((tt >= 0)[:, :, None], T[np.maximum(tt, 0)], 0).astype(np.float32)
    meta["texture_coord1"], meta["texture_coord2"], meta["texture_coord3"] = uv[:, 0], uv[:, 1], uv[:, 2]
    if tangents:
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        d1, d2 = uv[:, 1] - uv[:, 0], uv[:, 2] - uv[:, 0]
        det = (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1]).astype(np.float32)
        det = np.where(det == 0, np.float32(1.0), det)
        tan = ((e1 * d2[:, 1:2] - e2 * d1[:, 1:2]) / det[:, None]).astype(np.float32)
        bit = ((e2 * d1[:, 0:1] - e1 * d2[:, 0:1]) / det[:, None]).astype(np.float32)
        tan = (tan / np.maximum(np.linalg.norm(tan, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        for k, nk in enumerate(("1", "2", "3")):
            fb = np.cross(ns[:, k], tan).astype(np.float32)
            fb = np.where(((fb * bit).sum(1) < 0)[:, None], -fb, fb).astype(np.float32)
            meta["tangent" + nk], meta["bitangent" + nk] = tan, fb
    return verts, meta, kept


def load_obj(path: str):
    """Returns (verts[n,9], meta[n], emissive[n,3], [material name per triangle, MTL dict]); see
    read_obj_indexed / assemble_mesh."""
    d, names, mats = read_obj_indexed(path)
    verts, meta, kept = assemble_mesh(d)
    n = verts.shape[0]
    em = np.zeros((n, 3), np.float32)
    tm = [names[m] if m >= 0 else None for m in d["tri_m"][kept]]
    for i in range(n):
        m = mats.get(tm[i], {"Kd": (0.6, 0.6, 0.6), "Ke": (0, 0, 0)})
        meta["kD"][i] = m["Kd"]
        em[i] = m["Ke"]
    return verts, meta, em, [tm, mats]


def aircraft_layers_u8() -> np.ndarray:
    """The aircraft's material layers as uint8 [layers, h, w, 4] (what MaterialLoader holds before x / 255)."""
    return np.load(os.path.join(_HERE, "..", "tests", "golden", "aircraft.npz"))["layers"]


def aircraft_scene(sky_size=(1024, 512)) -> Scene:
    """BASELINE config 5 on its own asset, from the committed data fixture tests/golden/aircraft.npz
    (tests/golden/make_aircraft_fixture.py: assets/aircraft/aircraft.obj + the two JPEGs its MTL uses that
    are in the reference tree; the third, `E-45 _col.jpg`, is absent — .MISSING_LARGE_BLOBS — and replaced
    by a seeded procedural albedo).  The layers are texels the REFERENCE'S OWN MaterialLoader made (compiled from
    its files, oracle/_ref/libnova_ref_material.so: stb decode, flip, sRGB -> linear on uint8, stbir resize to the
    mean size, material_loader.cpp:11-80), in the order Model::process_mesh loads them (src/model/model.cpp:76-89):
    a 256 x 256 window of each 1792 x 1792 layer (the whole array is 205 MB; its SHA-256 is in the fixture);
    uint8 -> float4 / 255 (material_loader.cpp:69-76).  Sky: sky_size = None -> the 128 x 64 texels the reference's
    SkyLoader returned for a Radiance file of the analytic stand-in (lakeside_4k.hdr is absent); else the analytic
    stand-in itself at that size."""
    z = np.load(os.path.join(_HERE, "..", "tests", "golden", "aircraft.npz"))
    d = {k: z[k] for k in ("pos", "nrm", "tex", "tri_v", "tri_t", "tri_n", "tri_m")}
    verts, meta, kept = assemble_mesh(d, tangents=True)
    mat = d["tri_m"][kept]
    meta["kD"] = (1.0, 1.0, 1.0)  # both materials: Kd 1 1 1
    meta["diffuse_index"] = z["diffuse_layer"][mat]
    meta["normal_index"] = z["normal_layer"][mat]
    layers = (z["layers"].astype(np.float32) / np.float32(255.0)).astype(np.float32)
    sky = np.ascontiguousarray(z["sky"], np.float32) if sky_size is None else analytic_sky(*sky_size)
    return build_scene(verts, meta=meta, materials=layers, sky=sky)


AIRCRAFT_CAMERA = dict(position=[-1.6, 0.9, 1.9], target=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fovy=40.0)


# ------------------------------------------------------------------------------------------------
# Procedural stand-ins (assets absent from the reference tree, .MISSING_LARGE_BLOBS)
# ------------------------------------------------------------------------------------------------
def grid_to_triangles(P: np.ndarray, wrap_u=True, wrap_v=True) -> np.ndarray:
    """P[nu, nv, 3] -> 2*nu*nv triangles (periodic in both directions by default)."""
    nu, nv, _ = P.shape
    iu = np.arange(nu if wrap_u else nu - 1)
    iv = np.arange(nv if wrap_v else nv - 1)
    a, b = np.meshgrid(iu, iv, indexing="ij")
    a2, b2 = (a + 1) % nu, (b + 1) % nv
    p00, p10, p01, p11 = P[a, b], P[a2, b], P[a, b2], P[a2, b2]
    t1 = np.stack([p00, p11, p10], -2)
    t2 = np.stack([p00, p01, p11], -2)
    return np.concatenate([t1.reshape(-1, 9), t2.reshape(-1, 9)]).astype(np.float32)


def knot_mesh(nu=660, nv=660, seed=2) -> np.ndarray:
    """Dragon stand-in: a displaced (2,3) torus-knot tube, 2*nu*nv triangles (871,200 by default,
    the Stanford-dragon class SURVEY.md §8 assumes), normalised to [-1,1]."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    p, q = 2, 3
    r = 0.5 * (2 + np.cos(q * u))
    c = np.stack([r * np.cos(p * u), 0.9 * np.sin(q * u), r * np.sin(p * u)], -1)
    t = np.gradient(c, axis=0)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    ref = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(t, ref)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(t, n1)
    U, V = np.meshgrid(u, v, indexing="ij")
    rad = 0.28 * np.ones_like(U)
    for _ in range(12):
        fu, fv = rng.integers(1, 40), rng.integers(1, 12)
        ph = rng.uniform(0, 2 * np.pi, 2)
        rad += 0.05 / np.sqrt(fu) * np.sin(fu * U + ph[0]) * np.sin(fv * V + ph[1])
    P = c[:, None, :] + rad[..., None] * (np.cos(V)[..., None] * n1[:, None, :] +
                                          np.sin(V)[..., None] * n2[:, None, :])
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    P = (P - (lo + hi) / 2) / (max(hi - lo) / 2)
    return grid_to_triangles(P.astype(np.float32))


def soup(n: int, seed=4, edge=0.01) -> np.ndarray:
    """BASELINE config 4: centres uniform in [-1,1]^3, edge vectors uniform in [-edge,edge]^3."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
    e = rng.uniform(-edge, edge, (n, 3, 3)).astype(np.float32)
    return (c + e).reshape(n, 9).astype(np.float32)


def bumpy_sphere(center, radius, nu, nv, rng) -> np.ndarray:
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = (np.arange(nv) + 0.5) / nv * np.pi
    U, V = np.meshgrid(u, v, indexing="ij")
    r = radius * (1 + 0.15 * np.sin(rng.integers(2, 7) * U + rng.uniform(0, 6)) *
                  np.sin(rng.integers(2, 7) * V + rng.uniform(0, 6)))
    P = np.stack([r * np.sin(V) * np.cos(U), r * np.cos(V), r * np.sin(V) * np.sin(U)], -1)
    return grid_to_triangles((P + np.asarray(center)).astype(np.float32), wrap_u=True, wrap_v=False)


def room_scene(n_objects=140, res=23, seed=3):
    """fireplace_room stand-in: closed box room, seeded clutter (~143 k triangles by default) and
    one emissive ceiling quad.  Returns (verts, meta, emissive)."""
    rng = np.random.default_rng(seed)
    parts, kds = [], []

    def quad(a, b, c, d, n=8):
        s = (np.arange(n + 1) / n).astype(np.float32)
        S, T = np.meshgrid(s, s, indexing="ij")
        a, b, c, d = (np.asarray(x, np.float32) for x in (a, b, c, d))
        P = (a * ((1 - S) * (1 - T))[..., None] + b * (S * (1 - T))[..., None] +
             c * (S * T)[..., None] + d * ((1 - S) * T)[..., None])
        return grid_to_triangles(P, wrap_u=False, wrap_v=False)

    def facing(tris, toward=None, away_from=None):
        """Winding such that the face normal points toward a point (room surfaces: the room's centre;
        the reference shades with the stored normals as they are, kernels/material.hpp:37-41, so a wall
        wound outwards would never receive light) or away from one (objects: their centre)."""
        v = tris.reshape(-1, 3, 3)
        nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        cen = v.mean(1)
        ref = (np.asarray(toward, np.float32) - cen) if toward is not None else (cen - np.asarray(away_from, np.float32))
        flip = (nrm * ref).sum(1) < 0
        v = v.copy()
        v[flip] = v[flip][:, [0, 2, 1]]
        return np.ascontiguousarray(v.reshape(-1, 9), np.float32)

    X, Y, Z = 1.0, 0.6, 1.0
    walls = [
        ([-X, -Y, -Z], [X, -Y, -Z], [X, -Y, Z], [-X, -Y, Z], (0.55, 0.5, 0.45)),   # floor
        ([-X, Y, -Z], [-X, Y, Z], [X, Y, Z], [X, Y, -Z], (0.8, 0.8, 0.8)),        # ceiling
        ([-X, -Y, -Z], [-X, Y, -Z], [X, Y, -Z], [X, -Y, -Z], (0.7, 0.7, 0.65)),   # back
        ([-X, -Y, Z], [X, -Y, Z], [X, Y, Z], [-X, Y, Z], (0.7, 0.7, 0.65)),       # front
        ([-X, -Y, -Z], [-X, -Y, Z], [-X, Y, Z], [-X, Y, -Z], (0.6, 0.2, 0.15)),   # left
        ([X, -Y, -Z], [X, Y, -Z], [X, Y, Z], [X, -Y, Z], (0.2, 0.45, 0.2)),       # right
    ]
    for a, b, c, d, kd in walls:
        q = facing(quad(a, b, c, d), toward=[0.0, 0.0, 0.0])
        parts.append(q)
        kds.append(np.tile(np.asarray(kd, np.float32), (q.shape[0], 1)))
    for _ in range(n_objects):
        r = rng.uniform(0.03, 0.12)
        cpos = [rng.uniform(-X + r, X - r), rng.uniform(-Y + r, 0.1), rng.uniform(-Z + r, Z - r)]
        m = facing(bumpy_sphere(cpos, r, res, res, rng), away_from=cpos)
        parts.append(m)
        kds.append(np.tile(rng.uniform(0.2, 0.9, 3).astype(np.float32), (m.shape[0], 1)))
    light = facing(quad([-0.3, Y - 1e-3, -0.3], [0.3, Y - 1e-3, -0.3], [0.3, Y - 1e-3, 0.3],
                        [-0.3, Y - 1e-3, 0.3], n=1), toward=[0.0, 0.0, 0.0])
    parts.append(light)
    kds.append(np.tile(np.float32([0.78, 0.78, 0.78]), (light.shape[0], 1)))
    verts = np.concatenate(parts)
    kd = np.concatenate(kds)
    n = verts.shape[0]
    meta = default_meta(n)
    fn = face_normals(verts)
    meta["normal1"] = meta["normal2"] = meta["normal3"] = fn
    meta["kD"] = kd
    em = np.zeros((n, 3), np.float32)
    em[n - light.shape[0]:] = (40.0, 36.0, 30.0)
    return verts, meta, em


def encode_ids_in_kd(meta: np.ndarray) -> np.ndarray:
    """Parity-test helper: make kernel_raytrace report the primary closest-hit triangle index.

    With path_tracing = 0 the kernel returns `diffuse` for the first hit (kernels/raytrace.hpp:170-177)
    and diffuse = shading_diffuse * meta.kD when no texture is bound (:160-161, material.hpp:20-26),
    so kD = (id & 0xFFF, id >> 12, 1) with shading_diffuse = 1 makes the colour image carry the
    triangle index exactly.  Misses read the sky (use a constant negative sky to tell them apart)."""
    m = meta.copy()
    ids = np.arange(m.shape[0], dtype=np.int64)
    m["kD"][:, 0] = (ids & 0xFFF).astype(np.float32)
    m["kD"][:, 1] = (ids >> 12).astype(np.float32)
    m["kD"][:, 2] = 1.0
    for k in ("diffuse_index", "metallic_index", "roughness_index", "normal_index", "light_index"):
        m[k] = -1
    return m


def decode_ids(color: np.ndarray) -> np.ndarray:
    """Inverse of encode_ids_in_kd on a float4 colour image; misses (negative sky) -> -1."""
    c = np.asarray(color, np.float32).reshape(-1, 4)
    ids = c[:, 0].astype(np.int64) + (c[:, 1].astype(np.int64) << 12)
    ids[c[:, 2] != 1.0] = -1
    return ids.astype(np.int32)


def cornell_box() -> Scene:
    """BASELINE config 1 scene, from the committed data fixture tests/golden/cornell_box.npz
    (generated from the reference's public-domain Cornell box OBJ by make_cornell_fixture.py)."""
    path = os.path.join(_HERE, "..", "tests", "golden", "cornell_box.npz")
    z = np.load(path)
    meta = np.ascontiguousarray(z["meta"]).view(TRIANGLE_META).reshape(-1)
    return build_scene(z["verts"], meta, z["emissive"])


CORNELL_CAMERA = dict(position=[0.0, 0.0, 5.0], target=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fovy=30.0)
DRAGON_CAMERA = dict(position=[-1.7, 1.1, -0.8], target=[0.0, -0.1, 0.0], up=[0.0, 1.0, 0.0], fovy=45.0)


def dragon_standin(nu=660, nv=660) -> Scene:
    """BASELINE config 2 stand-in (assets/dragon/dragon.glb is absent): the displaced torus knot
    with the light and ground plane of scene.toml's dragon section (scene.toml:47-74)."""
    verts = knot_mesh(nu, nv)
    light = np.zeros(1, RECT_LIGHT)
    light["intensity"], light["position"] = (500.0, 500.0, 500.0), (-1.4, 1.7, -2.0)
    light["normal"], light["dims"] = (0.0, -1.0, 0.0), (0.5, 0.5)
    ground = np.zeros(1, GROUND_PLANE)
    ground["position"], ground["normal"], ground["dims"] = (0.0, -0.8, 0.0), (0.0, 1.0, 0.0), (5.0, 5.0)
    ground["diffuse"], ground["metallic"], ground["roughness"] = (0.2, 0.2, 0.2), 0.01, 0.01
    return build_scene(verts, rect_lights=light, ground=ground, sky=analytic_sky())


ROOM_CAMERA = dict(position=[0.85, 0.05, 0.9], target=[-0.3, -0.25, -0.4], up=[0.0, 1.0, 0.0], fovy=55.0)
ROOM_SHADING = dict(shading_metallic=0.1, shading_roughness=0.6)


def room_standin(n_objects=140, res=23, seed=3) -> Scene:
    """BASELINE config 3 stand-in (assets/fireplace_room/fireplace_room.glb is absent): a closed room
    with seeded clutter (~142 k triangles) lit by one emissive ceiling quad; see room_scene."""
    verts, meta, em = room_scene(n_objects, res, seed)
    return build_scene(verts, meta=meta, emissive=em)


# ------------------------------------------------------------------------------------------------
# The reference's hex dump format for FlatBVHNode / Triangle (src/bvh/bvh.cpp:251-281,
# src/intersectables/triangle.cpp:27-55): the float bit patterns as hex words separated by blanks.
# ------------------------------------------------------------------------------------------------
def to_hex_words(arr: np.ndarray) -> str:
    words = np.ascontiguousarray(arr).view(np.uint32).reshape(-1)
    return "".join(f"{w:x} " for w in words)


def from_hex_words(text: str, dtype) -> np.ndarray:
    words = np.array([int(t, 16) for t in text.split()], dtype=np.uint32)
    return words.view(np.dtype(dtype))
