"""Writes tests/golden/aircraft.npz (build container only: reads /root/reference as DATA and runs the reference's
OWN texture loaders, compiled from their files by oracle/Makefile into oracle/_ref/libnova_ref_material.so).

BASELINE config 5's own asset: assets/aircraft/aircraft.obj (8110 faces -> 16220 triangles), its MTL and the JPEGs the
MTL's diffuse / bump slots name that exist in the reference tree:
  * geometry as indexed arrays (positions, normals, texture coordinates, per-corner indices, material id);
  * texture layers: nova::MaterialLoader itself.  load_material(path, srgb) is called per texture in the order
    Model::process_mesh asks for them (src/model/model.cpp:76-89: per material the DIFFUSE map, srgb = true, then —
    OBJ bump maps arrive as HEIGHT — the normal map), then build() (src/material/material_loader.cpp:39-80): stb
    decode, vertical flip, sRGB -> linear on uint8, stbir_resize_uint8 to the mean size ((1024 + 1024 + 2048 + 3072)
    / 4 = 1792 squared), uint8 / 255.0f.  That array is 4 x 1792^2 float4 = 205 MB, so the fixture keeps
      - its SHA-256 and dimensions (`full_sha256`, `full_dims`),
      - a 256 x 256 WINDOW of every layer as uint8 (`layers`, origin `window_origin`; the float4 values are exactly
        layers / 255.0f — checked here for the whole array — and this is what the C5 tests upload: the reference
        loader's own texels, a window of them instead of PIL + LANCZOS stand-ins),
      - and a 64 x 64 float4 window (`window_f4`) for the bit test u8 / 255.0f == float4;
  * `E-45 _col.jpg` (the ship's diffuse map) is absent from the reference tree (.MISSING_LARGE_BLOBS:1): replaced by
    a seeded 2048^2 procedural albedo (panels + noise) written as a PNG for the loader to read, as SURVEY.md §8d
    prescribes;
  * sky: nova::SkyLoader itself (src/material/sky_loader.cpp:12-35, srgb = true as Raytracer::set_scene calls it,
    src/core/raytracer.cpp:152) on a 128 x 64 Radiance .hdr of the analytic stand-in sky (lakeside_4k.hdr is absent):
    `sky` = the float4 texels it returns (RGBE decode, flip, pow 2.2).
"""
import ctypes
import hashlib
import os
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nova_amd import scene as S  # noqa: E402

REF = "/root/reference/assets/aircraft"
LIB = os.path.join(ROOT, "oracle", "_ref", "libnova_ref_material.so")
WINDOW = 256
ORIGIN = (640, 704)  # x0, y0 of the window inside the 1792^2 layers: hull plates, rivets and the glass normal map's grain


def procedural_albedo(n=2048, seed=45):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n]
    panels = (((x // 128) + (y // 96)) % 2).astype(np.float32)
    base = np.stack([0.55 + 0.2 * panels, 0.58 + 0.15 * panels, 0.62 - 0.1 * panels], -1)
    base += 0.08 * rng.standard_normal((n // 8, n // 8, 3)).repeat(8, 0).repeat(8, 1)
    rgb = (np.clip(base, 0, 1) * 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((n, n, 1), 255, np.uint8)], -1)


def write_rgbe(path, img):
    """A flat (not run-length encoded) Radiance .hdr of float RGB [h, w, 3], top row first."""
    h, w, _ = img.shape
    m = img.max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0)
    scale = np.where(m > 1e-32, 256.0 / np.exp2(e), 0.0)
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(rgbe.tobytes())


def reference_layers(paths, srgb):
    lib = ctypes.CDLL(LIB)
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    flags = (ctypes.c_int * len(paths))(*[int(s) for s in srgb])
    dims = (ctypes.c_uint32 * 3)()
    assert lib.ref_material_build(arr, flags, len(paths), dims, None, ctypes.c_uint64(0)) == 0
    w, h, n = dims[0], dims[1], dims[2]
    out = np.zeros((n, h, w, 4), np.float32)
    assert lib.ref_material_build(arr, flags, len(paths), dims, ctypes.c_void_p(out.ctypes.data), ctypes.c_uint64(out.size)) == 0
    return out


def reference_sky(path, srgb=True):
    lib = ctypes.CDLL(LIB)
    dims = (ctypes.c_uint32 * 2)()
    assert lib.ref_sky_load(path.encode(), int(srgb), dims, None, ctypes.c_uint64(0)) == 0
    out = np.zeros((dims[1], dims[0], 4), np.float32)
    assert lib.ref_sky_load(path.encode(), int(srgb), dims, ctypes.c_void_p(out.ctypes.data), ctypes.c_uint64(out.size)) == 0
    return out


def build(tmp):
    d, names, mats = S.read_obj_indexed(os.path.join(REF, "aircraft.obj"))
    paths, srgb, diffuse_layer, normal_layer = [], [], [], []
    for name in names:
        m = mats[name]
        for slot, is_srgb, out in (("map_Kd", True, diffuse_layer), ("map_Bump", False, normal_layer)):
            rel = m.get(slot)
            if rel is None:
                out.append(-1)
                continue
            path = os.path.join(REF, rel)
            if not os.path.exists(path):
                assert rel.endswith("E-45 _col.jpg"), rel
                path = os.path.join(tmp, "E-45_col_standin.png")
                Image.fromarray(procedural_albedo(), "RGBA").save(path)
            out.append(len(paths))
            paths.append(path)
            srgb.append(is_srgb)
    full = reference_layers(paths, srgb)
    u8 = np.rint(full * np.float32(255.0)).astype(np.uint8)
    assert ((u8.astype(np.float32) / np.float32(255.0)).view(np.uint32) == full.view(np.uint32)).all()  # :69-76, every texel
    x0, y0 = ORIGIN
    window = np.ascontiguousarray(u8[:, y0:y0 + WINDOW, x0:x0 + WINDOW])
    window_f4 = np.ascontiguousarray(full[:, y0:y0 + 64, x0:x0 + 64])
    sky_path = os.path.join(tmp, "sky_standin.hdr")
    write_rgbe(sky_path, S.analytic_sky(128, 64)[..., :3].astype(np.float64))
    sky = reference_sky(sky_path)
    return dict(layers=window, window_origin=np.int32([x0, y0]), window_f4=window_f4,
                full_dims=np.int32([full.shape[2], full.shape[1], full.shape[0]]),
                full_sha256=np.asarray(hashlib.sha256(full.tobytes()).hexdigest()),
                layer_sources=np.asarray([os.path.basename(p) for p in paths]), layer_srgb=np.asarray(srgb),
                diffuse_layer=np.asarray(diffuse_layer, np.int32), normal_layer=np.asarray(normal_layer, np.int32),
                material_names=np.asarray(names), sky=sky, **d)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        data = build(tmp)
    out = os.path.join(ROOT, "tests", "golden", "aircraft.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes;", data["tri_v"].shape[0], "triangles;", data["layers"].shape, "window of",
          data["full_dims"], "sha256", str(data["full_sha256"])[:16], "sky", data["sky"].shape, float(data["sky"].max()))


if __name__ == "__main__":
    main()
