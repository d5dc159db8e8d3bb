"""Writes tests/golden/aircraft.npz (build container only: reads /root/reference as DATA).

BASELINE config 5's own asset: assets/aircraft/aircraft.obj (8110 faces -> 16220 triangles), its MTL and
the JPEGs the MTL's diffuse / bump slots name that exist in the reference tree, decoded with PIL:
  * geometry as indexed arrays (positions, normals, texture coordinates, per-corner indices, material id);
  * texture layers in the order Model::process_mesh loads them (src/model/model.cpp:76-89: per material the
    DIFFUSE map, then — OBJ bump maps arrive as HEIGHT — the normal map), each flipped vertically
    (stbi_set_flip_vertically_on_load(true), material_loader.cpp:11), RGBA uint8, the diffuse ones passed
    through MaterialLoader's sRGB -> linear rule ON uint8 (material_loader.cpp:24-33:
    uint8(pow(x / 255, 2.2) * 255), float arithmetic, truncating);
  * MaterialLoader::build resizes every layer to the mean width x height (material_loader.cpp:46-66; here
    (1024 + 1024 + 2048 + 3072) / 4 = 1792).  The fixture keeps layers at 256 x 256, i.e. that size divided by
    7, to stay small, and resizes with PIL's LANCZOS filter, not stb_image_resize: these are same-shaped
    inputs for the shading path, not the bytes the reference would upload;
  * `E-45 _col.jpg` (the ship's diffuse map) is absent from the reference tree (.MISSING_LARGE_BLOBS:1):
    replaced by a seeded 2048^2 procedural albedo (panels + noise), as SURVEY.md §8d prescribes.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nova_amd import scene as S  # noqa: E402

REF = "/root/reference/assets/aircraft"
SIZE = 256


def srgb_to_linear_u8(a):
    x = a.astype(np.float32) / np.float32(255.0)
    return (np.power(x, np.float32(2.2), dtype=np.float32) * np.float32(255.0)).astype(np.uint8)


def load(path, srgb):
    im = Image.open(path).convert("RGBA")
    a = np.asarray(im)[::-1].copy()  # stb flips vertically on load
    if srgb:
        a[..., :3] = srgb_to_linear_u8(a[..., :3])
    return a


def procedural_albedo(n=2048, seed=45):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n]
    panels = (((x // 128) + (y // 96)) % 2).astype(np.float32)
    base = np.stack([0.55 + 0.2 * panels, 0.58 + 0.15 * panels, 0.62 - 0.1 * panels], -1)
    base += 0.08 * rng.standard_normal((n // 8, n // 8, 3)).repeat(8, 0).repeat(8, 1)
    rgb = (np.clip(base, 0, 1) * 255).astype(np.uint8)
    return np.concatenate([rgb, np.full((n, n, 1), 255, np.uint8)], -1)


def resize(a):
    return np.asarray(Image.fromarray(a, "RGBA").resize((SIZE, SIZE), Image.LANCZOS))


def main():
    d, names, mats = S.read_obj_indexed(os.path.join(REF, "aircraft.obj"))
    layers, diffuse_layer, normal_layer = [], [], []
    for name in names:
        m = mats[name]
        for slot, srgb, out in (("map_Kd", True, diffuse_layer), ("map_Bump", False, normal_layer)):
            rel = m.get(slot)
            if rel is None:
                out.append(-1)
                continue
            path = os.path.join(REF, rel)
            if os.path.exists(path):
                a = load(path, srgb)
            else:
                assert rel.endswith("E-45 _col.jpg"), rel
                a = procedural_albedo()
                a[..., :3] = srgb_to_linear_u8(a[..., :3])
            out.append(len(layers))
            layers.append(resize(a))
    out = os.path.join(ROOT, "tests", "golden", "aircraft.npz")
    np.savez_compressed(out, layers=np.stack(layers), diffuse_layer=np.asarray(diffuse_layer, np.int32),
                        normal_layer=np.asarray(normal_layer, np.int32), material_names=np.asarray(names), **d)
    print(out, os.path.getsize(out), "bytes;", d["tri_v"].shape[0], "triangles;", len(layers), "layers;", names)


if __name__ == "__main__":
    main()
