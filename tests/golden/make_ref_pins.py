"""Writes tests/golden/ref_pins.npz: outputs of the REFERENCE ITSELF (oracle/_ref/*.so, compiled by oracle/Makefile
from the reference's own files where they lie under /root/reference; build container only) on seeded inputs.
They pin, wherever the tests run (the GPU box has no /root/reference):
  * the Woop matrices (TriangleData) of the CPU scene builder and of the device builder — glm::cross / normalize /
    inverse / transpose as src/intersectables/intersectable_manager.cpp:129-145 calls them;
  * triangle bounds and the hex dump format — nova::Triangle (src/intersectables/triangle.hpp:18-23,
    triangle.cpp:27-55);
  * flatten<float4, float, 3> / pack<float4, float, 3> — src/vector/vector_utils.hpp:8-51, the host half of the
    denoise hand-off that nova_hip_denoise_read / _write do on the device."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from nova_amd import scene as S  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def vp(a):
    return ctypes.c_void_p(a.ctypes.data)


def ref_woop(verts):
    """TriangleData rows (float32 [n, 12]) by the reference's glm calls; the matrix columns are assembled as
    intersectable_manager.cpp:137-141 assembles them (e1, e2, normal - v1, v1)."""
    G = ctypes.CDLL(os.path.join(REF, "libnova_ref_glm.so"))
    v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3, 3)
    n = v.shape[0]
    e1 = np.ascontiguousarray(v[:, 1] - v[:, 0])
    e2 = np.ascontiguousarray(v[:, 2] - v[:, 0])
    nrm = np.zeros((n, 3), np.float32)
    G.ref_glm_normalize_cross(vp(e1), vp(e2), ctypes.c_uint64(n), vp(nrm))
    m = np.zeros((n, 4, 4), np.float32)  # [triangle][column][row]
    m[:, 0, :3], m[:, 1, :3], m[:, 2, :3] = e1, e2, nrm - v[:, 0]
    m[:, 3, :3], m[:, 3, 3] = v[:, 0], 1.0
    out = np.zeros((n, 4, 4), np.float32)
    G.ref_glm_transpose_inverse(vp(m), ctypes.c_uint64(n), vp(out))
    return np.ascontiguousarray(out[:, :3, :].reshape(n, 12))


def pin_triangles():
    rng = np.random.default_rng(11)
    parts = [np.load(os.path.join(HERE, "cornell_box.npz"))["verts"]]
    parts.append(S.soup(1500, seed=12, edge=0.05))
    parts.append(S.knot_mesh(24, 24))
    parts.append((S.soup(300, seed=13, edge=1e-3) * np.float32(100.0)).astype(np.float32))  # far from the origin
    # needles and near-degenerate triangles: where n . v1 is close to 1 the matrix is near-singular
    base = rng.uniform(-1, 1, (200, 1, 3)).astype(np.float32)
    thin = (base + rng.uniform(-1, 1, (200, 3, 3)).astype(np.float32) * np.float32([1.0, 1e-3, 1e-5])).astype(np.float32)
    parts.append(thin.reshape(-1, 9))
    return np.ascontiguousarray(np.concatenate(parts), np.float32)


def main():
    verts = pin_triangles()
    n = verts.shape[0]
    woop = ref_woop(verts)
    T = ctypes.CDLL(os.path.join(REF, "libnova_ref_triangle.so"))
    T.ref_triangle_write.restype = ctypes.c_uint64
    bounds = np.zeros((n, 6), np.float32)
    hexes = []
    buf = ctypes.create_string_buffer(256)
    back = np.zeros(9, np.float32)
    for i in range(n):
        T.ref_triangle_bounds(vp(verts[i]), vp(bounds[i]))
        if i < 200:
            assert T.ref_triangle_write(vp(verts[i]), buf, ctypes.c_uint64(256)) > 0
            hexes.append(buf.value.decode())
            assert T.ref_triangle_read(buf.value, vp(back)) == 1 and back.tobytes() == verts[i].tobytes()
    V = ctypes.CDLL(os.path.join(REF, "libnova_ref_vector_utils.so"))
    rng = np.random.default_rng(14)
    rgba = rng.standard_normal((37 * 23, 4)).astype(np.float32)
    rgba[5] = [np.nan, np.inf, -0.0, 7.0]
    flat = np.zeros(rgba.shape[0] * 3, np.float32)
    V.ref_flatten_float4_to_3(vp(rgba), ctypes.c_uint64(rgba.shape[0]), vp(flat))
    packed = np.full((rgba.shape[0], 4), 9.0, np.float32)
    V.ref_pack_3_to_float4(vp(flat), ctypes.c_uint64(rgba.shape[0]), vp(packed))
    out = os.path.join(HERE, "ref_pins.npz")
    np.savez_compressed(out, verts=verts, woop=woop, bounds=bounds, hex_text=np.asarray(hexes),
                        rgba=rgba, flat=flat, packed=packed, image_dims=np.int32([37, 23]))
    print(out, os.path.getsize(out), "bytes;", n, "triangles")


if __name__ == "__main__":
    main()
