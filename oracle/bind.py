"""TEST INFRASTRUCTURE ONLY.  ctypes binding to oracle/libnova_oracle.so (oracle/nova_oracle.cpp),
the CPU restatement of Nova's hot path.  Importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product path (nova_amd/) never imports it.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnova_oracle.so")

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build() first")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_hash.restype = ctypes.c_uint32
        _lib.oracle_hash.argtypes = [ctypes.c_uint32]
        _lib.oracle_rand.restype = ctypes.c_float
        _lib.oracle_rand.argtypes = [ctypes.c_void_p]
        _lib.oracle_intersects_aabb.restype = ctypes.c_int
        _lib.oracle_intersects_aabb.argtypes = [ctypes.c_void_p] * 4
        _lib.oracle_brdf.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_float,
                                                             ctypes.c_void_p, ctypes.c_void_p]
        _lib.oracle_set_threads.argtypes = [ctypes.c_int]
        _lib.oracle_set_clamp_to_edge.argtypes = [ctypes.c_int]
        _lib.oracle_set_mis_pretest.argtypes = [ctypes.c_uint32]
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def set_threads(n: int):
    lib().oracle_set_threads(int(n))


def set_clamp_to_edge(on: bool):
    """CLAMP beyond the image: False = border colour 0 (OpenCL backend, default), True = edge texel (CUDA backend)."""
    lib().oracle_set_clamp_to_edge(1 if on else 0)


def kernel_raytrace(scene, params, time, width, height, denoise_available=0, meta=None, sky=None,
                    max_depth=0, ordered=False, count_work=False, rows=None, bands=None, counters=None,
                    mis_pretest=False):
    """kernels/kernels.cu:9-40.  Returns (color, albedo, normal) float32[h, w, 4]; `counters`
    (uint64[8], optional) accumulates primary/bounce/shadow rays, segments, node visits, tri tests and — with
    `mis_pretest` (libnova_hip's two rules for rays whose answer cannot matter, not in the reference: nova_oracle.cpp) —
    the MIS rays / shadow rays not traced ([6], [7])."""
    meta = scene.meta if meta is None else meta
    lib().oracle_set_mis_pretest(int(meta.shape[0]) if mis_pretest else 0)
    sky = scene.sky if sky is None else sky
    color = np.zeros((height, width, 4), np.float32)
    albedo = np.zeros((height, width, 4), np.float32)
    normal = np.zeros((height, width, 4), np.float32)
    mats = scene.materials
    params = np.ascontiguousarray(params)
    r0, r1 = (0, height) if rows is None else rows
    band_rows, band_period = (0, 0) if bands is None else bands  # render 1 band of rows in `period`
    rc = lib().oracle_kernel_raytrace(
        _p(params), ctypes.c_uint32(time), _p(color), ctypes.c_uint32(width),
        ctypes.c_uint32(height), _p(scene.triangles), _p(meta), _p(scene.bvh),
        _p(scene.lights) if scene.num_lights else None, ctypes.c_uint32(scene.num_lights),
        _p(mats), ctypes.c_uint32(mats.shape[2]), ctypes.c_uint32(mats.shape[1]),
        ctypes.c_uint32(mats.shape[0]), _p(sky), ctypes.c_uint32(sky.shape[1]),
        ctypes.c_uint32(sky.shape[0]), ctypes.c_uint32(denoise_available), _p(albedo), _p(normal),
        ctypes.c_uint32(max_depth), ctypes.c_int(1 if ordered else 0),
        ctypes.c_int(1 if count_work else 0), ctypes.c_uint32(r0), ctypes.c_uint32(r1),
        ctypes.c_uint32(band_rows), ctypes.c_uint32(band_period), _p(counters))
    lib().oracle_set_mis_pretest(0)
    assert rc == 0
    return color, albedo, normal


def kernel_accumulate(sample_index, denoise_available, color1, albedo1, normal1, prev_color,
                      prev_albedo, prev_normal):
    h, w = color1.shape[:2]
    out = [np.zeros((h, w, 4), np.float32) for _ in range(3)]
    ins = [np.ascontiguousarray(a, np.float32) for a in
           (color1, albedo1, normal1, prev_color, prev_albedo, prev_normal)]
    rc = lib().oracle_kernel_accumulate(ctypes.c_int32(sample_index),
                                        ctypes.c_uint32(denoise_available),
                                        *[_p(a) for a in ins], *[_p(a) for a in out],
                                        ctypes.c_uint32(w), ctypes.c_uint32(h))
    assert rc == 0
    return tuple(out)


def kernel_post_process(params, color2):
    h, w = color2.shape[:2]
    out = np.zeros((h, w, 4), np.uint8)
    color2 = np.ascontiguousarray(color2, np.float32)
    params = np.ascontiguousarray(params)
    rc = lib().oracle_kernel_post_process(_p(params), _p(color2), _p(out), ctypes.c_uint32(w),
                                          ctypes.c_uint32(h))
    assert rc == 0
    return out


def trace_rays(scene, point_tmax, dir_flags, ordered=False, counters=None, per_ray=None):
    """find_intersection on a batch; hits float32[n,4] = (t, u, v, id bits).  per_ray (optional
    uint32[n, 2]) receives every ray's box tests and triangle tests."""
    point_tmax = np.ascontiguousarray(point_tmax, np.float32)
    dir_flags = np.ascontiguousarray(dir_flags, np.float32)
    n = point_tmax.shape[0]
    hits = np.zeros((n, 4), np.float32)
    rc = lib().oracle_trace_rays_work(_p(scene.triangles), _p(scene.bvh), _p(point_tmax), _p(dir_flags),
                                      _p(hits), ctypes.c_uint64(n), ctypes.c_int(1 if ordered else 0),
                                      _p(counters), _p(per_ray))
    assert rc == 0
    return hits


MATH_FNS = {"sin": 0, "cos": 1, "atan": 2, "atan2": 3, "asin": 4, "acos": 5, "exp": 6, "log": 7,
            "pow": 8}


def math_fn(name: str, x, y=None):
    """include/nova_math.h evaluated on the host."""
    x = np.ascontiguousarray(x, np.float32)
    y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.empty_like(x)
    lib().oracle_math(ctypes.c_int(MATH_FNS[name]), _p(x), _p(y), _p(out), ctypes.c_uint64(x.size))
    return out
