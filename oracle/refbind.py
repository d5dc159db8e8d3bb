"""TEST INFRASTRUCTURE ONLY.  ctypes binding to oracle/_ref/libnova_ref.so — Nova's own kernel
headers compiled for the CPU (oracle/ref_driver.cpp).  Exists only in the build container
(/root/reference is not on the GPU box); used by tests/golden/make_golden.py and by the CPU
tests that validate the restatement (skipped when the library is absent).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libnova_ref.so")


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ref_hash.restype = ctypes.c_uint32
        _lib.ref_hash.argtypes = [ctypes.c_uint32]
        _lib.ref_rand.restype = ctypes.c_float
        _lib.ref_rand.argtypes = [ctypes.c_void_p]
        _lib.ref_intersects_aabb.restype = ctypes.c_int
        _lib.ref_intersects_aabb.argtypes = [ctypes.c_void_p] * 4
        _lib.ref_brdf.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_float,
                                                          ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _f4img(w, h):
    return np.zeros((h, w, 4), np.float32)


def kernel_raytrace(scene, params, time, width, height, denoise_available=0, meta=None,
                    sky=None):
    """kernels/kernels.cu:9-40.  Returns (color, albedo, normal) float32[h, w, 4]."""
    L = lib()
    meta = scene.meta if meta is None else meta
    sky = scene.sky if sky is None else sky
    color, albedo, normal = _f4img(width, height), _f4img(width, height), _f4img(width, height)
    mats = scene.materials
    params = np.ascontiguousarray(params)
    rc = L.ref_kernel_raytrace(
        _p(params), ctypes.c_uint32(time), _p(color), ctypes.c_uint32(width),
        ctypes.c_uint32(height), _p(scene.triangles), _p(meta),
        ctypes.c_uint64(scene.triangles.shape[0]), _p(scene.bvh),
        ctypes.c_uint64(scene.bvh.shape[0]), _p(scene.lights) if scene.num_lights else None,
        ctypes.c_uint32(scene.num_lights), _p(mats), ctypes.c_uint32(mats.shape[2]),
        ctypes.c_uint32(mats.shape[1]), ctypes.c_uint32(mats.shape[0]), _p(sky),
        ctypes.c_uint32(sky.shape[1]), ctypes.c_uint32(sky.shape[0]),
        ctypes.c_uint32(denoise_available), _p(albedo), _p(normal))
    assert rc == 0
    return color, albedo, normal


def kernel_accumulate(sample_index, denoise_available, color1, albedo1, normal1, prev_color,
                      prev_albedo, prev_normal):
    """kernels/kernels.cu:42-92.  Returns (color2, albedo2, normal2)."""
    h, w = color1.shape[:2]
    out = [_f4img(w, h) for _ in range(3)]
    ins = [np.ascontiguousarray(a, np.float32) for a in
           (color1, albedo1, normal1, prev_color, prev_albedo, prev_normal)]
    rc = lib().ref_kernel_accumulate(ctypes.c_int32(sample_index),
                                     ctypes.c_uint32(denoise_available), *[_p(a) for a in ins],
                                     *[_p(a) for a in out], ctypes.c_uint32(w), ctypes.c_uint32(h))
    assert rc == 0
    return tuple(out)


def kernel_post_process(params, color2):
    """kernels/kernels.cu:94-116.  Returns uint8[h, w, 4]."""
    h, w = color2.shape[:2]
    out = np.zeros((h, w, 4), np.uint8)
    color2 = np.ascontiguousarray(color2, np.float32)
    params = np.ascontiguousarray(params)
    rc = lib().ref_kernel_post_process(_p(params), _p(color2), _p(out), ctypes.c_uint32(w),
                                       ctypes.c_uint32(h))
    assert rc == 0
    return out


def find_intersection(scene, point_tmax, dir_flags):
    """kernels/raytrace.hpp:38-102 on a batch.  Returns hits float32[n, 4] (t, u, v, id bits)."""
    point_tmax = np.ascontiguousarray(point_tmax, np.float32)
    dir_flags = np.ascontiguousarray(dir_flags, np.float32)
    n = point_tmax.shape[0]
    hits = np.zeros((n, 4), np.float32)
    rc = lib().ref_find_intersection(_p(scene.triangles), ctypes.c_uint64(scene.triangles.shape[0]),
                                     _p(scene.bvh), ctypes.c_uint64(scene.bvh.shape[0]),
                                     _p(point_tmax), _p(dir_flags), _p(hits), ctypes.c_uint64(n))
    assert rc == 0
    return hits


def struct_sizes():
    out = np.zeros(6, np.int32)
    lib().ref_struct_sizes(_p(out))
    return [int(x) for x in out]
