"""CPU: the known answers the reference's own unit tests hold (tests/src/intersectables/aabb.cpp,
triangle.cpp) checked on this repository's host-side AABB / triangle helpers (the scene builder that
feeds the hot path), plus the reference's hex dump format for FlatBVHNode / Triangle."""
import ctypes

import numpy as np

from nova_amd import scene as S

FMAX = np.finfo(np.float32).max


def aabb(top, bottom, n=5):
    out = np.zeros(5, np.float32)
    t, b = np.float32(top), np.float32(bottom)
    S.scene_lib().nova_scene_test_aabb(ctypes.c_void_p(t.ctypes.data), ctypes.c_void_p(b.ctypes.data),
                                       ctypes.c_uint64(n), ctypes.c_void_p(out.ctypes.data))
    return out


def grow(ta, ba, tb, bb):
    out = np.zeros(6, np.float32)
    arrs = [np.float32(x) for x in (ta, ba, tb, bb)]
    S.scene_lib().nova_scene_test_aabb_grow(*[ctypes.c_void_p(a.ctypes.data) for a in arrs],
                                            ctypes.c_void_p(out.ctypes.data))
    return out[:3], out[3:]


def test_aabb_surface_area():  # aabb.cpp:7-30
    a1 = aabb([1, 1, 1], [0, 0, 0])
    assert np.isclose(a1[0], 6.0)
    assert np.isclose(a1[0], aabb([1.1] * 3, [0.1] * 3)[0])
    assert a1[0] > aabb([1.0] * 3, [0.1] * 3)[0]


def test_aabb_sah_cost():  # aabb.cpp:32-55
    assert np.isclose(aabb([1] * 3, [0] * 3, 5)[1], 30.0)
    assert np.isclose(aabb([1] * 3, [0] * 3, 5)[1], aabb([1.1] * 3, [0.1] * 3, 5)[1])
    assert aabb([1] * 3, [0] * 3, 6)[1] > aabb([1.0] * 3, [0.1] * 3, 5)[1]


def test_aabb_center():  # aabb.cpp:57-64
    assert (aabb([1] * 3, [0] * 3)[2:] == np.float32([0.5, 0.5, 0.5])).all()


def test_aabb_grow():  # aabb.cpp:66-123
    t, b = grow([-FMAX] * 3, [FMAX] * 3, [1, 2, 3], [-1, -2, -3])  # from the empty box
    assert (t == [1, 2, 3]).all() and (b == [-1, -2, -3]).all()
    t, b = grow([1, 1, 1], [0, 0, 0], [0.5, 0.5, 0.5], [0.2, 0.2, 0.2])  # contained: unchanged
    assert (t == [1, 1, 1]).all() and (b == [0, 0, 0]).all()
    t, b = grow([1, 1, 1], [0, 0, 0], [2, 0.5, 3], [-1, 0.2, 0.5])  # partial overlap
    assert (t == [2, 1, 3]).all() and (b == [-1, 0, 0]).all()
    t, b = grow([1, 1, 1], [0, 0, 0], [5, 5, 5], [4, 4, 4])  # disjoint
    assert (t == [5, 5, 5]).all() and (b == [0, 0, 0]).all()


def test_triangle_bounds():  # triangle.cpp:8-21
    v = np.float32([1.32, 3.56, -4.12, 0, 0, 0, -5.62, 9.83, -2.21])
    out = np.zeros(6, np.float32)
    S.scene_lib().nova_scene_test_tri_bounds(ctypes.c_void_p(v.ctypes.data), ctypes.c_void_p(out.ctypes.data))
    assert (out[:3] == np.float32([1.32, 9.83, 0.0])).all()
    assert (out[3:] == np.float32([-5.62, 0.0, -4.12])).all()


def test_hex_serialisation_round_trip(cornell):  # triangle.cpp:23-36, bvh.cpp:251-281
    tri = np.float32([1.32, 3.56, -4.12, 0, 0, 0, -5.62, 9.83, -2.21])
    text = S.to_hex_words(tri)
    assert text.split()[0] == "3fa8f5c3"  # 1.32f
    assert (S.from_hex_words(text, np.float32).view(np.uint32) == tri.view(np.uint32)).all()
    other = tri.copy()
    other[0] = np.float32(1.321)
    assert S.to_hex_words(other) != text  # 1.32 vs 1.321 stay distinct (triangle.cpp:38-55)
    back = S.from_hex_words(S.to_hex_words(cornell.bvh), S.BVH_NODE)
    assert back.tobytes() == cornell.bvh.tobytes()


def _ref_aabb_lib():
    import os
    path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libnova_ref_aabb.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def test_box_helpers_bit_equal_to_the_reference_aabb():
    """oracle/_ref/libnova_ref_aabb.so = the reference's header-only nova::AABB on its vendored glm,
    compiled unmodified (oracle/ref_aabb.cpp); exists only where /root/reference does."""
    import pytest

    R = _ref_aabb_lib()
    if R is None:
        pytest.skip("oracle/_ref is built only where /root/reference exists")
    rng = np.random.default_rng(7)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    empty = np.zeros(6, np.float32)
    R.ref_aabb_empty(vp(empty))
    assert (empty[:3] == -FMAX).all() and (empty[3:] == FMAX).all()
    for i in range(2000):
        scale = np.float32(10.0 ** rng.integers(-3, 4))
        b0 = (rng.uniform(-1, 1, 3) * scale).astype(np.float32)
        t0 = (b0 + rng.uniform(0, 1, 3) * scale).astype(np.float32)
        b1 = (rng.uniform(-1, 1, 3) * scale).astype(np.float32)
        t1 = (b1 + rng.uniform(0, 1, 3) * scale).astype(np.float32)
        n = int(rng.integers(1, 1 << 20))
        want = np.zeros(5, np.float32)
        R.ref_aabb(vp(t0), vp(b0), ctypes.c_uint64(n), vp(want))
        assert aabb(t0, b0, n).tobytes() == want.tobytes(), i
        g = np.zeros(6, np.float32)
        R.ref_aabb_combine(0, vp(t0), vp(b0), vp(t1), vp(b1), vp(g))
        t, b = grow(t0, b0, t1, b1)
        assert np.concatenate([t, b]).tobytes() == g.tobytes(), i
        if i % 10 == 0:  # growing the empty box yields the other box (bvh.cpp builds bounds this way)
            et, eb = empty[:3].copy(), empty[3:].copy()  # keep alive while the pointers are in use
            R.ref_aabb_combine(0, vp(et), vp(eb), vp(t1), vp(b1), vp(g))
            t, b = grow(et, eb, t1, b1)
            assert np.concatenate([t, b]).tobytes() == g.tobytes()
