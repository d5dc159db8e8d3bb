"""CPU: independent checks of the oracle (the CPU restatement), because no output of the reference
exists in this image to pin it ("parity unpinned").  Nothing here shares code with
oracle/nova_oracle.cpp: closest hits come from a brute-force float64 Moller-Trumbore over the scene's
source vertices, hash / xorshift from Python integers following the published algorithms, the BRDF
from the textbook Cook-Torrance formulas in float64, and the rest are properties any correct
implementation of the reference's kernels has (exact linearity in light power, FXAA leaves flat
images alone, the running mean)."""
import ctypes

import numpy as np
import pytest

from nova_amd import scene as S
from oracle import bind as O
from tests.conftest import random_rays

RAY_EPSILON = 1e-2  # kernels/constants.hpp:10


# ---------------------------------------------------------------------------------------------
# closest hit against brute-force geometry
# ---------------------------------------------------------------------------------------------
def brute_force(verts, point_tmax, dir_flags, chunk=512):
    """float64 Moller-Trumbore of every ray against every triangle.  Returns per ray the best and
    second-best t, the best id and the smallest distance of the best hit to a triangle edge (in
    barycentric units), for deciding which rays are too close to call."""
    v = verts.reshape(-1, 3, 3).astype(np.float64)
    v0, e1, e2 = v[:, 0], v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    n = point_tmax.shape[0]
    best_t = np.full(n, np.inf)
    second_t = np.full(n, np.inf)
    best_id = np.full(n, -1, np.int64)
    edge = np.full(n, np.inf)
    for s in range(0, n, chunk):
        d = dir_flags[s:s + chunk, :3].astype(np.float64)
        o = point_tmax[s:s + chunk, :3].astype(np.float64) + d * RAY_EPSILON
        tmax = point_tmax[s:s + chunk, 3].astype(np.float64)
        p = np.cross(d[:, None, :], e2[None])                  # [r, t, 3]
        det = (p * e1[None]).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o[:, None, :] - v0[None]
            u = (tv * p).sum(-1) * inv
            q = np.cross(tv, e1[None])
            w = (q * d[:, None, :]).sum(-1) * inv
            t = (q * e2[None]).sum(-1) * inv
        ok = np.isfinite(t) & (u >= 0) & (w >= 0) & (u + w <= 1) & (t >= 0) & (t < tmax[:, None])
        t = np.where(ok, t, np.inf)
        order = np.argsort(t, axis=1)[:, :2]
        r = np.arange(t.shape[0])
        best_t[s:s + chunk] = t[r, order[:, 0]]
        second_t[s:s + chunk] = t[r, order[:, 1]] if t.shape[1] > 1 else np.inf
        bid = order[:, 0]
        best_id[s:s + chunk] = np.where(np.isfinite(t[r, bid]), bid, -1)
        ub, wb = u[r, bid], w[r, bid]
        edge[s:s + chunk] = np.where(np.isfinite(t[r, bid]), np.minimum(np.minimum(ub, wb), 1 - ub - wb), np.inf)
    return best_t, second_t, best_id, edge


@pytest.mark.parametrize("which", ["cornell", "soup"])
@pytest.mark.parametrize("ordered", [False, True])
def test_closest_hits_agree_with_brute_force_geometry(which, ordered, cornell):
    sc = cornell if which == "cornell" else S.build_scene(S.soup(3000, seed=9, edge=0.25))
    pt, df = random_rays(6000, seed=21, tmax_every=5)
    hits = O.trace_rays(sc, pt, df, ordered=ordered)
    ids = hits[:, 3].view(np.int32).astype(np.int64)
    bt, st, bid, edge = brute_force(sc.verts, pt, df)
    # rays the float32 Woop test may legitimately decide differently: grazing an edge, two surfaces
    # at (nearly) the same distance, a hit at the very start or end of the ray's interval
    tmax = pt[:, 3].astype(np.float64)
    tol = 2e-4 * (1.0 + np.where(np.isfinite(bt), bt, 0.0))
    with np.errstate(invalid="ignore"):
        ambiguous = (edge < 2e-4) | (st - bt < tol) | (bt < tol) | (np.abs(tmax - bt) < tol)
    clear = ~ambiguous
    assert clear.mean() > 0.95  # the Cornell box has a few coincident faces (boxes standing on the floor)
    assert (ids[clear] == bid[clear]).all(), np.flatnonzero(clear & (ids != bid))[:10]
    hit = clear & (bid >= 0)
    assert hit.sum() > 1000
    assert np.allclose(hits[hit, 0], bt[hit], rtol=2e-4, atol=2e-5)


def test_any_hit_rays_report_a_hit_iff_geometry_has_one(cornell):
    pt, df = random_rays(4000, seed=22, anyhit_every=1, tmax_every=2)
    hits = O.trace_rays(cornell, pt, df)
    ids = hits[:, 3].view(np.int32)
    bt, st, bid, edge = brute_force(cornell.verts, pt, df)
    tmax = pt[:, 3].astype(np.float64)
    tol = 2e-4 * (1.0 + np.where(np.isfinite(bt), bt, 0.0))
    # any-hit may return any occluder: only "is there one" is defined.  A ray is clear when its
    # nearest occluder is clearly inside the interval and away from triangle edges, or absent.
    with np.errstate(invalid="ignore"):
        clear = (bid < 0) | ((edge > 2e-4) & (bt > tol) & (np.abs(tmax - bt) > tol))
    assert clear.mean() > 0.9
    assert ((ids[clear] >= 0) == (bid[clear] >= 0)).all()


def test_woop_matrices_map_each_triangle_to_the_unit_triangle(cornell):
    """intersectable_manager.cpp:129-145: rows 0-2 of inverse([e1 e2 n-v1 v1]) send v1, v2, v3 to
    (0,0,0), (1,0,0), (0,1,0)."""
    for sc in (cornell, S.build_scene(S.soup(500, seed=3, edge=0.3))):
        m = np.stack([sc.triangles["x"], sc.triangles["y"], sc.triangles["z"]], 1)
        m = m.astype(np.float64)                                  # [n, 3, 4]
        v = sc.verts.reshape(-1, 3, 3).astype(np.float64)
        h = np.concatenate([v, np.ones(v.shape[:2] + (1,))], -1)  # [n, vertex, 4]
        img = np.einsum("nrk,nvk->nvr", m, h)                     # [n, vertex, 3]
        want = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float64)
        scale = np.abs(m).max(axis=(1, 2))[:, None, None] * np.abs(h).max()
        assert (np.abs(img - want[None]) <= 1e-5 * np.maximum(scale, 1.0)).all()


# ---------------------------------------------------------------------------------------------
# hash / rand from the published algorithms, in Python integers
# ---------------------------------------------------------------------------------------------
def wang_hash(s):  # Thomas Wang's 32-bit integer hash, as used at kernels/random.hpp:11-18
    m = 0xFFFFFFFF
    s = ((s ^ 61) ^ (s >> 16)) & m
    s = (s * 9) & m
    s = s ^ (s >> 4)
    s = (s * 0x27D4EB2D) & m
    return s ^ (s >> 15)


def xorshift32(s):  # Marsaglia (13, 17, 5)
    m = 0xFFFFFFFF
    s ^= (s << 13) & m
    s ^= s >> 17
    s ^= (s << 5) & m
    return s


def test_hash_and_xorshift_follow_the_published_algorithms():
    L = O.lib()
    rng = np.random.default_rng(5)
    for s in [0, 1, 61, 0xFFFFFFFF] + [int(x) for x in rng.integers(0, 2 ** 32, 200)]:
        assert L.oracle_hash(s) == wang_hash(s)
    for s0 in [1, 12345, 0x9E3779B9, 0xFFFFFFFF]:
        st = np.array([s0], np.uint32)
        py = s0
        for _ in range(16):
            val = L.oracle_rand(ctypes.c_void_p(st.ctypes.data))
            py = xorshift32(py)
            assert int(st[0]) == py
            # random.hpp:21-26: u32 -> f32 (round to nearest) divided by (float) UINT_MAX = 2^32
            assert np.float32(val) == np.float32(np.float32(py) / np.float32(4294967296.0))
    st = np.array([0], np.uint32)  # state 0 is absorbing
    assert L.oracle_rand(ctypes.c_void_p(st.ctypes.data)) == 0.0 and st[0] == 0


# ---------------------------------------------------------------------------------------------
# Cook-Torrance against the textbook formulas in float64
# ---------------------------------------------------------------------------------------------
def test_brdf_eval_and_pdf_match_float64_cook_torrance():
    L = O.lib()
    rng = np.random.default_rng(17)
    n_cases = 3000
    checked = 0
    for i in range(n_cases):
        normal = rng.standard_normal(3)
        normal /= np.linalg.norm(normal)
        out = normal + 0.8 * rng.standard_normal(3)
        out /= np.linalg.norm(out)
        if out @ normal < 0.05:
            continue
        dif = rng.uniform(0.05, 1, 3)
        met, rough = float(rng.uniform(0, 1)), float(rng.uniform(0.2, 1))  # D is ill-conditioned in f32 below
        o32, n32, d32 = (np.ascontiguousarray(a, np.float32) for a in (out, normal, dif))
        st = np.array([rng.integers(1, 2 ** 32)], np.uint32)
        res = np.zeros(9, np.float32)
        L.oracle_brdf(ctypes.c_void_p(o32.ctypes.data), ctypes.c_void_p(n32.ctypes.data),
                      ctypes.c_void_p(d32.ctypes.data), ctypes.c_float(met), ctypes.c_float(rough),
                      ctypes.c_void_p(st.ctypes.data), ctypes.c_void_p(res.ctypes.data))
        wi = res[:3].astype(np.float64)
        assert abs(np.linalg.norm(wi) - 1.0) < 1e-4
        o, nn, kd_in = o32.astype(np.float64), n32.astype(np.float64), d32.astype(np.float64)
        met, rough = float(np.float32(met)), float(np.float32(rough))
        h = wi + o
        h /= np.linalg.norm(h)
        i_m, i_n, o_n, n_m = wi @ h, wi @ nn, max(o @ nn, 0.0), max(nn @ h, 0.0)
        if i_m <= 1e-3 or i_n <= 1e-3 or o_n <= 1e-3:
            continue  # below the horizon / grazing: eval is cut to 0, clamps dominate
        a2 = rough ** 4
        D = a2 / (np.pi * (n_m * n_m * (a2 - 1) + 1) ** 2)
        f0 = 0.04 + (kd_in - 0.04) * met
        F = f0 + (1 - f0) * max(1 - i_m, 0.0) ** 5
        k = (rough + 1) ** 2 / 8
        G = (i_n / (i_n * (1 - k) + k)) * (o_n / (o_n * (1 - k) + k))
        k_d = (1 - F) * (1 - met)
        f = k_d * kd_in / np.pi + D * F * G / (4 * o_n * i_n)
        t = np.clip(F / (F + k_d), 0, 1)
        pdf = (1 - t) * i_n / np.pi + t * D * n_m / (4 * i_m)
        assert np.allclose(res[3:6], f, rtol=2e-3, atol=1e-5), (i, res[3:6], f)
        assert np.allclose(res[6:9], pdf, rtol=2e-3, atol=1e-5), (i, res[6:9], pdf)
        checked += 1
    assert checked > n_cases // 2


# ---------------------------------------------------------------------------------------------
# properties of the kernels
# ---------------------------------------------------------------------------------------------
def test_radiance_is_exactly_linear_in_light_power(cornell):
    """Doubling every emitter (area lights and sky) doubles every pixel bit for bit: emission only
    ever multiplies the throughput, and Russian roulette looks at the throughput alone."""
    w = h = 96
    cam = S.Camera(width=w, height=h, **S.CORNELL_CAMERA)
    p = S.make_scene_params(cam, path_tracing=True)
    base = O.kernel_raytrace(cornell, p, 3, w, h, max_depth=4)[0]
    import copy
    sc2 = copy.copy(cornell)
    sc2.lights = cornell.lights.copy()
    sc2.lights["intensity"] *= np.float32(2.0)
    sky2 = cornell.sky * np.float32(2.0)
    twice = O.kernel_raytrace(sc2, p, 3, w, h, max_depth=4, sky=sky2)[0]
    assert base[..., :3].max() > 0
    assert (twice[..., :3] == base[..., :3] * np.float32(2.0)).all()


def test_running_mean_and_flat_image_post_process():
    rng = np.random.default_rng(1)
    h, w = 24, 40
    prev = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    cur = rng.uniform(0, 2, (h, w, 4)).astype(np.float32)
    z = np.zeros_like(prev)
    for i in (0, 1, 7, 1000):
        acc = O.kernel_accumulate(i, 0, cur, z, z, prev, z, z)[0]
        want = (prev[..., :3].astype(np.float64) * i + cur[..., :3]) / (i + 1)   # kernels.cu:61-63
        assert np.allclose(acc[..., :3], want, rtol=3e-7, atol=1e-7)
    # a flat image has no edges: FXAA must return it untouched, so both settings give the same bytes,
    # and the bytes follow tone map -> gamma -> truncation (transforms.hpp:14-26)
    flat = np.zeros((32, 32, 4), np.float32)
    flat[..., :3] = np.float32([0.3, 0.6, 1.7])
    cam = S.Camera(width=32, height=32, **S.CORNELL_CAMERA)
    plain = O.kernel_post_process(S.make_scene_params(cam, anti_aliasing=False, exposure=1.5), flat)
    fx = O.kernel_post_process(S.make_scene_params(cam, anti_aliasing=True, exposure=1.5), flat)
    inner = (slice(2, -2), slice(2, -2))  # CLAMP = border colour 0 makes the outermost pixels an edge
    assert (plain[inner] == fx[inner]).all()
    want = (1.0 - np.exp(-np.float64([0.3, 0.6, 1.7]) * 1.5)) ** (1 / 2.2) * 255.0
    assert (np.abs(plain[5, 5, :3].astype(np.float64) - np.floor(want)) <= 1).all()
    # CLAMP as the CUDA backend's clamp-to-edge (oracle_set_clamp_to_edge / NOVA_HIP_OPT_CLAMP_TO_EDGE): no black
    # ring around the frame, so FXAA leaves the WHOLE flat image alone
    O.set_clamp_to_edge(True)
    try:
        fx_edge = O.kernel_post_process(S.make_scene_params(cam, anti_aliasing=True, exposure=1.5), flat)
    finally:
        O.set_clamp_to_edge(False)
    assert (fx_edge == plain).all()
    assert (fx != plain).any()  # border colour 0 (the default, OpenCL backend) does touch the ring
