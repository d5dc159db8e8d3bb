"""CPU: the oracle (oracle/nova_oracle.cpp, the CPU restatement) against the vectors committed under
tests/golden/.  PARITY UNPINNED: the vectors were produced by the restatement itself
(tests/golden/make_golden.py) because the reference's kernels cannot be built or run in this image;
this file guards against drift (compiler, flags, later edits) and checks that the scene fixtures
still hash to what the vectors were rendered from.  Integer / index / hit records: bit-exact.
Radiance and features: rel-RMSE <= 1e-6.  Bytes: exact."""
import ctypes

import numpy as np
import pytest

from nova_amd import scene as S
from oracle import bind as O
from tests.conftest import rel_rmse
from tests.helpers import SOUP_CAMERA, check_scene, golden, id_inputs


def test_hash_and_rand():
    g = golden("kat.npz")
    L = O.lib()
    assert [L.oracle_hash(int(s)) for s in g["hash_in"]] == [int(x) for x in g["hash_out"]]
    for i, s0 in enumerate(g["rand_state"]):
        st = np.array([s0], np.uint32)
        for k in range(8):
            v = L.oracle_rand(ctypes.c_void_p(st.ctypes.data))
            assert np.float32(v) == g["rand_val"][i, k] and st[0] == g["rand_after"][i, k]
    # the divisor (float)UINT_MAX is 2^32: rand reaches exactly 1.0 (random.hpp:21-26)
    assert g["rand_val"].max() <= 1.0


def test_intersects_aabb():
    g = golden("kat.npz")
    L = O.lib()
    got = []
    for i in range(g["aabb_hit"].shape[0]):
        rows = [np.ascontiguousarray(g[k][i]) for k in ("aabb_point", "aabb_dir", "aabb_top", "aabb_bottom")]
        got.append(L.oracle_intersects_aabb(*[ctypes.c_void_p(r.ctypes.data) for r in rows]))
    assert (np.array(got) == g["aabb_hit"]).all()
    assert 0 < g["aabb_hit"].sum() < g["aabb_hit"].shape[0]


def test_brdf_sample_eval_pdf():
    g = golden("kat.npz")
    L = O.lib()
    n = g["brdf_result"].shape[0]
    out = np.zeros((n, 9), np.float32)
    states = g["brdf_rng"].copy()
    for i in range(n):
        o, nn, dd = (np.ascontiguousarray(g[k][i]) for k in ("brdf_out_dir", "brdf_normal", "brdf_diffuse"))
        st = states[i:i + 1]
        r = np.zeros(9, np.float32)
        L.oracle_brdf(ctypes.c_void_p(o.ctypes.data), ctypes.c_void_p(nn.ctypes.data),
                      ctypes.c_void_p(dd.ctypes.data), float(g["brdf_metallic"][i]),
                      float(g["brdf_roughness"][i]), ctypes.c_void_p(st.ctypes.data),
                      ctypes.c_void_p(r.ctypes.data))
        out[i] = r
    assert (states == g["brdf_rng_after"]).all()
    assert (out.view(np.uint32) == g["brdf_result"].view(np.uint32)).all()  # bit-exact


@pytest.mark.parametrize("fixture,file,cam,size", [
    ("cornell", "ids_cornell_512.npz", S.CORNELL_CAMERA, 512),
    ("soup50k", "ids_soup_256.npz", SOUP_CAMERA, 256),
    ("knot_small", "ids_knot_256.npz", S.DRAGON_CAMERA, 256),
])
@pytest.mark.parametrize("ordered", [False, True])
def test_primary_hit_ids(fixture, file, cam, size, ordered, request):
    sc = request.getfixturevalue(fixture)
    g = golden(file)
    check_scene(sc, g)
    p = S.make_scene_params(S.Camera(width=size, height=size, **cam), path_tracing=False)
    meta, sky = id_inputs(sc)
    c = O.kernel_raytrace(sc, p, int(g["time"]), size, size, meta=meta, sky=sky, ordered=ordered)[0]
    ids = S.decode_ids(c).reshape(size, size)
    assert (ids == g["ids"]).all()
    assert (g["ids"] >= 0).mean() > 0.3


@pytest.mark.parametrize("fixture,file", [("cornell", "rays_cornell.npz"), ("soup50k", "rays_soup.npz"),
                                          ("knot_small", "rays_knot.npz")])
def test_find_intersection_ray_batches(fixture, file, request):
    sc = request.getfixturevalue(fixture)
    g = golden(file)
    check_scene(sc, g)
    ref = g["hits_bits"]
    anyhit = (g["dir_flags"][:, 3].view(np.uint32) & 1) == 1
    walk = O.trace_rays(sc, g["point_tmax"], g["dir_flags"]).view(np.uint32)
    assert (walk == ref).all()  # the restatement of the reference walk: bit-exact incl. any-hit records
    ordered = O.trace_rays(sc, g["point_tmax"], g["dir_flags"], ordered=True).view(np.uint32)
    assert (ordered[~anyhit] == ref[~anyhit]).all()
    # any-hit: an out-of-order traversal may report a different blocker; occlusion itself must agree
    assert ((ordered[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()


def test_cornell_radiance_accumulate_post_process(cornell):
    g = golden("cornell_radiance.npz")
    check_scene(cornell, g)
    w = h = 128
    cam = S.Camera(width=w, height=h, **S.CORNELL_CAMERA)
    p = S.make_scene_params(cam, path_tracing=True)
    prev = [np.zeros((h, w, 4), np.float32) for _ in range(3)]
    for s in range(16):
        c1, a1, n1 = O.kernel_raytrace(cornell, p, int(g["time"]) + s, w, h, denoise_available=1)
        prev = list(O.kernel_accumulate(s, 1, c1, a1, n1, *prev))
        if s == 0:
            assert rel_rmse(prev[0], g["color_1spp"]) <= 1e-6
            assert rel_rmse(prev[1], g["albedo_1spp"]) <= 1e-6
            assert rel_rmse(prev[2], g["normal_1spp"]) <= 1e-6
    assert rel_rmse(prev[0], g["color_16spp"]) <= 1e-6
    assert rel_rmse(prev[1], g["albedo_16spp"]) <= 1e-6
    assert rel_rmse(prev[2], g["normal_16spp"]) <= 1e-6
    # bytes from the reference's own accumulated plane: exact
    assert (O.kernel_post_process(p, g["color_16spp"]) == g["pixels_16spp"]).all()
    p_aa = S.make_scene_params(cam, path_tracing=True, anti_aliasing=True)
    assert (O.kernel_post_process(p_aa, g["color_16spp"]) == g["pixels_16spp_fxaa"]).all()
    assert (g["pixels_16spp_fxaa"] != g["pixels_16spp"]).any()


def test_fxaa_step_edge():
    g = golden("fxaa_edge.npz")
    cam = S.Camera(width=32, height=32, **S.CORNELL_CAMERA)
    aa = O.kernel_post_process(S.make_scene_params(cam, anti_aliasing=True, exposure=1.5), g["image"])
    plain = O.kernel_post_process(S.make_scene_params(cam, anti_aliasing=False, exposure=1.5), g["image"])
    assert (aa == g["pixels_fxaa"]).all() and (plain == g["pixels_plain"]).all()
    assert (aa != plain).any()


def test_depth_cap_is_a_prefix_of_the_unbounded_path(cornell):
    """The bounce cap is a backend extension (the reference has none): capped at d, radiance is the
    sum of the first d segments' contributions, so it grows monotonically towards the unbounded one."""
    w = h = 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
    full = O.kernel_raytrace(cornell, p, 5, w, h)[0][..., :3]
    last = np.zeros_like(full)
    for d in (1, 2, 4, 8):
        c = O.kernel_raytrace(cornell, p, 5, w, h, max_depth=d)[0][..., :3]
        assert (c >= last - 1e-6).all() and (c <= full + 1e-5).all()
        last = c
