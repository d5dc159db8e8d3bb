"""GPU (-m gpu): the HIP path, called through the C ABI (nova_amd.capi -> libnova_hip.so), against
  * the vectors under tests/golden/*.npz (written by the CPU oracle; parity unpinned),
  * the CPU oracle on seeded inputs,
  * and, at BASELINE.json's full sizes, size-independent properties: the production (ordered,
    culled) traversal must equal the reference walk (fixed order, no culling) ray for ray.
Bar: triangle ids, hit records and bytes bit-exact; radiance rel-RMSE <= 1e-3 (north_star) — in
practice 0: host and device share one arithmetic contract (DESIGN.md "Numerics")."""
import os

import numpy as np
import pytest

from nova_amd import capi, scene as S
from nova_amd.raytracer import Raytracer
from tests.conftest import random_rays, rel_rmse
from tests.helpers import SOUP_CAMERA, check_scene, golden, id_inputs

pytestmark = pytest.mark.gpu
RADIANCE_TOL = 1e-3  # north_star: accumulated radiance within relative RMSE 1e-3


@pytest.fixture(scope="module", autouse=True)
def device():
    capi.init(0)
    yield
    for opt in (capi.OPT_MAX_DEPTH, capi.OPT_TRAVERSAL, capi.OPT_COUNT_WORK, capi.OPT_TILE_RANK,
                capi.OPT_TRAVERSE_KERNEL):
        capi.set_option(opt, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)


def oracle():
    from oracle import bind as O
    return O


def gpu_primary_ids(sc, cam_kw, w, h, time, traversal=0, kernel=0):
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam_kw), path_tracing=False)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    meta, sky = id_inputs(sc)
    rt.set_meta(meta)
    rt.sky_img.write(sky)
    capi.set_option(capi.OPT_TRAVERSAL, traversal)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
    rt.kernel_raytrace(time)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
    return S.decode_ids(rt.temp_color_img1.read()).reshape(h, w)


@pytest.mark.parametrize("fixture,file,cam,size", [
    ("cornell", "ids_cornell_512.npz", S.CORNELL_CAMERA, 512),
    ("soup50k", "ids_soup_256.npz", SOUP_CAMERA, 256),
    ("knot_small", "ids_knot_256.npz", S.DRAGON_CAMERA, 256),
])
@pytest.mark.parametrize("traversal,kernel", [(0, 0), (1, 0), (0, 6), (0, 11), (0, 12)])
def test_primary_hit_ids_match_the_golden_vectors(fixture, file, cam, size, traversal, kernel, request):
    sc = request.getfixturevalue(fixture)
    g = golden(file)
    check_scene(sc, g)
    ids = gpu_primary_ids(sc, cam, size, size, int(g["time"]), traversal, kernel)
    assert (ids == g["ids"]).all()


@pytest.mark.parametrize("fixture,file", [("cornell", "rays_cornell.npz"), ("soup50k", "rays_soup.npz"),
                                          ("knot_small", "rays_knot.npz")])
@pytest.mark.parametrize("traversal", [0, 1])
def test_ray_batches_match_the_golden_vectors(fixture, file, traversal, request):
    sc = request.getfixturevalue(fixture)
    g = golden(file)
    check_scene(sc, g)
    n = g["point_tmax"].shape[0]
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    ro, rd, hb = capi.Buffer(g["point_tmax"]), capi.Buffer(g["dir_flags"]), capi.Buffer(nbytes=16 * n)
    capi.set_option(capi.OPT_TRAVERSAL, traversal)
    capi.trace_rays(tb, bb, ro, rd, hb, n)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    got = hb.read(np.uint32).reshape(n, 4)
    ref = g["hits_bits"]
    anyhit = (g["dir_flags"][:, 3].view(np.uint32) & 1) == 1
    assert (got[~anyhit] == ref[~anyhit]).all()  # t, barycentrics, triangle index: bit-exact
    # any-hit rays stop at the first blocker found: occlusion must agree, the blocker need not
    assert ((got[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()
    assert ((ref[:, 3].view(np.int32) >= 0).mean() > 0.2)


def test_cornell_sixteen_samples_against_the_golden_vectors(cornell):
    """The reference's whole per-sample schedule (raytrace, accumulate, copy, post-process) x16."""
    g = golden("cornell_radiance.npz")
    check_scene(cornell, g)
    w = h = 128
    cam = S.Camera(width=w, height=h, **S.CORNELL_CAMERA)
    p = S.make_scene_params(cam, path_tracing=True)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(cornell, p, w, h)
    rt.start()
    pixels = None
    for s in range(16):
        pixels = rt.raytrace(int(g["time"]) + s)
        if s == 0:
            assert rel_rmse(rt.temp_color_img2.read(), g["color_1spp"]) <= RADIANCE_TOL
            assert rel_rmse(rt.albedo_img2.read(), g["albedo_1spp"]) <= RADIANCE_TOL
            assert rel_rmse(rt.normal_img2.read(), g["normal_1spp"]) <= RADIANCE_TOL
    color = rt.temp_color_img2.read()
    assert rel_rmse(color, g["color_16spp"]) <= RADIANCE_TOL
    assert rel_rmse(rt.albedo_img2.read(), g["albedo_16spp"]) <= RADIANCE_TOL
    assert rel_rmse(rt.normal_img2.read(), g["normal_16spp"]) <= RADIANCE_TOL
    assert (color == g["color_16spp"]).all(-1).mean() > 0.999  # in practice every pixel, bit for bit
    assert (pixels == g["pixels_16spp"]).all()
    rt.params = S.make_scene_params(cam, path_tracing=True, anti_aliasing=True)
    rt.kernel_post_process()
    assert (rt.color_img.read() == g["pixels_16spp_fxaa"]).all()


def test_post_process_on_the_golden_plane():
    g = golden("fxaa_edge.npz")
    cam = S.Camera(width=32, height=32, **S.CORNELL_CAMERA)
    src = capi.Image(32, 32, data=g["image"])
    dst = capi.Image(32, 32, texel_bytes=4)
    capi.launch_post_process(S.make_scene_params(cam, anti_aliasing=True, exposure=1.5), src, dst, 32, 32)
    assert (dst.read() == g["pixels_fxaa"]).all()
    capi.launch_post_process(S.make_scene_params(cam, anti_aliasing=False, exposure=1.5), src, dst, 32, 32)
    assert (dst.read() == g["pixels_plain"]).all()


@pytest.mark.parametrize("fixture,cam,depth", [("knot_small", S.DRAGON_CAMERA, 5), ("knot_small", S.DRAGON_CAMERA, 0),
                                               ("soup50k", SOUP_CAMERA, 0), ("cornell", S.CORNELL_CAMERA, 1)])
def test_path_tracing_against_the_oracle(fixture, cam, depth, request):
    sc = request.getfixturevalue(fixture)
    O = oracle()
    w, h = 320, 200  # not multiples of the 64-pixel tile: ragged tile edges
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.4, shading_roughness=0.1)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    capi.reset_counters()
    rt.kernel_raytrace(42)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    k = capi.counters()
    cnt = np.zeros(8, np.uint64)
    ref = O.kernel_raytrace(sc, p, 42, w, h, denoise_available=1, max_depth=depth, counters=cnt)
    for img, r in zip((rt.temp_color_img1, rt.albedo_img1, rt.normal_img1), ref):
        assert rel_rmse(img.read(), r) <= RADIANCE_TOL
    # the same paths were followed: identical ray and segment counts
    assert (k["rays_primary"], k["rays_bounce"], k["rays_shadow"], k["path_segments"]) == tuple(int(x) for x in cnt[:4])


def test_textures_sky_and_normal_maps_against_the_oracle(knot_small):
    """BASELINE config 5's shading path: layered material array (bilinear, REPEAT), normal map TBN,
    equirect sky — on synthetic textures (the reference's image files need a JPEG decoder)."""
    O = oracle()
    sc = knot_small
    rng = np.random.default_rng(8)
    mats = rng.uniform(0.05, 1.0, (4, 16, 32, 4)).astype(np.float32)
    mats[3, ..., :3] = 0.5 + 0.3 * rng.standard_normal((16, 32, 3)).astype(np.float32)  # normal map
    meta = sc.meta.copy()
    n = meta.shape[0]
    uv = rng.uniform(-2, 3, (n, 3, 2)).astype(np.float32)
    meta["texture_coord1"], meta["texture_coord2"], meta["texture_coord3"] = uv[:, 0], uv[:, 1], uv[:, 2]
    textured = rng.uniform(size=n) < 0.7
    meta["diffuse_index"][textured] = 0
    meta["metallic_index"][textured & (rng.uniform(size=n) < 0.5)] = 1
    meta["roughness_index"][textured & (rng.uniform(size=n) < 0.5)] = 2
    nm = textured & (rng.uniform(size=n) < 0.5)
    meta["normal_index"][nm] = 3
    t = rng.standard_normal((n, 3)).astype(np.float32)
    meta["tangent1"] = meta["tangent2"] = meta["tangent3"] = t
    meta["bitangent1"] = meta["bitangent2"] = meta["bitangent3"] = np.cross(meta["normal1"], t).astype(np.float32)
    import dataclasses
    sc2 = dataclasses.replace(sc, meta=meta, materials=mats, sky=S.analytic_sky(64, 32))
    w, h = 256, 144
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), shading_metallic=0.8, shading_roughness=0.5)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc2, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 4)
    rt.kernel_raytrace(5)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    ref = O.kernel_raytrace(sc2, p, 5, w, h, denoise_available=1, max_depth=4)
    for img, r in zip((rt.temp_color_img1, rt.albedo_img1, rt.normal_img1), ref):
        assert rel_rmse(img.read(), r) <= RADIANCE_TOL


def test_full_size_culled_traversal_equals_the_reference_walk():
    """BASELINE config 2 at full size (871k-triangle stand-in, 1920x1080): a size-independent
    property instead of the oracle — production traversal == reference-order exhaustive walk, for
    the primary-hit ids and for the radiance of a 5-segment path (every bounce, shadow and MIS ray)."""
    sc = S.dragon_standin()
    w, h = 1920, 1080
    ids = [gpu_primary_ids(sc, S.DRAGON_CAMERA, w, h, 1, traversal=t) for t in (0, 1)]
    assert (ids[0] == ids[1]).all()
    assert (ids[0] >= 0).mean() > 0.5
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.1)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    out = []
    for t in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, t)
        rt.kernel_raytrace(1)
        out.append(rt.temp_color_img1.read())
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    assert (out[0] == out[1]).all()
    # and a band of rows against the oracle
    O = oracle()
    ref = O.kernel_raytrace(sc, p, 1, w, h, max_depth=5, rows=(500, 532))[0]
    assert rel_rmse(out[0][500:532], ref[500:532]) <= RADIANCE_TOL


def test_packet_kernel_keeps_lane_masks_above_lane_31():
    """Regression (found by scripts/gpu_soak.py): the camera-ray packet kernel rebuilt a popped 64-bit
    lane mask from two readfirstlane results, and a low word with lane 31 set sign-extended into lanes
    32-63; those lanes then tested a popped LEAF without having passed its box, and a triangle whose
    Woop matrix is near-singular accepted a hit there.  The fixture is the minimal failing packet:
    33 camera rays of the 871 k-triangle stand-in, the victim in lane 32."""
    g = golden("packet_mask_regression.npz")
    pt, df = g["point_tmax"], g["dir_flags"]
    sc = S.dragon_standin()
    walk = _trace(sc.triangles, sc.bvh, pt, df, traversal=1)
    ref = oracle().trace_rays(sc, pt, df).view(np.uint32)
    assert (walk == ref).all()
    for kernel in (12, 0, 11):  # 12 = the packet kernel for every launch
        got = _trace(sc.triangles, sc.bvh, pt, df, kernel=kernel)
        assert (got == ref).all(), kernel


def test_incoherent_ray_batch_on_a_large_soup():
    """BASELINE config 4 in miniature (2 M triangles, 4 M rays): culled == exhaustive, bitwise."""
    sc = S.build_scene(S.soup(2_000_000, seed=4))
    n = 4_000_000
    pt, df = random_rays(n, seed=5, lo=-1.0, hi=1.0)
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    res = []
    for t in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, t)
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        res.append(hb.read(np.uint32).reshape(n, 4))
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    assert (res[0] == res[1]).all()
    O = oracle()
    ref = O.trace_rays(sc, pt[:100000], df[:100000]).view(np.uint32)
    assert (res[0][:100000] == ref).all()


def _trace(sc_tris, sc_bvh, pt, df, traversal=0, kernel=0):
    n = pt.shape[0]
    tb, bb = capi.Buffer(sc_tris), capi.Buffer(sc_bvh)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    capi.set_option(capi.OPT_TRAVERSAL, traversal)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
    capi.trace_rays(tb, bb, ro, rd, hb, n)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
    return hb.read(np.uint32).reshape(n, 4)


def _irregular_rays(n, seed):
    """Random rays of which many have exactly-zero direction components (both signs of zero) and
    origins at 0: their 1/d is infinite and -o/d is NaN, the case the four-wide tree is not exact
    for (traverse.hpp) and the reference's min/max NaN behaviour decides (kernels/types.hpp:19-20)."""
    pt, df = random_rays(n, seed=seed, anyhit_every=7, tmax_every=5)
    d = df[:, :3].copy()
    d[::11, 0] = 0.0
    d[::13, 1] = -0.0
    d[::17, 2] = 0.0
    d[::11 * 13, :2] = 0.0
    norm = np.linalg.norm(d, axis=1, keepdims=True)
    norm[norm == 0] = 1
    df[:, :3] = d / norm
    df[::11, 0] = 0.0
    df[::13, 1] = -0.0
    pt[::19, :3] = 0.0
    return pt, df


@pytest.mark.parametrize("fixture", ["cornell", "knot_small", "soup50k"])
@pytest.mark.parametrize("kernel", [0, 6, 11, 12])  # 6 / 11 = batch kernels (four-wide / box-pair tree), 12 = packet kernel
def test_axis_parallel_rays_take_the_exact_path(fixture, kernel, request):
    sc = request.getfixturevalue(fixture)
    pt, df = _irregular_rays(100000, seed=31)
    walk = _trace(sc.triangles, sc.bvh, pt, df, traversal=1)
    got = _trace(sc.triangles, sc.bvh, pt, df, kernel=kernel)
    ref = oracle().trace_rays(sc, pt, df).view(np.uint32)
    anyhit = (df[:, 3].view(np.uint32) & 1) == 1
    assert (walk[~anyhit] == ref[~anyhit]).all()
    assert (got[~anyhit] == ref[~anyhit]).all()
    assert ((got[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()


def test_tree_with_boxes_that_are_not_nested_is_walked_like_the_reference(soup50k):
    """The four-wide tree is only built when every child box lies inside its parent's.  A tree
    that violates it (here: some inner boxes shrunk) must still give the reference's answer,
    which then depends on every ancestor box."""
    O = oracle()
    bvh = soup50k.bvh.copy()
    inner = np.flatnonzero(bvh["bottom_num_right"][:, 3] >= 0)
    rng = np.random.default_rng(3)
    pick = rng.choice(inner[1:], 200, replace=False)
    mid = 0.5 * (bvh["top_offset_left"][pick, :3] + bvh["bottom_num_right"][pick, :3])
    bvh["top_offset_left"][pick, :3] = mid + 0.25 * (bvh["top_offset_left"][pick, :3] - mid)
    bvh["bottom_num_right"][pick, :3] = mid + 0.25 * (bvh["bottom_num_right"][pick, :3] - mid)
    import dataclasses
    sc = dataclasses.replace(soup50k, bvh=bvh)
    pt, df = random_rays(200000, seed=8)
    ref = O.trace_rays(sc, pt, df).view(np.uint32)
    base = O.trace_rays(soup50k, pt, df).view(np.uint32)
    assert (ref != base).any()  # the shrunk boxes do hide triangles from the reference walk
    for kernel in (0, 6):
        got = _trace(sc.triangles, bvh, pt, df, kernel=kernel)
        assert (got == ref).all()


def test_tree_with_a_box_whose_corners_are_swapped_is_walked_like_the_reference(soup50k):
    """intersects_aabb takes min / max of the two bounds of an axis (kernels/intersection.hpp:40-52), so a box
    whose `top` is below its `bottom` on an axis is the same box to the reference.  The four-wide node step reads
    the nearer plane by the sign of 1/d instead, which is the same value only when bottom <= top: such a tree
    must not be collapsed (Packer::ordered) and still gives the reference's hits."""
    O = oracle()
    bvh = soup50k.bvh.copy()
    rng = np.random.default_rng(5)
    pick = rng.choice(np.arange(1, bvh.shape[0]), 300, replace=False)
    axis = rng.integers(0, 3, pick.size)
    t = bvh["top_offset_left"][pick, axis].copy()
    bvh["top_offset_left"][pick, axis] = bvh["bottom_num_right"][pick, axis]
    bvh["bottom_num_right"][pick, axis] = t
    import dataclasses
    sc = dataclasses.replace(soup50k, bvh=bvh)
    pt, df = random_rays(200000, seed=9)
    ref = O.trace_rays(sc, pt, df).view(np.uint32)
    base = O.trace_rays(soup50k, pt, df).view(np.uint32)
    assert (ref == base).all()  # the same boxes to the reference
    for kernel in (0, 6):
        got = _trace(sc.triangles, bvh, pt, df, kernel=kernel)
        assert (got == ref).all()


def test_overlapping_boxes_fill_the_stack_beyond_its_lds_part():
    """Large triangles all over the scene box: nearly every child box of every node is pierced, so a lane holds three
    more entries per level and runs past the 16 LDS rows of its stack into the spill area (HybridStack::push_upto3 /
    pop_lds: the row that takes those writes, the copy in global memory).  Checked once with a printf in that branch:
    14 419 pushes into the spill area on this batch, stack pointers up to 20."""
    sc = S.build_scene(S.soup(30000, seed=7, edge=0.9))
    pt, df = random_rays(60000, seed=12)
    walk = _trace(sc.triangles, sc.bvh, pt, df, traversal=1)
    got = _trace(sc.triangles, sc.bvh, pt, df)
    ref = oracle().trace_rays(sc, pt, df).view(np.uint32)
    anyhit = (df[:, 3].view(np.uint32) & 1) == 1
    assert (walk[~anyhit] == ref[~anyhit]).all()
    assert (got[~anyhit] == ref[~anyhit]).all()
    assert ((got[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()
    # the stack really is that deep: work counters of a culled closest-hit batch on this scene
    capi.set_option(capi.OPT_COUNT_WORK, 1)
    try:
        capi.reset_counters()
        _trace(sc.triangles, sc.bvh, pt, df)
        k = capi.counters()
    finally:
        capi.set_option(capi.OPT_COUNT_WORK, 0)
    assert k["node_visits"] / 2 / len(pt) > 60  # four-wide node steps per ray: far more than a surface mesh needs


@pytest.mark.parametrize("flags", [0x100, 0x1, 0x2, 0x3, 0x5, 0xD])
def test_ray_queue_order_does_not_change_results(flags, knot_small):
    """The ray queue may hand rays to waves in any order (sub-queue mapping per XCD, the three rays of a
    path adjacent or not: traverse.hpp RayQueue): every frame must be the same bits."""
    w, h = 256, 192
    cam = S.Camera(width=w, height=h, **S.DRAGON_CAMERA)
    p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
    rt = Raytracer(0)
    rt.set_scene(knot_small, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 4)
    rt.kernel_raytrace(7)
    base = rt.temp_color_img1.read().copy()
    capi.set_option(capi.OPT_TUNE, flags << 16)
    try:
        rt.kernel_raytrace(7)
        got = rt.temp_color_img1.read()
    finally:
        capi.set_option(capi.OPT_TUNE, 0)
        capi.set_option(capi.OPT_MAX_DEPTH, 0)
    assert (got.view(np.uint32) == base.view(np.uint32)).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 5000])
def test_small_batches_run_in_drain_mode_from_the_start(n, knot_small):
    """With fewer rays than lanes in the grid the ray queue is empty at once and the persistent
    kernel splits every running ray among idle lanes (k_traverse_p, FORK): closest hits, bounded rays
    and any-hit answers must still be the reference's."""
    sc = knot_small
    pt, df = random_rays(n, seed=100 + n, anyhit_every=5, tmax_every=3)
    got = _trace(sc.triangles, sc.bvh, pt, df)
    ref = oracle().trace_rays(sc, pt, df).view(np.uint32)
    anyhit = (df[:, 3].view(np.uint32) & 1) == 1
    assert (got[~anyhit] == ref[~anyhit]).all()
    assert ((got[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()


def test_tiles_partition_is_bit_identical_to_one_gpu(knot_small):
    """Multi-GPU rule on one GPU: render the frame as 3 ranks' tile sets, one after the other, into
    the same image; the union must equal the single-rank frame bit for bit."""
    w, h = 300, 200
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA))
    rt = Raytracer(0)
    rt.set_scene(knot_small, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 3)
    rt.kernel_raytrace(9)
    whole = rt.temp_color_img1.read()
    rt.temp_color_img1.fill([0, 0, 0, 0])
    capi.set_option(capi.OPT_TILE_RANKS, 3)
    for r in range(3):
        capi.set_option(capi.OPT_TILE_RANK, r)
        rt.kernel_raytrace(9)
    capi.set_option(capi.OPT_TILE_RANK, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    assert (rt.temp_color_img1.read() == whole).all()


def test_scene_edge_cases():
    O = oracle()
    # one triangle: the BVH root is a leaf
    tri = np.float32([[-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.6, 0]])
    sc = S.build_scene(tri)
    assert sc.bvh.shape[0] == 1
    cam = dict(position=[0, 0, 2], target=[0, 0, 0], up=[0, 1, 0], fovy=45.0)
    ids = gpu_primary_ids(sc, cam, 64, 64, 1)
    meta, sky = id_inputs(sc)
    p = S.make_scene_params(S.Camera(width=64, height=64, **cam), path_tracing=False)
    ref = S.decode_ids(O.kernel_raytrace(sc, p, 1, 64, 64, meta=meta, sky=sky)[0]).reshape(64, 64)
    assert (ids == ref).all() and (ids == 0).any() and (ids == -1).any()
    # a leaf above the 128-triangle chunk limit (300 coincident-centre triangles: no SAH split pays)
    rng = np.random.default_rng(0)
    many = (rng.uniform(-0.3, 0.3, (300, 3, 3)) * [1, 1, 0.001]).reshape(300, 9).astype(np.float32)
    sc = S.build_scene(many)
    ids = gpu_primary_ids(sc, cam, 64, 64, 1)
    meta, sky = id_inputs(sc)
    ref = S.decode_ids(O.kernel_raytrace(sc, p, 1, 64, 64, meta=meta, sky=sky)[0]).reshape(64, 64)
    assert (ids == ref).all()


def test_error_behaviour():
    # empty inputs are rejected like Accelerator::create_* do (cuda/accelerator/accelerator.hpp:43-45)
    with pytest.raises(ValueError):
        capi.Buffer(np.zeros(0, np.float32))
    with pytest.raises(ValueError):
        capi.Image(0, 4)
    # a foreign device pointer surfaces as an error code, not a fault
    sc = S.cornell_box()
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    lib = capi.load()
    rc = lib.nova_hip_trace_rays(tb.ptr, bb.ptr, 0x1000, 0x2000, 0x3000, 16)
    assert rc == -2 and b"not allocated" in lib.nova_hip_error_string(rc)
    # a malformed BVH (right child pointing backwards) is refused
    bad = sc.bvh.copy()
    bad["bottom_num_right"][0, 3] = 0.0
    bb2 = capi.Buffer(bad)
    ro, rd, hb = capi.Buffer(np.zeros((4, 4), np.float32)), capi.Buffer(np.zeros((4, 4), np.float32)), capi.Buffer(nbytes=64)
    assert lib.nova_hip_trace_rays(tb.ptr, bb2.ptr, ro.ptr, rd.ptr, hb.ptr, 4) == -3
    # rewriting a buffer invalidates the cached scene: results follow the new contents
    bb3 = capi.Buffer(sc.bvh)
    pt, df = random_rays(1000, seed=1)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16000)
    capi.trace_rays(tb, bb3, ro, rd, hb, 1000)
    first = hb.read(np.uint32).copy()
    moved = sc.bvh.copy()
    moved["top_offset_left"][:, :3] += 100.0
    moved["bottom_num_right"][:, :3] += 100.0
    bb3.write(moved)
    capi.trace_rays(tb, bb3, ro, rd, hb, 1000)
    assert (hb.read(np.int32).reshape(-1, 4)[:, 3] == -1).all() and (first.view(np.int32).reshape(-1, 4)[:, 3] >= 0).any()


def test_random_configurations_against_the_oracle_bitwise():
    """A short run of scripts/gpu_soak_oracle.py (the long runs are logged in profiles/): the whole
    per-sample schedule on random scenes, cameras, odd frame sizes, depth caps, shading defaults,
    textures, feature images, FXAA, accumulated samples and tile partitions, GPU == oracle bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_soak_oracle.py"), "40", "200"],
                         cwd=root, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "SOAK OK" in res.stdout


def test_clamp_to_edge_option_against_the_oracle():
    """NOVA_HIP_OPT_CLAMP_TO_EDGE: CLAMP as the CUDA backend's clamp-to-edge instead of the OpenCL
    backend's border colour 0.  Only FXAA's taps at the outermost pixel ring differ; GPU == oracle in both modes,
    the ring included, and the two modes agree far from the border."""
    from oracle import bind as O
    rng = np.random.default_rng(21)
    w, h = 112, 96
    img = rng.uniform(0.0, 2.0, (h, w, 4)).astype(np.float32)
    img[:, : w // 2] *= 0.1  # a strong vertical edge that runs into the top and bottom borders
    cam = S.Camera(width=w, height=h, **S.CORNELL_CAMERA)
    params = S.make_scene_params(cam, anti_aliasing=True, exposure=1.2)
    src = capi.Image(w, h, data=img)
    dst = capi.Image(w, h, texel_bytes=4)
    out = {}
    try:
        for mode in (0, 1):
            capi.set_option(capi.OPT_CLAMP_TO_EDGE, mode)
            O.set_clamp_to_edge(bool(mode))
            capi.launch_post_process(params, src, dst, w, h)
            out[mode] = dst.read().copy()
            assert (out[mode] == O.kernel_post_process(params, img)).all()
    finally:
        capi.set_option(capi.OPT_CLAMP_TO_EDGE, 0)
        O.set_clamp_to_edge(False)
    # FXAA searches up to ~26 pixels along an edge (kernels/anti_aliasing.hpp:99-116): farther in than that the
    # border cannot be seen
    assert (out[0][32:-32, 32:-32] == out[1][32:-32, 32:-32]).all()
    assert (out[0] != out[1]).any()  # nearer to it, it can
