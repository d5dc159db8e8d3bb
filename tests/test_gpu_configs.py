"""GPU (-m gpu): the BASELINE.json configurations that are not bench lines, at FULL size, through the C ABI.
  C3  fireplace_room stand-in, 3840x2160: production traversal == reference walk on every pixel; a 3-rank tile
      partition assembles the one-GPU frame; a band of rows against the CPU oracle.
  C4  10 M-triangle soup built on the device, 16.7 M incoherent closest-hit rays in one batch: the first 1 M rays
      production == reference walk, the first 100 k against the CPU oracle, every hit record bit for bit.
  C5  assets/aircraft (the committed data fixture: real mesh, UVs and the two JPEGs of its MTL that the reference
      tree holds) with textures, normal maps and a sky, 3840x2160: production == reference walk; a 256x144 frame
      with feature images against the oracle.
Sizes the oracle cannot finish in seconds are covered by the size-independent property (the reference-order,
un-culled walk of kernels/raytrace.hpp:38-102 run on the GPU), smaller samples of the same inputs by the oracle."""
import types

import numpy as np
import pytest

from nova_amd import capi, scene as S
from nova_amd.raytracer import Raytracer
from tests.conftest import random_rays, rel_rmse

pytestmark = pytest.mark.gpu
RADIANCE_TOL = 1e-3  # north_star; in practice the planes are bit-identical


@pytest.fixture(autouse=True)
def options():
    capi.init(0)
    yield
    for opt in (capi.OPT_MAX_DEPTH, capi.OPT_TRAVERSAL, capi.OPT_TILE_RANK):
        capi.set_option(opt, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)


def oracle():
    from oracle import bind as O
    return O


def production_and_walk(rt, time, depth):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    out = []
    for trav in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, trav)
        rt.kernel_raytrace(time)
        out.append(rt.temp_color_img1.read())
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    return out


def test_c3_room_4k_walk_tiles_and_oracle_band():
    sc = S.room_standin()
    assert 130_000 < sc.triangles.shape[0] < 160_000 and sc.num_lights >= 1
    w, h = 3840, 2160
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.ROOM_CAMERA), **S.ROOM_SHADING)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    prod, walk = production_and_walk(rt, 7, 5)
    assert (prod.view(np.uint32) == walk.view(np.uint32)).all()
    assert (prod[..., :3] > 0).any(-1).mean() > 0.4  # a closed, lit room (1 sample: many paths end unlit)
    rows = (1000, 1012)
    ref = oracle().kernel_raytrace(sc, p, 7, w, h, max_depth=5, rows=rows)[0]
    assert rel_rmse(prod[rows[0]:rows[1]], ref[rows[0]:rows[1]]) <= RADIANCE_TOL
    assert (prod[rows[0]:rows[1]].view(np.uint32) == ref[rows[0]:rows[1]].view(np.uint32)).all(-1).mean() > 0.999
    # two accumulated samples, whole frame on one GPU vs three ranks with their own images
    rt.start()
    for t in (7, 8):
        whole = rt.raytrace(t)
    ranks = 3
    nbytes = capi.tiles_stack_bytes(w, h, ranks, 4)
    stacks = capi.Buffer(nbytes=ranks * nbytes)
    root = None
    for r in range(ranks):
        capi.set_option(capi.OPT_TILE_RANKS, ranks)
        capi.set_option(capi.OPT_TILE_RANK, r)
        part = Raytracer(0)
        part.set_scene(sc, p, w, h)
        part.start()
        for t in (7, 8):
            part.raytrace(t, read_back=False)
        capi.tiles_pack(part.color_img, r, ranks, stacks.ptr + r * nbytes)
        root = part if r == 0 else root
        if r:
            del part
    capi.tiles_unpack(root.color_img, ranks, stacks.ptr)
    assert (root.color_img.read() == whole).all()


def test_c4_ten_million_triangle_soup_sixteen_million_rays():
    n_tri, n_rays = 10_000_000, 1 << 24
    verts = S.soup(n_tri, seed=4)
    vb = capi.Buffer(verts)
    bvh, tris, order, n_nodes, stats = capi.build_bvh_device(vb, n_tri)  # BVH::build on the device
    assert 3_000_000 < n_nodes < 4_500_000
    vb.free()
    del verts
    pt, df = random_rays(n_rays, seed=5, lo=-1.0, hi=1.0)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n_rays)
    capi.trace_rays(tris, bvh, ro, rd, hb, n_rays)
    got = hb.read(np.uint32).reshape(n_rays, 4)
    hit = got[:, 3].view(np.int32) >= 0
    assert 0.5 < hit.mean() <= 1.0
    assert (got[:, 3].view(np.int32) < n_tri).all()
    m = 1 << 20
    capi.set_option(capi.OPT_TRAVERSAL, 1)
    hw = capi.Buffer(nbytes=16 * m)
    capi.trace_rays(tris, bvh, ro, rd, hw, m)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    assert (hw.read(np.uint32).reshape(m, 4) == got[:m]).all()
    k = 100_000
    host = types.SimpleNamespace(bvh=bvh.read(S.BVH_NODE, n_nodes), triangles=tris.read(S.TRIANGLE_DATA, n_tri))
    ref = oracle().trace_rays(host, pt[:k], df[:k]).view(np.uint32)
    assert (got[:k] == ref).all()


def test_c5_aircraft_textures_normal_maps_and_sky():
    sc = S.aircraft_scene()
    assert sc.triangles.shape[0] == 16220 and sc.materials.shape == (4, 256, 256, 4)
    assert (sc.meta["diffuse_index"] >= 0).all() and (sc.meta["normal_index"] >= 0).all()
    uv = sc.meta["texture_coord1"]
    assert np.ptp(uv[:, 0]) > 0.5 and np.ptp(uv[:, 1]) > 0.5  # the asset's own texture coordinates
    # 4K, whole shading path, size-independent property
    w, h = 3840, 2160
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.AIRCRAFT_CAMERA), shading_metallic=0.6, shading_roughness=0.3)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    prod, walk = production_and_walk(rt, 3, 5)
    assert (prod.view(np.uint32) == walk.view(np.uint32)).all()
    assert np.isfinite(prod).all()
    del rt
    # 256x144 with feature images against the oracle: texture look-ups (bilinear, REPEAT), TBN normal mapping, sky —
    # here with the sky texels the reference's own SkyLoader made (tests/golden/aircraft.npz, make_aircraft_fixture.py)
    sc = S.aircraft_scene(sky_size=None)
    assert sc.sky.shape == (64, 128, 4)
    w, h = 256, 144
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.AIRCRAFT_CAMERA), shading_metallic=0.6, shading_roughness=0.3)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    rt.kernel_raytrace(3)
    ref = oracle().kernel_raytrace(sc, p, 3, w, h, denoise_available=1, max_depth=5)
    hit_model = 0
    for img, r in zip((rt.temp_color_img1, rt.albedo_img1, rt.normal_img1), ref):
        got = img.read()
        assert rel_rmse(got, r) <= RADIANCE_TOL
        assert (got.view(np.uint32) == r.view(np.uint32)).all(-1).mean() > 0.999
        hit_model = max(hit_model, float((r[..., :3] != 0).any(-1).mean()))
    # the normal feature is zero on sky pixels and a unit vector on the aircraft: it must cover part of the frame
    n_ref = ref[2][..., :3]
    on_model = np.abs(np.linalg.norm(n_ref, axis=-1) - 1.0) < 1e-3
    assert 0.1 < on_model.mean() < 0.95


def test_c5_eight_bit_material_layers_give_the_same_frame_bit_for_bit():
    """SURVEY §8f rank 4 (texture path): the layers uploaded as uchar4 (a quarter of the bytes) and converted
    x / 255.0f at sampling time give exactly the frame of the reference's float4 array — and the oracle's.  Both
    arrays are the reference loader's own output (a window of MaterialLoader::build's float4 array and the uint8
    texels under it: tests/golden/aircraft.npz, tests/test_reference_pins.py), not stand-ins."""
    from tests.helpers import golden
    sc = S.aircraft_scene(sky_size=None)
    u8 = S.aircraft_layers_u8()
    assert (sc.materials == (u8.astype(np.float32) / np.float32(255.0))).all()
    f4 = golden("aircraft.npz")["window_f4"]
    assert (sc.materials[:, :64, :64].view(np.uint32) == f4.view(np.uint32)).all()  # = MaterialLoader::build's values
    w, h = 512, 288
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.AIRCRAFT_CAMERA), shading_metallic=0.6, shading_roughness=0.3)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    out = []
    for layers in (None, u8):
        rt = Raytracer(0, denoise_available=True)
        rt.set_scene(sc, p, w, h, materials_u8=layers)
        rt.kernel_raytrace(11)
        out.append([im.read() for im in (rt.temp_color_img1, rt.albedo_img1, rt.normal_img1)])
    for a, b in zip(*out):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    ref = oracle().kernel_raytrace(sc, p, 11, w, h, denoise_available=1, max_depth=5)
    for got, r in zip(out[1], ref):
        assert rel_rmse(got, r) <= RADIANCE_TOL
        assert (got.view(np.uint32) == r.view(np.uint32)).all(-1).mean() > 0.999
    # a uchar4 array of the wrong texel size is still refused elsewhere: the sky stays float4
    lib = capi.load()
    rt8 = Raytracer(0)
    rt8.set_scene(sc, p, w, h, materials_u8=u8)
    sky8 = capi.Image(8, 4, texel_bytes=4)
    rc = lib.nova_hip_launch_raytrace(np.ascontiguousarray(p).ctypes.data, 1, rt8.temp_color_img1.handle, w, h,
                                      rt8.triangle_buf.ptr, rt8.tri_meta_buf.ptr, rt8.bvh_buf.ptr, None, 0,
                                      rt8.material_imgs.handle, sky8.handle, 0, None, None)
    assert rc == capi.E_INVALID_ARGUMENT
