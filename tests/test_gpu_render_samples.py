"""GPU (-m gpu): nova_hip_render_samples — several samples of
Raytracer::raytrace's device schedule (src/core/raytracer.cpp:169-262) as one enqueue with one host
wait — against the four-call schedule through the same C ABI, and against the CPU oracle.
Bar: every image byte-identical (same kernels, same per-pixel order; only host waits, the
per-sample copies and the intermediate post-process passes are gone)."""
import numpy as np
import pytest

from nova_amd import capi, scene as S
from nova_amd.raytracer import Raytracer
from tests.helpers import SOUP_CAMERA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    capi.init(0)
    yield
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    capi.set_option(capi.OPT_TILE_RANK, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)


def images_of(rt):
    names = ["temp_color_img1", "temp_color_img2", "prev_color_img", "color_img"]
    if rt.denoise_available:
        names += ["albedo_img1", "normal_img1", "albedo_img2", "normal_img2", "prev_albedo_img", "prev_normal_img"]
    return {n: getattr(rt, n).read() for n in names}


def same_bytes(a, b):
    return a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all()


def make(sc, p, w, h, denoise):
    rt = Raytracer(0, denoise_available=denoise)
    rt.set_scene(sc, p, w, h)
    rt.start()
    if denoise:  # the reference never clears them (the first accumulate multiplies them by 0)
        rt.prev_albedo_img.fill([0.0] * 4)
        rt.prev_normal_img.fill([0.0] * 4)
    return rt


@pytest.mark.parametrize("fixture,cam,w,h,depth,denoise,fxaa,n", [
    ("cornell", S.CORNELL_CAMERA, 128, 128, 5, True, False, 5),   # odd count: final mean in temp2's plane
    ("cornell", S.CORNELL_CAMERA, 96, 72, 0, False, True, 4),     # unbounded paths in a closed box (dozens of segments)
    ("knot_small", S.DRAGON_CAMERA, 320, 200, 5, True, True, 6),  # ragged tiles, even count
    ("soup50k", SOUP_CAMERA, 200, 130, 3, False, False, 1),       # a single sample
    ("knot_small", S.DRAGON_CAMERA, 256, 160, 0, True, False, 7),  # unbounded: long paths finished when the lane comes round
])
def test_fused_samples_leave_the_four_call_schedules_images(fixture, cam, w, h, depth, denoise, fxaa, n, request):
    sc = request.getfixturevalue(fixture)
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), path_tracing=True, anti_aliasing=fxaa,
                            shading_metallic=0.3, shading_roughness=0.2)
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    times = [7 + 3 * k for k in range(n)]
    a = make(sc, p, w, h, denoise)
    for t in times:
        pix_a = a.raytrace(t)
    b = make(sc, p, w, h, denoise)
    pix_b = b.raytrace_samples(times)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    assert b.sample_index == a.sample_index == n
    assert same_bytes(pix_a, pix_b)
    ia, ib = images_of(a), images_of(b)
    for name in ia:
        assert same_bytes(ia[name], ib[name]), name
    assert same_bytes(ib["temp_color_img2"], ib["prev_color_img"])
    assert np.abs(ib["temp_color_img2"][..., :3]).sum() > 0


def test_fused_samples_continue_a_running_mean(cornell):
    """2 samples by the four-call schedule, 3 fused, 1 more by four calls == 6 by four calls."""
    w = h = 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 4)
    a = make(cornell, p, w, h, True)
    for k in range(6):
        a.raytrace(100 + k)
    b = make(cornell, p, w, h, True)
    b.raytrace(100)
    b.raytrace(101)
    b.raytrace_samples([102, 103, 104])
    b.raytrace(105)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    ia, ib = images_of(a), images_of(b)
    for name in ia:
        assert same_bytes(ia[name], ib[name]), name


def test_fused_calls_of_different_lengths_in_a_row(knot_small):
    """The library rotates consecutive samples over its lanes (own stream and path state each) and the
    number of lanes in use follows the call's length: 3, 2, 1, 4 and 7 samples on one frame == 17."""
    w, h = 192, 128
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), path_tracing=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    a = make(knot_small, p, w, h, True)
    for k in range(17):
        a.raytrace(50 + k)
    b = make(knot_small, p, w, h, True)
    k = 0
    for n in (3, 2, 1, 4, 7):
        b.raytrace_samples([50 + k + i for i in range(n)])
        k += n
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    ia, ib = images_of(a), images_of(b)
    for name in ia:
        assert same_bytes(ia[name], ib[name]), name


def test_fused_samples_keep_the_ray_counters(knot_small):
    w, h = 192, 128
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), path_tracing=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    a = make(knot_small, p, w, h, False)
    capi.reset_counters()
    for k in range(5):
        a.raytrace(9 + k)
    ka = capi.counters()
    capi.reset_counters()
    a.start()
    a.raytrace_samples([9 + k for k in range(5)])
    kb = capi.counters()
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    for name in ("rays_primary", "rays_bounce", "rays_shadow", "path_segments", "traverse_launches"):
        assert ka[name] == kb[name], name


def test_fused_samples_against_the_oracle(cornell):
    from oracle import bind as O
    w = h = 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    rt = make(cornell, p, w, h, False)
    pixels = rt.raytrace_samples([1, 2, 3])
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    z = np.zeros((h, w, 4), np.float32)
    acc = z
    for k, t in enumerate([1, 2, 3]):
        c = O.kernel_raytrace(cornell, p, t, w, h, max_depth=5)[0]
        acc = O.kernel_accumulate(k, 0, c, z, z, acc, z, z)[0]
    assert same_bytes(rt.temp_color_img2.read(), acc)
    assert (pixels == O.kernel_post_process(p, acc)).all()


def test_fused_samples_under_a_tile_partition(knot_small):
    """Multi-GPU unit: a rank's tiles rendered by the fused call == the same tiles of the whole frame."""
    w, h = 256, 192
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), path_tracing=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 4)
    whole = make(knot_small, p, w, h, False)
    whole.raytrace_samples([11, 12, 13])
    ref = whole.temp_color_img2.read()
    capi.set_option(capi.OPT_TILE_RANKS, 3)
    capi.set_option(capi.OPT_TILE_RANK, 1)
    part = make(knot_small, p, w, h, False)
    part.temp_color_img1.fill([0.25, 0.5, 0.75, 1.0])  # what other ranks' pixels hold is the caller's business,
    part.raytrace_samples([11, 12, 13])                # but it must come out as the four-call schedule leaves it
    calls = make(knot_small, p, w, h, False)
    calls.temp_color_img1.fill([0.25, 0.5, 0.75, 1.0])
    for t in (11, 12, 13):
        calls.raytrace(t)
    ip, ic = images_of(part), images_of(calls)
    for name in ip:
        assert same_bytes(ip[name], ic[name]), name
    capi.set_option(capi.OPT_TILE_RANK, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    got = part.temp_color_img2.read()
    tiles_x = (w + 63) // 64
    mine = 0
    for ty in range((h + 63) // 64):
        for tx in range(tiles_x):
            if (ty * tiles_x + tx) % 3 == 1:
                sl = (slice(ty * 64, min(h, ty * 64 + 64)), slice(tx * 64, min(w, tx * 64 + 64)))
                assert same_bytes(np.ascontiguousarray(got[sl]), np.ascontiguousarray(ref[sl]))
                mine += 1
    assert mine == 4


def test_fused_samples_error_behaviour(cornell):
    w = h = 32
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
    rt = make(cornell, p, w, h, False)
    with pytest.raises(capi.NovaHipError) as e:  # no samples
        rt.raytrace_samples([])
    assert e.value.err() == capi.E_INVALID_ARGUMENT
    rt.denoise_available = 1  # feature images asked for but absent
    with pytest.raises(capi.NovaHipError) as e:
        rt.raytrace_samples([1])
    assert e.value.err() == capi.E_INVALID_ARGUMENT
    rt.denoise_available = 0
    rt.sample_index = 2**31 - 2  # sample index would overflow int32
    with pytest.raises(capi.NovaHipError) as e:
        rt.raytrace_samples([1, 2, 3])
    assert e.value.err() == capi.E_TOO_LARGE
    rt.sample_index = 0
    small = capi.Image(16, 16, texel_bytes=4)  # pixels image of the wrong size
    imgs = rt.frame_images()
    imgs.pixels = small.handle
    with pytest.raises(capi.NovaHipError) as e:
        capi.render_samples(rt.params, [1], 0, imgs, w, h, rt.triangle_buf, rt.tri_meta_buf, rt.bvh_buf,
                            rt.light_buf, rt.num_lights, rt.material_imgs, rt.sky_img, 0)
    assert e.value.err() == capi.E_INVALID_ARGUMENT
    assert rt.raytrace_samples([1]).shape == (h, w, 4)  # and the object is still usable
