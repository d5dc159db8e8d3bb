"""GPU (-m gpu): NOVA_HIP_OPT_MIS_PRETEST (k_shade's light pre-test of MIS rays, default on) through the C ABI.
Frames with the pre-test are the frames without it and the oracle's, bit for bit; the counters say how many MIS rays
went untraced, and agree with the oracle's restatement of the rule; scenes with more light triangles than the list
holds trace every MIS ray; a rewritten meta buffer gets a new list."""
import numpy as np
import pytest

from nova_amd import capi, scene as S
from nova_amd.raytracer import Raytracer
from tests.helpers import SOUP_CAMERA, lit_soup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    capi.init(0)
    yield
    for opt in (capi.OPT_MAX_DEPTH, capi.OPT_TRAVERSAL, capi.OPT_TRAVERSE_KERNEL):
        capi.set_option(opt, 0)
    capi.set_option(capi.OPT_MIS_PRETEST, 1)


def oracle():
    from oracle import bind as O
    return O


def gpu_frame(rt, seed, pretest, depth, kernel=0, traversal=0):
    capi.set_option(capi.OPT_MIS_PRETEST, pretest)
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
    capi.set_option(capi.OPT_TRAVERSAL, traversal)
    capi.reset_counters()
    rt.kernel_raytrace(seed)
    k = capi.counters()
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_MIS_PRETEST, 1)
    return [im.read().view(np.uint32) for im in (rt.temp_color_img1, rt.albedo_img1, rt.normal_img1)], k


def scene_of(name, request):
    if name == "cornell":
        return request.getfixturevalue("cornell"), S.CORNELL_CAMERA
    if name == "knot":
        return request.getfixturevalue("knot_small"), S.DRAGON_CAMERA
    return lit_soup(n_emissive=int(name.rsplit("_", 1)[1])), SOUP_CAMERA


@pytest.mark.parametrize("name,depth", [("cornell", 5), ("knot", 0), ("lit_soup_6", 4), ("lit_soup_16", 4), ("lit_soup_17", 4),
                                        ("lit_soup_40", 0)])
def test_frames_with_and_without_the_pretest_and_the_oracle(name, depth, request):
    sc, cam = scene_of(name, request)
    O = oracle()
    w, h = 200, 136  # ragged tiles
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.3, shading_roughness=0.4)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc, p, w, h)
    n_light_tris = int((sc.meta["light_index"] != -1).sum())
    for seed in (5, 6):
        on, k_on = gpu_frame(rt, seed, 1, depth)
        off, k_off = gpu_frame(rt, seed, 0, depth)
        cnt = np.zeros(8, np.uint64)
        ref = O.kernel_raytrace(sc, p, seed, w, h, denoise_available=1, max_depth=depth, counters=cnt, mis_pretest=True)
        for a, b, r in zip(on, off, ref):
            assert (a == b).all() and (a == r.view(np.uint32)).all()
        for k in (k_on, k_off):  # the schedule's calls are counted alike
            assert (k["rays_primary"], k["rays_bounce"], k["rays_shadow"], k["path_segments"]) == tuple(int(x) for x in cnt[:4])
        assert k_off["rays_mis_pretested"] == 0 and k_off["rays_shadow_skipped"] == 0
        assert k_on["rays_shadow_skipped"] == int(cnt[7]) > 0
        if n_light_tris <= 16:
            assert k_on["rays_mis_pretested"] == int(cnt[6]) > 0
        else:  # more light triangles than the list holds: every MIS ray is traced
            assert k_on["rays_mis_pretested"] == 0


@pytest.mark.parametrize("kernel,traversal", [(6, 0), (11, 0), (12, 0), (0, 1)])
def test_the_validation_kernels_read_the_compacted_mis_rays_too(kernel, traversal, request):
    sc, cam = scene_of("lit_soup_6", request)
    w, h = 160, 96
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.3, shading_roughness=0.4)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc, p, w, h)
    base, k0 = gpu_frame(rt, 9, 1, 4)
    other, k1 = gpu_frame(rt, 9, 1, 4, kernel=kernel, traversal=traversal)
    for a, b in zip(base, other):
        assert (a == b).all()
    assert k0["rays_mis_pretested"] == k1["rays_mis_pretested"] > 0
    assert k0["rays_shadow_skipped"] == k1["rays_shadow_skipped"] > 0


def test_a_rewritten_meta_buffer_gets_a_new_list(request):
    """Same device pointers, new contents: the light triangles stop being light triangles (the lights stay in the light
    array and are still sampled; only rays that HIT those triangles see the change).  A stale list would keep tracing
    the MIS rays that pass them."""
    sc, cam = scene_of("lit_soup_6", request)
    O = oracle()
    w, h = 160, 96
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.3, shading_roughness=0.4)
    rt = Raytracer(0, denoise_available=True)
    rt.set_scene(sc, p, w, h)
    first, k_first = gpu_frame(rt, 9, 1, 4)
    meta = sc.meta.copy()
    meta["light_index"] = -1
    rt.set_meta(meta)
    got, k = gpu_frame(rt, 9, 1, 4)
    cnt = np.zeros(8, np.uint64)
    ref = O.kernel_raytrace(sc, p, 9, w, h, denoise_available=1, max_depth=4, meta=meta, counters=cnt, mis_pretest=True)
    assert (got[0] == ref[0].view(np.uint32)).all()
    assert k["rays_mis_pretested"] == int(cnt[6]) == int(cnt[2]) > k_first["rays_mis_pretested"]  # every MIS ray now
    assert k["rays_bounce"] == int(cnt[1])


def test_render_samples_with_the_pretest(request):
    sc, cam = scene_of("knot", request)
    w, h = 200, 136
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.4, shading_roughness=0.1)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    capi.set_option(capi.OPT_MAX_DEPTH, 5)
    frames = []
    for pretest in (1, 0):
        capi.set_option(capi.OPT_MIS_PRETEST, pretest)
        rt.start()
        frames.append(rt.raytrace_samples([20 + i for i in range(7)]))
        frames.append(rt.temp_color_img2.read().view(np.uint32))
    capi.set_option(capi.OPT_MIS_PRETEST, 1)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    assert (frames[0] == frames[2]).all() and (frames[1] == frames[3]).all()
