"""GPU (-m gpu): the multi-GPU rule on ONE GPU, as separate processes would run it.  Every "rank" is its own
Raytracer with its own images (a rank never sees another rank's pixels), the finished tiles travel through the
C ABI's tile stacks (nova_hip_tiles_pack / _unpack, the buffers of the one all_gather per final frame), and the
assembled frame must equal the frame one GPU renders alone, byte for byte — with FXAA on (float4 means gathered,
whole-frame post-process on the root) and off (uchar4 pixels gathered)."""
import numpy as np
import pytest

from nova_amd import capi, scene as S, tiles
from nova_amd.raytracer import Raytracer

pytestmark = pytest.mark.gpu
NAN4 = [float("nan")] * 4


@pytest.fixture(autouse=True)
def options():
    capi.init(0)
    yield
    capi.set_option(capi.OPT_TILE_RANK, 0)
    capi.set_option(capi.OPT_TILE_RANKS, 1)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)


def render_ranks(sc, params, w, h, ranks, times, poison=True, denoise=False):
    """Each rank renders its tiles with its own images; returns the per-rank Raytracers."""
    out = []
    for r in range(ranks):
        capi.set_option(capi.OPT_TILE_RANKS, ranks)
        capi.set_option(capi.OPT_TILE_RANK, r)
        rt = Raytracer(0, denoise_available=denoise)
        rt.set_scene(sc, params, w, h)
        rt.start()
        if poison:  # whatever other ranks' pixels hold in this rank's memory must not matter
            for im in (rt.temp_color_img1, rt.temp_color_img2, rt.prev_color_img):
                im.fill(NAN4)
        for t in times:
            rt.raytrace(t, read_back=False, post_process=not bool(params["anti_aliasing"]))
        out.append(rt)
    return out


def gather_on_root(rts, image_of, ranks):
    """What one all_gather + nova_hip_tiles_unpack leave in rank 0's image."""
    im0 = image_of(rts[0])
    nbytes = capi.tiles_stack_bytes(im0.width, im0.height, ranks, im0.texel_bytes)
    stacks = capi.Buffer(nbytes=ranks * nbytes)
    for r, rt in enumerate(rts):
        capi.tiles_pack(image_of(rt), r, ranks, stacks.ptr + r * nbytes)
    capi.tiles_unpack(im0, ranks, stacks.ptr)
    return im0


@pytest.mark.parametrize("w,h,ranks", [(300, 200, 3), (448, 256, 2), (1920, 1080, 8)])
@pytest.mark.parametrize("aa", [True, False])
def test_ranks_with_their_own_images_assemble_the_one_gpu_frame(w, h, ranks, aa, knot_small):
    cam = S.Camera(width=w, height=h, **S.DRAGON_CAMERA)
    p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1, anti_aliasing=aa)
    times = [5, 6, 7]
    capi.set_option(capi.OPT_MAX_DEPTH, 3)
    one = Raytracer(0)
    one.set_scene(knot_small, p, w, h)
    one.start()
    for t in times:
        one.raytrace(t, read_back=False)
    ref_mean, ref_pixels = one.temp_color_img2.read(), one.color_img.read()
    rts = render_ranks(knot_small, p, w, h, ranks, times)
    if aa:
        # a rank cannot post-process with FXAA: refused, not silently wrong
        with pytest.raises(capi.NovaHipError) as e:
            rts[0].kernel_post_process()
        assert e.value.err() == capi.E_INVALID_ARGUMENT
        gather_on_root(rts, lambda rt: rt.temp_color_img2, ranks)
        assert (rts[0].temp_color_img2.read().view(np.uint32) == ref_mean.view(np.uint32)).all()
        capi.set_option(capi.OPT_TILE_RANKS, 1)
        capi.set_option(capi.OPT_TILE_RANK, 0)
        rts[0].kernel_post_process()
        got = rts[0].color_img.read()
    else:
        got = gather_on_root(rts, lambda rt: rt.color_img, ranks).read()
    assert (got == ref_pixels).all()
    assert len(np.unique(ref_pixels[..., :3])) > 50


def test_halo_pixels_are_what_makes_the_running_mean_exact(knot_small):
    """300 and 200 are sizes at which the pixel-centre sample of some tile-edge columns / rows gives a
    neighbouring texel a weight of ~1e-6: the ranks' plans must hold extra pixels there, and only there."""
    w, h = 300, 200
    x = np.arange(w, dtype=np.float32)
    f = (x + np.float32(0.5)) / np.float32(w) * np.float32(w) - np.float32(0.5)
    assert ((np.floor(f) != x) | (f - np.floor(f) != 0)).any()
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA))
    capi.set_option(capi.OPT_MAX_DEPTH, 2)
    capi.reset_counters()
    rts = render_ranks(knot_small, p, w, h, 3, [1], poison=False)
    extra = capi.counters()["rays_primary"] - w * h
    assert 0 < extra < 0.05 * w * h
    del rts


def test_stack_layout_is_the_one_the_gloo_tests_gather():
    """nova_hip_tiles_pack's stack == tiles.pack_tiles (the torch restatement the CPU gloo tests use)."""
    import torch
    w, h, ranks = 200, 136, 3
    rng = np.random.default_rng(2)
    for texel_bytes, arr in ((16, rng.standard_normal((h, w, 4)).astype(np.float32)),
                             (4, rng.integers(0, 256, (h, w, 4), dtype=np.uint8))):
        im = capi.Image(w, h, texel_bytes=texel_bytes, data=arr)
        nbytes = capi.tiles_stack_bytes(w, h, ranks, texel_bytes)
        assert nbytes == tiles.max_tiles_per_rank(w, h, ranks) * 64 * 64 * texel_bytes
        stacks = capi.Buffer(nbytes=ranks * nbytes)
        for r in range(ranks):
            capi.tiles_pack(im, r, ranks, stacks.ptr + r * nbytes)
            ref = tiles.pack_tiles(torch.from_numpy(arr), w, h, r, ranks).numpy()
            got = stacks.read(arr.dtype)[r * nbytes // arr.itemsize:(r + 1) * nbytes // arr.itemsize].reshape(ref.shape)
            assert (got.view(np.uint8) == ref.view(np.uint8)).all()
        back = capi.Image(w, h, texel_bytes=texel_bytes)
        capi.tiles_unpack(back, ranks, stacks.ptr)
        assert (back.read().view(np.uint8) == arr.view(np.uint8)).all()
    lib = capi.load()
    assert lib.nova_hip_tiles_pack(im.handle, 3, 3, stacks.ptr) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_tiles_unpack(None, 3, stacks.ptr) == capi.E_INVALID_ARGUMENT


def test_shutdown_then_init_renders_again(cornell):
    """nova_hip_shutdown releases the library's device state and forgets the device."""
    w = h = 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA))
    capi.set_option(capi.OPT_MAX_DEPTH, 3)
    rt = Raytracer(0)
    rt.set_scene(cornell, p, w, h)
    rt.start()
    first = rt.raytrace(3).copy()
    capi.check(capi.load().nova_hip_shutdown())
    capi.init(0)
    assert capi.get_option(capi.OPT_MAX_DEPTH) == 3  # options are per process: they survive a shutdown
    rt.start()
    assert (rt.raytrace(3) == first).all()
    capi.set_option(capi.OPT_MAX_DEPTH, 0)


def test_allgather_through_the_c_abi_on_a_one_rank_rccl_communicator(cornell):
    """nova_hip_tiles_allgather = pack -> ncclAllGather -> unpack inside libnova_hip.so.  One GPU
    can hold a communicator of one rank, where the all-gather is a copy: uchar4 and float4 images, with sizes that
    are not multiples of the tile, must come back bit for bit, and a frame gathered this way must equal the plain
    frame.  (The N-rank layout is the same pack / unpack pair the 2-, 3- and 8-rank tests above check.)"""
    lib = capi.load()
    im = capi.Image(200, 150, texel_bytes=4)
    assert lib.nova_hip_tiles_allgather(im.handle) == capi.E_NO_COMM  # no communicator yet
    uid = capi.tiles_unique_id()
    assert len(uid) == capi.UNIQUE_ID_BYTES and any(uid)
    capi.tiles_comm_init(uid, 0, 1)
    info = capi.tiles_comm_info()
    assert info["ranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 0
    assert "rccl" in info["rccl_path"] and len(info["pci_bus_id"]) >= 7
    rng = np.random.default_rng(12)
    for (w, h, tb) in ((200, 150, 4), (200, 150, 16), (64, 64, 16), (1921, 1081, 4)):
        data = (rng.integers(0, 256, (h, w, 4), dtype=np.uint8) if tb == 4
                else rng.standard_normal((h, w, 4), dtype=np.float32))
        img = capi.Image(w, h, texel_bytes=tb, data=data)
        capi.tiles_allgather(img)
        got = img.read()
        assert (got.view(np.uint8) == data.view(np.uint8)).all(), (w, h, tb)
    # a rendered frame through finish_frame(via="capi"), FXAA off and on, against the plain schedule
    w, h = 320, 200
    for aa in (False, True):
        p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), anti_aliasing=aa)
        capi.set_option(capi.OPT_MAX_DEPTH, 3)
        rt = Raytracer(0)
        rt.set_scene(cornell, p, w, h)
        rt.start()
        for t in (5, 6):
            plain = rt.raytrace(t).copy()
        assert tiles.finish_frame(rt, 0, 1, via="capi") is True
        assert (rt.color_img.read() == plain).all(), aa
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    capi.tiles_comm_destroy()
    assert lib.nova_hip_tiles_allgather(im.handle) == capi.E_NO_COMM
    assert lib.nova_hip_tiles_comm_init(None, 0, 1) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_tiles_comm_init(uid, 2, 2) == capi.E_INVALID_ARGUMENT
