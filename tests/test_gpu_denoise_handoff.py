"""GPU (-m gpu): the device half of the denoise hand-off (reference
src/core/raytracer.cpp:211-242) through the C ABI.  Pinned: tests/golden/ref_pins.npz holds what the reference's
OWN flatten<float4, float, 3> / pack<float4, float, 3> (src/vector/vector_utils.hpp, built from the reference's
file: oracle/ref_vector_utils.cpp) give on a seeded 37x23 plane with NaN, inf and -0 in it; the HIP path must give
the same words.  Larger and ragged sizes against numpy's statement of the same rule."""
import numpy as np
import pytest

from nova_amd import capi
from tests.helpers import golden

pytestmark = pytest.mark.gpu


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_read_and_write_equal_the_reference_templates_on_the_golden_plane():
    capi.init(0)
    g = golden("ref_pins.npz")
    w, h = (int(x) for x in g["image_dims"])
    rgba = g["rgba"].reshape(h, w, 4)
    color = capi.Image(w, h, data=rgba)
    albedo = capi.Image(w, h, data=rgba[::-1].copy())
    normal = capi.Image(w, h, data=(rgba * np.float32(2.0)).astype(np.float32))
    c3, a3, n3, out3 = capi.denoise_read(color, albedo, normal)
    assert (words(c3).reshape(-1) == words(g["flat"])).all()  # flatten<float4, float, 3> of the reference
    assert (words(a3) == words(rgba[::-1, :, :3])).all()
    assert (words(n3) == words((rgba * np.float32(2.0))[..., :3])).all()
    out3[...] = c3  # the "denoiser": identity
    capi.denoise_write(color)
    assert (words(color.read()).reshape(-1, 4) == words(g["packed"])).all()  # pack<float4, float, 3>: w = 0


@pytest.mark.parametrize("w,h", [(1, 1), (3, 1), (5, 7), (64, 64), (1920, 1080), (1281, 721)])
def test_sizes_and_remainders(w, h):
    capi.init(0)
    rng = np.random.default_rng(w * 10007 + h)
    planes = [rng.standard_normal((h, w, 4)).astype(np.float32) for _ in range(3)]
    imgs = [capi.Image(w, h, data=p) for p in planes]
    got = capi.denoise_read(*imgs)
    for k in range(3):
        assert (words(got[k]) == words(planes[k][..., :3])).all()
    res = rng.standard_normal((h, w, 3)).astype(np.float32)
    got[3][...] = res
    capi.denoise_write(imgs[0])
    back = imgs[0].read()
    assert (words(back[..., :3]) == words(res)).all() and (words(back[..., 3]) == 0).all()
    assert (words(imgs[1].read()) == words(planes[1])).all()  # the feature images are only read


def test_error_codes():
    capi.init(0)
    lib = capi.load()
    a, b = capi.Image(8, 8), capi.Image(4, 4)
    planes = capi.DenoisePlanes()
    import ctypes
    assert lib.nova_hip_denoise_read(a.handle, a.handle, b.handle, 8, 8, ctypes.byref(planes)) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_denoise_read(a.handle, a.handle, a.handle, 8, 8, None) == capi.E_INVALID_ARGUMENT
    u8 = capi.Image(8, 8, texel_bytes=4)
    assert lib.nova_hip_denoise_write(u8.handle, 8, 8) == capi.E_INVALID_ARGUMENT
