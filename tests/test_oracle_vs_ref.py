"""CPU, build container only: the CPU restatement against oracle/_ref (the reference's own kernel
headers compiled on the CPU backend shim) on seeded inputs beyond the committed fixtures.
Skipped where /root/reference (hence oracle/_ref) does not exist."""
import numpy as np
import pytest

from nova_amd import scene as S
from oracle import bind as O, refbind as R
from tests.conftest import random_rays, rel_rmse
from tests.helpers import id_inputs

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref is built only where /root/reference exists")


@pytest.mark.parametrize("fixture,cam", [("cornell", S.CORNELL_CAMERA), ("knot_small", S.DRAGON_CAMERA)])
@pytest.mark.parametrize("time", [3, 77])
def test_full_path_tracing_is_bit_identical(fixture, cam, time, request):
    sc = request.getfixturevalue(fixture)
    w, h = 96, 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.4, shading_roughness=0.1)
    ref = R.kernel_raytrace(sc, p, time, w, h, denoise_available=1)
    for ordered in (False, True):
        got = O.kernel_raytrace(sc, p, time, w, h, denoise_available=1, ordered=ordered)
        for a, b in zip(got, ref):
            assert rel_rmse(a, b) == 0.0


def test_ids_and_rays(soup50k):
    sc = soup50k
    meta, sky = id_inputs(sc)
    p = S.make_scene_params(S.Camera(width=160, height=120, position=[0.3, 0.2, 3.0], target=[0, 0, 0],
                                     up=[0, 1, 0], fovy=50.0), path_tracing=False)
    a = S.decode_ids(R.kernel_raytrace(sc, p, 9, 160, 120, meta=meta, sky=sky)[0])
    b = S.decode_ids(O.kernel_raytrace(sc, p, 9, 160, 120, meta=meta, sky=sky)[0])
    assert (a == b).all()
    pt, df = random_rays(50000, seed=5, anyhit_every=2, tmax_every=3)
    assert (R.find_intersection(sc, pt, df).view(np.uint32) == O.trace_rays(sc, pt, df).view(np.uint32)).all()


def test_accumulate_and_post_process(cornell):
    w, h = 80, 48  # not a power of two: (x+0.5)/W*W is inexact, the sampler blends by ~1e-7
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), anti_aliasing=True, exposure=1.3)
    pr = po = [np.zeros((h, w, 4), np.float32)] * 3
    for s in range(3):
        c = R.kernel_raytrace(cornell, p, 20 + s, w, h, denoise_available=1)
        pr = R.kernel_accumulate(s, 1, *c, *pr)
        po = O.kernel_accumulate(s, 1, *c, *po)
        for a, b in zip(po, pr):
            assert rel_rmse(a, b) == 0.0
    assert (R.kernel_post_process(p, pr[0]) == O.kernel_post_process(p, po[0])).all()
