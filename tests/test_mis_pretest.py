"""CPU: the light pre-test of MIS rays and the skipped shadow rays of zero light-sample terms (nova_amd/csrc/shade.hpp,
restated in oracle/nova_oracle.cpp `mis_pretest` and trace_ray) do not change a frame by a bit.  It is NOT in the reference: the reference traces every MIS ray
(kernels/raytrace.hpp:234-263).  The claim checked here is the one the library's default relies on — skipping the MIS
rays that pass no light triangle, and shortening the others to the farthest light triangle they pass, leaves radiance,
albedo, normal and the reference's ray counts exactly as they were — on scenes chosen to stress it: two-triangle
lights, many emissive triangles (more than the GPU's cap of 16), several lights seen from one another, paths that land
on a light (the `light != the shaded triangle's` filter), unbounded path length, both walks of the oracle."""
import numpy as np
import pytest

from nova_amd import scene as S
from oracle import bind as O
from tests.helpers import SOUP_CAMERA, lit_soup


def frames(sc, p, w, h, depth, seed, ordered, pretest, counters=None):
    return O.kernel_raytrace(sc, p, seed, w, h, denoise_available=1, max_depth=depth, ordered=ordered,
                             counters=counters, mis_pretest=pretest)


def same_bits(a, b):
    return all((x.view(np.uint32) == y.view(np.uint32)).all() for x, y in zip(a, b))


@pytest.mark.parametrize("name,depth", [("cornell", 5), ("cornell", 0), ("knot", 5), ("lit_soup_6", 4), ("lit_soup_40", 4),
                                        ("lit_soup_6", 0)])
def test_frames_and_ray_counts_do_not_change(name, depth, cornell, knot_small):
    if name == "cornell":
        sc, cam, (w, h) = cornell, S.CORNELL_CAMERA, (96, 96)
    elif name == "knot":
        sc, cam, (w, h) = knot_small, S.DRAGON_CAMERA, (160, 90)
    else:
        sc, cam, (w, h) = lit_soup(n_emissive=int(name.rsplit("_", 1)[1])), SOUP_CAMERA, (96, 64)
    p = S.make_scene_params(S.Camera(width=w, height=h, **cam), shading_metallic=0.3, shading_roughness=0.4)
    for seed in (3, 4):
        c_ref, c_pre = np.zeros(8, np.uint64), np.zeros(8, np.uint64)
        ref = frames(sc, p, w, h, depth, seed, False, False, c_ref)       # the reference's schedule and walk
        pre = frames(sc, p, w, h, depth, seed, False, True, c_pre)        # pre-test, reference walk
        pre_ordered = frames(sc, p, w, h, depth, seed, True, True)        # pre-test, ordered + culled walk
        assert same_bits(ref, pre) and same_bits(ref, pre_ordered)
        assert (c_ref[:4] == c_pre[:4]).all()                             # the same paths, the same calls of the schedule
        assert int(c_ref[6]) == 0 and int(c_ref[7]) == 0
        assert 0 < int(c_pre[6]) <= int(c_ref[2])                         # one MIS ray per light sample (raytrace.hpp:198-263)
        assert 0 < int(c_pre[7]) < int(c_ref[2])                          # some shadow rays belong to zero terms


def test_the_pretest_meets_lights_often_in_the_stress_scene():
    """lit_soup is only a test of the shortened rays if many MIS rays do pass a light triangle."""
    sc = lit_soup(n_emissive=40)
    w, h = 96, 64
    p = S.make_scene_params(S.Camera(width=w, height=h, **SOUP_CAMERA), shading_metallic=0.3, shading_roughness=0.4)
    c = np.zeros(8, np.uint64)
    frames(sc, p, w, h, 4, 3, True, True, c)
    shortened = int(c[2]) - int(c[6])  # MIS rays that pass a light triangle and are traced up to it
    assert shortened > 200 and shortened * 50 > int(c[6])
