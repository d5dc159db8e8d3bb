"""The oracle against an independent second reading of the reference (tests/independent_tracer.py:
pure Python float64, brute-force intersection, written from the reference's source only).  The two
share the inputs and the reference's random-number rule, nothing else.  Every pixel whose discrete
decisions are not within rounding of flipping must agree to 1e-4 of its brightest channel (observed:
~1e-7); the one decision the reference itself leaves to the last bit — a shadow ray ending on the
sampled light's own triangle — is accepted either way.  This does not pin the oracle to an OUTPUT of
the reference (parity unpinned); it checks that two readings of its source agree."""
import numpy as np
import pytest

from nova_amd import scene as S
from oracle import bind as O
from tests import independent_tracer as T


def _compare(sc, params, w, h, time, depth):
    ref = O.kernel_raytrace(sc, params, time, w, h, max_depth=depth)[0][..., :3].astype(np.float64)
    scene = T.Scene(sc)
    ok = ambiguous = 0
    bad = []
    for y in range(h):
        for x in range(w):
            color, _, decisions, amb, optional = T.trace_pixel(scene, params, x, y, w, time, max_depth=depth)
            if amb:
                ambiguous += 1
            elif T.matches(color, optional, ref[y, x]):
                ok += 1
            else:
                bad.append(((x, y), color, ref[y, x], decisions))
    return ok, ambiguous, bad


@pytest.mark.parametrize("depth", [1, 4, 0])
@pytest.mark.parametrize("metallic,roughness", [(0.5, 0.3), (0.0, 0.9)])
def test_cornell_box_triangle_lights(depth, metallic, roughness):
    sc = S.cornell_box()
    w = h = 28
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), shading_metallic=metallic,
                            shading_roughness=roughness)
    ok, ambiguous, bad = _compare(sc, p, w, h, 7 + depth, depth)
    assert not bad, bad[:3]
    assert ok >= 0.6 * w * h, (ok, ambiguous)


def test_rect_light_ground_plane_and_sky():
    """The stand-in scene's set-up (scene.toml's dragon section): a RECT area light, a ground plane,
    shading defaults from the scene parameters, camera rays that leave the scene and read the sky."""
    sc = S.dragon_standin(24, 24)
    w, h = 30, 18
    # roughness 0.35, not the scene's 0.1: with alpha^2 = 1e-4 the GGX term n.m^2 (alpha^2 - 1) + 1 cancels
    # to ~1e-3 relative in float32, which is the oracle's (and the reference's) arithmetic, not a misreading
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.35)
    ok, ambiguous, bad = _compare(sc, p, w, h, 3, 3)
    assert not bad, bad[:3]
    assert ok >= 0.5 * w * h, (ok, ambiguous)


def test_primary_only_mode_and_no_lights():
    sc = S.build_scene(S.soup(300, seed=5, edge=0.3))
    w, h = 24, 16
    cam = S.Camera(width=w, height=h, position=[0, 0, 3.5], target=[0, 0, 0], up=[0, 1, 0], fovy=40.0)
    for pt in (False, True):
        p = S.make_scene_params(cam, shading_metallic=0.2, shading_roughness=0.5, path_tracing=pt)
        ok, ambiguous, bad = _compare(sc, p, w, h, 11, 3)
        assert not bad, bad[:3]
        assert ok >= 0.6 * w * h, (ok, ambiguous)
