"""The CPU builder (nova_amd/csrc/host/scene_build.cpp — the checker of the device BVH build) against an
independent second reading of the reference's src/bvh/bvh.cpp in numpy float32 (tests/independent_bvh.py).
Float32 arithmetic in a fixed order is reproducible, so nodes and triangle order must be equal bit for bit
(the sign of a zero aside).  As with the other second readings, this checks that two readings of the
source agree; it is not an output of the reference."""
import numpy as np
import pytest

from nova_amd import scene as S
from tests import independent_bvh as B
from tests.helpers import golden


def scenes():
    rng = np.random.default_rng(17)
    out = {"cornell": golden("cornell_box.npz")["verts"],
           "soup9": S.soup(9, seed=1, edge=0.3), "soup40": S.soup(40, seed=2, edge=0.3),
           "soup1500": S.soup(1500, seed=3, edge=0.08), "knot": S.knot_mesh(24, 24),
           "flat": S.soup(600, seed=5, edge=0.05) * np.tile(np.array([1, 1, 0], np.float32), 3)}
    centres = rng.uniform(-1, 1, (4, 3))
    pts = np.concatenate([centres[i] + rng.normal(size=(300, 3)) * 0.01 * (1 + 3 * i) for i in range(4)])
    out["clusters"] = (pts[:, None, :] + rng.normal(size=(len(pts), 3, 3)) * 0.004).reshape(-1, 9).astype(np.float32)
    g = np.stack(np.meshgrid(np.arange(20, dtype=np.float32), np.arange(20, dtype=np.float32)), -1).reshape(-1, 2)
    lattice = np.zeros((len(g), 9), np.float32)  # equal costs: the later split must win
    lattice[:, 0:2] = g
    lattice[:, 3:5] = g + np.array([1, 0], np.float32)
    lattice[:, 6:8] = g + np.array([0, 1], np.float32)
    out["lattice"] = lattice / np.float32(20.0)
    return out


SCENES = scenes()


@pytest.mark.parametrize("name", sorted(SCENES))
def test_cpu_builder_equals_the_independent_reading(name):
    verts = np.ascontiguousarray(SCENES[name], np.float32).reshape(-1, 9)
    flat, order = B.build(verts)
    sc = S.build_scene(verts)
    assert len(flat) == len(sc.bvh)
    np.testing.assert_array_equal(flat[:, 0:4], sc.bvh["top_offset_left"])
    np.testing.assert_array_equal(flat[:, 4:8], sc.bvh["bottom_num_right"])
    np.testing.assert_array_equal(order, sc.tri_order)


def test_the_reading_is_sensitive_to_what_it_checks():
    """Guards against vacuity: the quirks that matter really change the tree of a test scene."""
    verts = np.ascontiguousarray(SCENES["soup1500"], np.float32)
    flat, _ = B.build(verts)
    saved = B.MAX_BINS
    try:
        B.MAX_BINS = B.F(512.0)
        other, _ = B.build(verts)
    finally:
        B.MAX_BINS = saved
    assert len(other) != len(flat) or not np.array_equal(other, flat)


@pytest.mark.parametrize("name", ["cornell", "soup40", "knot", "clusters"])
def test_woop_matrices_follow_glm_inverse_operation_order(name):
    """TriangleData = rows 0..2 of transpose(glm::inverse([e1 e2 n-v1 v1])) (intersectable_manager.cpp:129-145):
    the CPU builder's matrices against a numpy float32 reading of glm's scalar mat4 inverse, bit for bit."""
    verts = np.ascontiguousarray(SCENES[name], np.float32).reshape(-1, 9)[:300]
    sc = S.build_scene(verts)
    got = sc.triangles.view(np.float32).reshape(-1, 3, 4)
    for i, t in enumerate(sc.tri_order):
        want = B.woop(verts[t].reshape(3, 3))
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (i, got[i], want)
