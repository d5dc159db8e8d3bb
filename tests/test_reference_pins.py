"""CPU: what the reference lets this repository pin with its own code.
tests/golden/ref_pins.npz holds OUTPUTS OF THE REFERENCE ITSELF — oracle/_ref/*.so, built by oracle/Makefile from
the reference's files where they lie (glm::inverse / transpose / cross / normalize as IntersectableManager::build
calls them, nova::Triangle's bounds and hex stream operators, flatten / pack of src/vector/vector_utils.hpp) —
written by tests/golden/make_ref_pins.py in the build container.  The tests below run everywhere; where
oracle/_ref exists (the build container) the fixture itself is re-derived and more scenes are checked live."""
import ctypes
import os

import numpy as np
import pytest

from nova_amd import scene as S
from tests.helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
have_ref = all(os.path.exists(os.path.join(REF, "libnova_ref_%s.so" % n)) for n in ("glm", "triangle", "vector_utils"))
needs_ref = pytest.mark.skipif(not have_ref, reason="oracle/_ref is built only where /root/reference exists")


def builder_woop(verts):
    """TriangleData of every input triangle, in INPUT order, from the CPU scene builder."""
    sc = S.build_scene(verts)
    out = np.zeros((verts.shape[0], 12), np.float32)
    out[sc.tri_order] = sc.triangles.view(np.float32).reshape(-1, 12)
    return out


def test_woop_matrices_equal_the_reference_glm_arithmetic_bitwise():
    g = golden("ref_pins.npz")
    got = builder_woop(g["verts"])
    ref = g["woop"]
    same = (got.view(np.uint32) == ref.view(np.uint32)).all(1)
    # both sides may hold NaN/inf rows for degenerate inputs; they must then agree in kind
    assert (np.isfinite(got).all(1) == np.isfinite(ref).all(1)).all()
    fin = np.isfinite(ref).all(1)
    assert fin.mean() > 0.95
    assert same[fin].all(), np.flatnonzero(~same & fin)[:10]


def test_triangle_bounds_equal_the_reference_bitwise():
    g = golden("ref_pins.npz")
    v = g["verts"]
    out = np.zeros(6, np.float32)
    for i in range(0, v.shape[0], 7):
        tri = np.ascontiguousarray(v[i])
        S.scene_lib().nova_scene_test_tri_bounds(ctypes.c_void_p(tri.ctypes.data), ctypes.c_void_p(out.ctypes.data))
        assert out.tobytes() == g["bounds"][i].tobytes(), i


def test_hex_dump_format_is_the_reference_stream_operator():
    g = golden("ref_pins.npz")
    for i, text in enumerate(g["hex_text"]):
        assert S.to_hex_words(g["verts"][i]) == str(text)  # operator<< (triangle.cpp:42-55)
        assert S.from_hex_words(str(text), np.float32).tobytes() == g["verts"][i].tobytes()  # operator>> (:27-40)


def test_flatten_and_pack_known_answers():
    """The host half of the denoise hand-off (raytracer.cpp:219-221, :240) on the reference's own templates:
    flatten keeps x, y, z of every float4; pack writes them back and leaves w = 0."""
    g = golden("ref_pins.npz")
    rgba, flat, packed = g["rgba"], g["flat"], g["packed"]
    assert flat.view(np.uint32).tobytes() == np.ascontiguousarray(rgba[:, :3]).view(np.uint32).tobytes()
    assert packed[:, :3].view(np.uint32).tobytes() == np.ascontiguousarray(rgba[:, :3]).view(np.uint32).tobytes()
    assert (packed[:, 3].view(np.uint32) == 0).all()


@needs_ref
def test_the_committed_fixture_is_what_the_reference_build_gives_now():
    import subprocess
    import sys
    import tempfile
    before = {k: golden("ref_pins.npz")[k] for k in ("verts", "woop", "bounds", "flat", "packed")}
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_ref_pins as M
    v = M.pin_triangles()
    assert v.tobytes() == before["verts"].tobytes()
    assert M.ref_woop(v).tobytes() == before["woop"].tobytes()


@needs_ref
@pytest.mark.parametrize("name", ["cornell", "knot96", "soup50k", "room20", "aircraft"])
def test_every_test_scenes_triangle_data_live_against_the_reference(name, request):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_ref_pins as M
    sc = {"cornell": S.cornell_box, "knot96": lambda: S.dragon_standin(96, 96),
          "soup50k": lambda: S.build_scene(S.soup(50000, seed=4, edge=0.05)),
          "room20": lambda: S.room_standin(n_objects=20, res=9), "aircraft": S.aircraft_scene}[name]()
    mine = np.isfinite(sc.verts).all(1)  # triangles the builder adds itself (ground plane) carry no input vertices
    ref = M.ref_woop(sc.verts[mine])
    got = sc.triangles.view(np.float32).reshape(-1, 12)[mine]
    fin = np.isfinite(ref).all(1)
    assert fin.mean() > 0.99
    assert (got[fin].view(np.uint32) == ref[fin].view(np.uint32)).all()
