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


# ---- the texture path: the reference's own MaterialLoader / SkyLoader --------------------
def test_material_layers_are_the_reference_loaders_texels_and_u8_over_255_is_the_float4_bitwise():
    """tests/golden/aircraft.npz holds what nova::MaterialLoader::build returned for the aircraft's textures
    (oracle/ref_material.cpp: the reference's files compiled as they are): a uint8 window of every layer and a
    float4 window of the array itself.  The 8-bit path of the backend uploads the uint8 texels and divides by 255.0f
    at sampling time: that must be the reference's float4, bit for bit."""
    z = golden("aircraft.npz")
    u8, f4 = z["layers"], z["window_f4"]
    assert u8.shape == (4, 256, 256, 4) and f4.shape == (4, 64, 64, 4) and tuple(z["full_dims"]) == (1792, 1792, 4)
    conv = u8[:, :64, :64].astype(np.float32) / np.float32(255.0)  # material_loader.cpp:69-76
    assert (conv.view(np.uint32) == f4.view(np.uint32)).all()
    assert (S.aircraft_scene().materials[:, :64, :64].view(np.uint32) == f4.view(np.uint32)).all()
    assert (u8[..., 3] == 255).all() and u8[..., :3].std() > 20  # opaque, and not a flat patch
    # the sRGB -> linear step ran on the diffuse layers only (material_loader.cpp:24-33): the glass material uses the
    # same JPEG as diffuse (srgb) and as normal map (not): layer 2 = uint8(pow(layer 3 / 255, 2.2) * 255), up to the
    # resize in between (the conversion runs BEFORE stbir_resize_uint8)
    assert list(z["layer_srgb"]) == [True, False, True, False]
    a, b = u8[2, ..., :3].astype(np.float64), u8[3, ..., :3].astype(np.float64)
    assert np.abs(a - np.power(b / 255.0, 2.2) * 255.0).mean() < 1.5
    sky = z["sky"]
    assert sky.shape == (64, 128, 4) and np.isfinite(sky).all() and sky[..., :3].max() > 1e4  # the sun after pow 2.2


@needs_ref
def test_the_aircraft_fixture_is_what_the_reference_loaders_give_now(tmp_path):
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_aircraft_fixture as M
    if not os.path.exists(M.LIB):
        pytest.skip("oracle/_ref/libnova_ref_material.so not built")
    z = golden("aircraft.npz")
    now = M.build(str(tmp_path))
    assert str(now["full_sha256"]) == str(z["full_sha256"])
    for k in ("layers", "window_f4", "sky", "diffuse_layer", "normal_layer"):
        assert now[k].tobytes() == z[k].tobytes(), k
    # SkyLoader flips vertically and applies pow(x, 2.2): the first row returned is the LAST row of the file
    src = S.analytic_sky(128, 64)[..., :3]
    got = z["sky"][..., :3]
    back = np.power(np.maximum(got[::-1], 0), 1 / 2.2)
    rel = np.abs(back - src) / np.maximum(src, 1e-3)
    assert np.median(rel) < 0.01  # RGBE keeps 8 bits of mantissa
