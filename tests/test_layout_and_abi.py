"""CPU: struct layouts, and that the C-ABI library loads and exports every symbol include/nova_hip.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np

from nova_amd import capi, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nova_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nova_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    assert declared_functions() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_error_strings_without_device():
    lib = capi.load()
    assert lib.nova_hip_error_string(0) == b"success"
    for code in (-1, -2, -3, -4, -5):
        assert lib.nova_hip_error_string(code).startswith(b"nova_hip:")


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import pytest
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.NovaHipError):
        capi.init(0)
    with pytest.raises(capi.NovaHipError):
        capi.Buffer(np.zeros(4, np.float32))
    with pytest.raises(capi.NovaHipError):  # the BVH build has no CPU path behind the C ABI either
        capi.build_bvh(S.soup(20, seed=1, edge=0.3))


def test_numpy_layouts_match_c():
    S.scene_lib()  # raises on mismatch
    assert S.BVH_NODE.itemsize == 32 and S.TRIANGLE_DATA.itemsize == 48
    assert S.TRIANGLE_META.itemsize == 176 and S.AREA_LIGHT.itemsize == 52
    assert S.SCENE_PARAMS.itemsize == 104
    assert S.TRIANGLE_META.fields["texture_coord1"][1] == 112
    assert S.TRIANGLE_META.fields["light_index"][1] == 172
    assert S.SCENE_PARAMS.fields["path_tracing"][1] == 84


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pat = re.compile(r"(^|\n)\s*(import\s+oracle|from\s+oracle|from\s+\.+oracle|#\s*include\s*[\"<][^\">]*oracle)")
    for dp, _, files in os.walk(os.path.join(ROOT, "nova_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert not pat.search(text), os.path.join(dp, f)


def test_bvh_node_bound_holds_for_the_cpu_builder():
    """nova_hip_bvh_max_nodes (no device needed): every split leaves more than two triangles on each side
    (src/bvh/bvh.cpp:80), so a tree over n triangles has at most 2 * max(1, n // 3) - 1 nodes."""
    lib = capi.load()
    for n, seed, edge in ((1, 1, 0.3), (8, 2, 0.3), (9, 3, 0.3), (100, 4, 0.3), (5000, 5, 0.05), (60000, 6, 0.01)):
        sc = S.build_scene(S.soup(n, seed=seed, edge=edge))
        bound = int(lib.nova_hip_bvh_max_nodes(n))
        assert bound == 2 * max(1, n // 3) - 1
        assert len(sc.bvh) <= bound
        leaves = sc.bvh["bottom_num_right"][:, 3] < 0
        assert len(sc.bvh) == 1 or (-sc.bvh["bottom_num_right"][leaves, 3]).min() >= 3
