"""GPU (-m gpu): nova_hip_build_bvh (BVH::build + the TriangleData loop on the device, SURVEY.md §8f
rank 1) through the C ABI against the CPU builder (nova_amd/csrc/host/scene_build.cpp, the restatement
of src/bvh/bvh.cpp and intersectable_manager.cpp:129-145 that the reference's own AABB / Triangle known
answers pin, tests/test_reference_kats.py).  Bar: the same tree — every FlatBVHNode equal (the sign of
a zero box coordinate aside, which compares equal), the same triangle order, Woop matrices bit for bit."""
import ctypes

import numpy as np
import pytest

from nova_amd import capi, scene as S
from tests.helpers import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def device():
    capi.init(0)
    yield


def assert_same_tree(verts):
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    host = S.build_scene(verts)
    bvh, tris, order, stats = capi.build_bvh(verts)
    assert len(bvh) == len(host.bvh)
    for f in ("top_offset_left", "bottom_num_right"):
        np.testing.assert_array_equal(bvh[f], host.bvh[f])
    np.testing.assert_array_equal(order, host.tri_order)
    want = host.triangles.view(np.float32).reshape(-1, 12)
    both_nan = np.isnan(tris) & np.isnan(want)  # a zero-area triangle has no inverse; NaN payloads are the CPU's
    assert np.array_equal(tris.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])
    assert stats["levels"] == host.max_depth
    return host, stats


def test_cornell_box():
    z = golden("cornell_box.npz")
    host, _ = assert_same_tree(z["verts"])
    assert len(host.bvh) > 1


@pytest.mark.parametrize("n", [1, 2, 8, 9, 10, 17, 64, 65, 1000])
def test_small_inputs_and_the_leaf_threshold(n):
    assert_same_tree(S.soup(n, seed=n, edge=0.3))


@pytest.mark.parametrize("name", ["soup50k", "knot96", "room", "thin_long", "clustered"])
def test_scenes(name):
    rng = np.random.default_rng(9)
    if name == "soup50k":
        v = S.soup(50000, seed=4, edge=0.05)
    elif name == "knot96":
        v = S.knot_mesh(96, 96)
    elif name == "room":
        v = S.room_scene(n_objects=20, res=9)[0]
    elif name == "thin_long":  # long needles: centres far from most of the box, deep unbalanced trees
        c = rng.uniform(-1, 1, (20000, 1, 3)).astype(np.float32)
        d = rng.normal(size=(20000, 1, 3)).astype(np.float32)
        t = np.array([0.0, 0.5, 1.0], np.float32).reshape(1, 3, 1) * np.float32(0.6)
        v = (c + d * t + rng.normal(size=(20000, 3, 3)).astype(np.float32) * np.float32(0.002)).reshape(-1, 9)
    else:  # a few dense clusters and a sparse background: very uneven bins
        centres = rng.uniform(-1, 1, (6, 3))
        pts = np.concatenate([centres[i % 6] + rng.normal(size=(5000, 3)) * 0.01 * (1 + i) for i in range(6)] +
                             [rng.uniform(-1, 1, (3000, 3))]).astype(np.float32)
        v = (pts[:, None, :] + rng.normal(size=(len(pts), 3, 3)).astype(np.float32) * 0.004).reshape(-1, 9)
    assert_same_tree(v)


def test_degenerate_inputs():
    rng = np.random.default_rng(3)
    one = S.soup(1, seed=1, edge=0.3)
    # 40 copies of one triangle: every centre in one bin, no split is valid, one big leaf
    host, _ = assert_same_tree(np.repeat(one, 40, axis=0))
    assert len(host.bvh) == 1
    # all triangles in the plane z = 0: the z axis is not split (bvh.cpp:169)
    flat = S.soup(3000, seed=5, edge=0.05) * np.tile(np.array([1, 1, 0], np.float32), 3)
    assert_same_tree(flat)
    # exact zeros and negative zeros among the coordinates, zero-area triangles
    v = S.soup(2000, seed=6, edge=0.2)
    v[rng.random(v.shape) < 0.05] = 0.0
    v[rng.random(v.shape) < 0.02] = -0.0
    v[::97, 3:6] = v[::97, 0:3]
    assert_same_tree(v)
    # far from the origin and tiny: steps near the float spacing
    assert_same_tree(S.soup(5000, seed=7, edge=0.001) * np.float32(1e-3) + np.float32(1000.0))
    # a regular grid: many triangles exactly on bin boundaries, many equal costs (later split wins)
    g = np.stack(np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32)), -1).reshape(-1, 2)
    quad = np.zeros((len(g), 9), np.float32)
    quad[:, 0:2] = g
    quad[:, 3:5] = g + np.array([1, 0], np.float32)
    quad[:, 6:8] = g + np.array([0, 1], np.float32)
    quad[:, 2] = quad[:, 5] = quad[:, 8] = (g[:, 0] % 2) * 0.5
    assert_same_tree(quad / np.float32(64.0))


def test_trees_that_end_inside_the_blind_levels():
    """From 4096 triangles on, the first levels of a build are enqueued without the host knowing how many nodes they
    hold (bvh_build.hpp: blind levels, as many as log2(n / 2048) promises).  A tree that stops splitting sooner
    must come out the same: the levels enqueued past its end find no nodes."""
    one = S.soup(1, seed=3, edge=0.3)
    # 30 000 copies of one triangle: the root is a leaf, three blind levels run on nothing
    host, _ = assert_same_tree(np.repeat(one, 30000, axis=0))
    assert len(host.bvh) == 1
    # one split, then two unsplittable halves
    two = np.concatenate([np.repeat(one, 20000, axis=0), np.repeat(one + np.float32(1.5), 20000, axis=0)])
    host, _ = assert_same_tree(two)
    assert len(host.bvh) == 3
    # a small splittable part beside a big unsplittable one: the levels thin out long before log2(n / 2048)
    mixed = np.concatenate([np.repeat(one, 60000, axis=0), S.soup(3000, seed=8, edge=0.05)])
    assert_same_tree(mixed)


def test_full_size_frame_scene():
    """BASELINE config 2's stand-in (871 200 triangles): the same tree, and faster than the CPU."""
    host, stats = assert_same_tree(S.knot_mesh(660, 660))
    assert len(host.bvh) > 250_000
    assert stats["device_ms"] < 100.0


def test_errors():
    lib = capi.load()
    v = S.soup(10, seed=1, edge=0.3)
    cap = int(lib.nova_hip_bvh_max_nodes(10))
    bvh = np.zeros(cap, S.BVH_NODE)
    tris = np.zeros((10, 12), np.float32)
    order = np.zeros(10, np.uint32)
    n_nodes = ctypes.c_uint64(0)
    args = (v.ctypes.data, 10, bvh.ctypes.data, cap, ctypes.byref(n_nodes), tris.ctypes.data, order.ctypes.data, None)
    assert lib.nova_hip_build_bvh(*args) == 0 and n_nodes.value >= 1
    assert lib.nova_hip_build_bvh(None, *args[1:]) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_build_bvh(args[0], 0, *args[2:]) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_build_bvh(*args[:3], cap - 1, *args[4:]) == capi.E_INVALID_ARGUMENT
    assert lib.nova_hip_build_bvh(args[0], (1 << 24) + 1, *args[2:]) == capi.E_TOO_LARGE
    assert lib.nova_hip_bvh_max_nodes(1) == 1 and lib.nova_hip_bvh_max_nodes(9) == 5


def test_a_scene_built_on_the_device_renders_the_same_frame():
    from nova_amd.raytracer import Raytracer
    verts = S.soup(20000, seed=8, edge=0.08)
    host = S.build_scene(verts)
    bvh, tris, order, _ = capi.build_bvh(verts)
    dev = S.Scene(bvh, tris.view(S.TRIANGLE_DATA).reshape(-1), host.meta, host.lights, order, host.verts,
                  host.max_depth, host.materials, host.sky)
    w, h = 96, 64
    p = S.make_scene_params(S.Camera(width=w, height=h, position=[0, 0, 3.5], target=[0, 0, 0], up=[0, 1, 0],
                                     fovy=40.0))
    frames = []
    for sc in (host, dev):
        rt = Raytracer(0)
        rt.set_scene(sc, p, w, h)
        rt.kernel_raytrace(3)
        frames.append(rt.temp_color_img1.read())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))


def test_device_resident_build_feeds_trace_rays():
    """nova_hip_build_bvh_device: nothing crosses PCIe; the buffers it fills are traced directly and give
    the hits of the CPU-built scene; rebuilding into the same buffers is seen (scene cache versioning)."""
    from tests.conftest import random_rays
    results = []
    bufs = None
    for seed in (11, 12):
        verts = S.soup(30000, seed=seed, edge=0.06)
        host = S.build_scene(verts)
        if bufs is None:
            dv = capi.Buffer(verts)
            bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, len(verts))
            bufs = (dv, bvh, tris, order)
        else:  # second scene into the SAME device buffers
            dv, bvh, tris, order = bufs
            dv.write(verts)
            lib = capi.load()
            n = ctypes.c_uint64(0)
            capi.check(lib.nova_hip_build_bvh_device(dv.ptr, len(verts), bvh.ptr, bvh.nbytes // 32, ctypes.byref(n),
                                                     tris.ptr, order.ptr, None))
            n_nodes = int(n.value)
        assert n_nodes == len(host.bvh)
        got = bvh.read(S.BVH_NODE, n_nodes)
        for f in ("top_offset_left", "bottom_num_right"):
            np.testing.assert_array_equal(got[f], host.bvh[f])
        np.testing.assert_array_equal(order.read(np.uint32, len(verts)), host.tri_order)
        o, d = random_rays(20000, seed=seed)
        ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=len(o) * 16)
        capi.trace_rays(tris, bvh, ro, rd, hits, len(o))
        dev_hits = hits.read(np.float32).reshape(-1, 4).copy()
        hb, ht = capi.Buffer(host.bvh), capi.Buffer(host.triangles)
        capi.trace_rays(ht, hb, ro, rd, hits, len(o))
        np.testing.assert_array_equal(dev_hits.view(np.uint32), hits.read(np.float32).reshape(-1, 4).view(np.uint32))
        results.append(dev_hits)
    assert not np.array_equal(results[0], results[1])


@pytest.mark.parametrize("name", ["cornell", "soup9", "soup300", "soup30k", "knot96", "room", "needles", "grid"])
def test_traversal_layout_made_on_the_device_equals_the_host_packer(name):
    """nova_hip_build_bvh_device also makes the traversal layout (box-pair and four-wide nodes, cull
    slacks, stack needs) from the builder's tree; nova_hip_validate_scene_pack re-makes it with the host
    packer from the FlatBVHNode[] and counts the words that differ."""
    rng = np.random.default_rng(5)
    if name == "cornell":
        v = golden("cornell_box.npz")["verts"]
    elif name.startswith("soup"):
        n = {"soup9": 9, "soup300": 300, "soup30k": 30000}[name]
        v = S.soup(n, seed=n, edge=0.3 if n < 1000 else 0.05)
    elif name == "knot96":
        v = S.knot_mesh(96, 96)
    elif name == "room":
        v = S.room_scene(n_objects=20, res=9)[0]
    elif name == "needles":
        c = rng.uniform(-1, 1, (20000, 1, 3)).astype(np.float32)
        d = rng.normal(size=(20000, 1, 3)).astype(np.float32)
        t = np.array([0.0, 0.5, 1.0], np.float32).reshape(1, 3, 1) * np.float32(0.6)
        v = (c + d * t + rng.normal(size=(20000, 3, 3)).astype(np.float32) * np.float32(0.002)).reshape(-1, 9)
    else:
        g = np.stack(np.meshgrid(np.arange(48, dtype=np.float32), np.arange(48, dtype=np.float32)), -1).reshape(-1, 2)
        v = np.zeros((len(g), 9), np.float32)
        v[:, 0:2] = g
        v[:, 3:5] = g + np.array([1, 0], np.float32)
        v[:, 6:8] = g + np.array([0, 1], np.float32)
        v /= np.float32(48.0)
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
    dv = capi.Buffer(v)
    bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, len(v))
    assert stats["pack_ms"] > 0.0  # the layout was made by the build call
    assert capi.validate_scene_pack(bvh, tris) == 0


def test_trees_left_to_the_host_packer_still_trace():
    """A single leaf and a leaf of more than 128 triangles (the host packer chains it) are not packed on
    the device: the first launch packs them on the host as before."""
    from tests.conftest import random_rays
    one = S.soup(1, seed=1, edge=0.5)
    for v in (S.soup(5, seed=2, edge=0.4), np.concatenate([np.repeat(one, 200, axis=0), S.soup(300, seed=3, edge=0.2)])):
        v = np.ascontiguousarray(v, np.float32)
        host = S.build_scene(v)
        dv = capi.Buffer(v)
        bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, len(v))
        assert stats["pack_ms"] == 0.0
        o, d = random_rays(5000, seed=4)
        ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=len(o) * 16)
        capi.trace_rays(tris, bvh, ro, rd, hits, len(o))
        got = hits.read(np.float32).reshape(-1, 4).copy()
        capi.trace_rays(capi.Buffer(host.triangles), capi.Buffer(host.bvh), ro, rd, hits, len(o))
        np.testing.assert_array_equal(got.view(np.uint32), hits.read(np.float32).reshape(-1, 4).view(np.uint32))


def test_random_scenes_short_soak():
    """A short run of scripts/gpu_soak_build.py (the long run is logged in profiles/r01_soak_build.log:
    1500 scenes, 37.5 M triangles): soups, surfaces, clusters, needles, flat sets, lattices, duplicates
    with zeros and wide-range coordinates — same tree, same traversal layout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_soak_build.py"), "120", "100", "100000"],
                         cwd=root, capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "0 different" in res.stdout.splitlines()[-1]


@pytest.mark.parametrize("name", ["cornell", "soup2k", "knot96", "knot40_with_light_and_ground", "room20"])
def test_callers_flat_tree_is_packed_on_the_device_like_the_host_packer(name, monkeypatch):
    """A FlatBVHNode[] uploaded by the caller (Nova's own host-built tree): the first launch validates it
    and makes the traversal layout on the device; the host packer must give the same words."""
    from tests.conftest import random_rays
    if name == "cornell":
        sc = S.cornell_box()
    elif name == "soup2k":
        sc = S.build_scene(S.soup(2000, seed=2, edge=0.1))
    elif name == "knot96":
        sc = S.dragon_standin(96, 96)
    elif name == "room20":
        sc = S.room_standin(n_objects=20, res=9)  # closed room, emissive ceiling quad (TRI lights)
    else:
        sc = S.dragon_standin(40, 40)
    bvh, tris = capi.Buffer(sc.bvh), capi.Buffer(sc.triangles)
    o, d = random_rays(3000, seed=1)
    ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=len(o) * 16)
    capi.trace_rays(tris, bvh, ro, rd, hits, len(o))  # packs
    assert capi.validate_scene_pack(bvh, tris) == 0


def test_malformed_flat_trees_still_get_the_host_packers_verdict():
    sc = S.build_scene(S.soup(500, seed=3, edge=0.2))
    o = np.zeros((64, 4), np.float32); o[:, 3] = 3e38
    d = np.zeros((64, 4), np.float32); d[:, 2] = 1
    ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=64 * 16)
    tris = capi.Buffer(sc.triangles)
    cases = []
    b = sc.bvh.copy(); b["bottom_num_right"][0, 3] = 1.0          # right child = left child
    cases.append(b)
    b = sc.bvh.copy(); b["bottom_num_right"][0, 3] = 1e9           # right child out of range
    cases.append(b)
    b = sc.bvh.copy()
    leaf = np.flatnonzero(b["bottom_num_right"][:, 3] < 0)[0]
    b["top_offset_left"][leaf, 3] = 1e7                             # leaf offset beyond the triangles
    cases.append(b)
    for b in cases:
        with pytest.raises(capi.NovaHipError) as e:
            capi.trace_rays(tris, capi.Buffer(b), ro, rd, hits, 64)
        assert e.value.err() == capi.E_BAD_SCENE
    # a right child that is not where the left subtree ends is no tree in pre-order, but Packer::ref_of does
    # not check that: the device front end declines it and the host packer walks it as before
    b = sc.bvh.copy(); b["bottom_num_right"][0, 3] += 1.0
    capi.trace_rays(tris, capi.Buffer(b), ro, rd, hits, 64)


def test_device_built_triangle_data_equals_the_reference_glm_vectors():
    """tests/golden/ref_pins.npz holds the Woop matrices the REFERENCE's own glm calls give (oracle/_ref, built from
    the reference's files; tests/golden/make_ref_pins.py): k_emit_triangles must reproduce them bit for bit."""
    from tests.helpers import golden
    g = golden("ref_pins.npz")
    _, tris, order, _ = capi.build_bvh(g["verts"])
    ref = g["woop"][order]
    got = np.ascontiguousarray(tris, np.float32).reshape(-1, 12)
    fin = np.isfinite(ref).all(1)
    assert fin.mean() > 0.95
    assert (got[fin].view(np.uint32) == ref[fin].view(np.uint32)).all()


@pytest.mark.parametrize("flavour", ["sizes_9_to_34", "shared_centres", "flat_clusters"])
def test_small_node_path_of_the_subtree_kernel(flavour):
    """k_subtree handles nodes of at most 16 triangles four at a time and of 17..32 two at a time, a lane per
    triangle (bvh_build.hpp: tiny_nodes).  Scenes whose subtrees end in clusters of exactly those sizes, with many
    triangles in one bin (equal centres: no split between them), with empty bins between occupied ones (the two
    kinds of right boxes) and with a degenerate axis."""
    rng = np.random.default_rng({"sizes_9_to_34": 31, "shared_centres": 32, "flat_clusters": 33}[flavour])
    tris = []
    for c in range(700):
        k = 9 + c % 26  # 9..34 triangles per cluster
        centre = rng.uniform(-50, 50, 3).astype(np.float32)
        if flavour == "shared_centres":
            # a few distinct centres per cluster, each shared by several triangles of different size
            spots = centre + rng.uniform(-0.5, 0.5, (3, 3)).astype(np.float32)
            pos = spots[rng.integers(0, 3, k)]
        else:
            pos = centre + rng.uniform(-0.5, 0.5, (k, 3)).astype(np.float32)
        for p in pos:
            half = rng.uniform(0.01, 0.2, (3, 3)).astype(np.float32)
            half -= half.mean(axis=0, keepdims=True)  # the vertices' mean is p; the box centre is near it
            if flavour == "flat_clusters":
                half[:, 1] = 0.0
                p = p.copy()
                p[1] = centre[1]  # the whole cluster in one plane: the y axis cannot be split
            tris.append((p + half).reshape(9))
    host, stats = assert_same_tree(np.asarray(tris, np.float32))
    assert stats["levels"] > 8  # the clusters are below depth 7, i.e. in k_subtree's hands
