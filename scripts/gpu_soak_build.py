"""Soak of the device BVH build: random scenes of random size and shape, each built on the device and by
the CPU builder; nodes, triangle order and Woop matrices must be equal and the device-made traversal
layout must equal the host packer's word for word.
usage: python scripts/gpu_soak_build.py [scenes] [seconds] [max_triangles]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402


def random_scene(rng, max_tris):
    kind = rng.integers(0, 8)
    n = int(np.exp(rng.uniform(np.log(3), np.log(max_tris))))
    if kind == 0:    # uniform soup
        v = S.soup(n, seed=int(rng.integers(1 << 30)), edge=float(np.exp(rng.uniform(np.log(1e-3), np.log(0.5)))))
    elif kind == 1:  # tessellated surface
        m = max(3, int(np.sqrt(n / 2)))
        v = S.knot_mesh(m, m, seed=int(rng.integers(1 << 30)))
    elif kind == 2:  # clusters of very different density
        k = int(rng.integers(2, 9))
        centres = rng.uniform(-1, 1, (k, 3))
        scale = np.exp(rng.uniform(np.log(1e-3), np.log(0.3), k))
        which = rng.integers(0, k, n)
        pts = centres[which] + rng.normal(size=(n, 3)) * scale[which, None]
        v = (pts[:, None, :] + rng.normal(size=(n, 3, 3)) * scale[which, None, None] * 0.2).reshape(-1, 9)
    elif kind == 3:  # needles
        c = rng.uniform(-1, 1, (n, 1, 3))
        d = rng.normal(size=(n, 1, 3)) * rng.uniform(0.01, 1.0)
        t = np.array([0.0, 0.5, 1.0]).reshape(1, 3, 1)
        v = (c + d * t + rng.normal(size=(n, 3, 3)) * 1e-3).reshape(-1, 9)
    elif kind == 4:  # everything in a plane or on a line (axes skipped)
        v = S.soup(n, seed=int(rng.integers(1 << 30)), edge=0.05).reshape(-1, 3, 3).copy()
        for a in rng.choice(3, int(rng.integers(1, 3)), replace=False):
            v[:, :, a] = np.float32(rng.uniform(-1, 1)) if rng.random() < 0.5 else 0.0
        v = v.reshape(-1, 9)
    elif kind == 5:  # lattice: triangles on bin boundaries, many equal costs
        m = max(2, int(np.sqrt(n)))
        g = np.stack(np.meshgrid(np.arange(m, dtype=np.float32), np.arange(m, dtype=np.float32)), -1).reshape(-1, 2)
        v = np.zeros((len(g), 9), np.float32)
        v[:, 0:2] = g
        v[:, 3:5] = g + np.array([1, 0], np.float32)
        v[:, 6:8] = g + np.array([0, 1], np.float32)
        v[:, 2] = v[:, 5] = v[:, 8] = (g[:, 0] % int(rng.integers(1, 4))) * np.float32(0.5)
        v = v / np.float32(rng.choice([m, 64.0, 3.0]))
    elif kind == 6:  # duplicates and zeros
        base = S.soup(max(1, n // int(rng.integers(2, 40))), seed=int(rng.integers(1 << 30)), edge=0.2)
        v = base[rng.integers(0, len(base), n)].copy()
        v[rng.random(v.shape) < 0.03] = 0.0
        v[rng.random(v.shape) < 0.01] = -0.0
    else:            # huge dynamic range of positions and sizes
        c = rng.uniform(-1, 1, (n, 1, 3)) * np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (n, 1, 1)))
        v = (c + rng.normal(size=(n, 3, 3)) * np.exp(rng.uniform(np.log(1e-4), np.log(1.0), (n, 1, 1)))).reshape(-1, 9)
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
    return ["soup", "surface", "clusters", "needles", "flat", "lattice", "duplicates", "range"][kind], v


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    max_tris = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000
    rng = np.random.default_rng(20261110)
    capi.init(0)
    t0 = time.time()
    done = tris_total = packed = bad = uploaded = 0
    o = np.zeros((64, 4), np.float32); o[:, 3] = 3e38
    d = np.zeros((64, 4), np.float32); d[:, 2] = 1
    ray_o, ray_d, ray_hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=64 * 16)
    for i in range(scenes):
        if time.time() - t0 > budget:
            break
        kind, v = random_scene(rng, max_tris)
        host = S.build_scene(v)
        dv = capi.Buffer(v)
        bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, len(v))
        got = bvh.read(S.BVH_NODE, n_nodes)
        ok = n_nodes == len(host.bvh)
        if ok:
            ok = all(np.array_equal(got[f], host.bvh[f]) for f in ("top_offset_left", "bottom_num_right"))
            ok = ok and np.array_equal(order.read(np.uint32, len(v)), host.tri_order)
            a = tris.read(np.float32, len(v) * 12)
            b = host.triangles.view(np.float32).reshape(-1)
            nan = np.isnan(a) & np.isnan(b)
            ok = ok and np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])
        different = -1
        if ok and stats["pack_ms"] > 0:
            different = capi.validate_scene_pack(bvh, tris)
            packed += 1
            ok = different == 0
        if ok and len(host.bvh) >= 3:
            # the same tree as a caller's array: uploaded, packed at the first launch, compared again
            hb, ht = capi.Buffer(host.bvh), capi.Buffer(host.triangles)
            try:
                capi.trace_rays(ht, hb, ray_o, ray_d, ray_hits, 64)
                ok = capi.validate_scene_pack(hb, ht) == 0
                uploaded += 1
            except capi.NovaHipError as e:  # deeper than the reference's own stack (with the chains the host
                # packer makes of leaves above 128 triangles): refused by the host packer, as before
                big_leaf = float((-host.bvh["bottom_num_right"][:, 3]).max()) > 128
                ok = e.err() == capi.E_TOO_LARGE and (host.max_depth > 64 or big_leaf)
            hb.free()
            ht.free()
        if not ok:
            bad += 1
            np.save(f"gpurun_out/soak_build_bad_{i}.npy", v)
            print(f"scene {i} ({kind}, {len(v)} triangles): DIFFERENT (nodes {n_nodes} vs {len(host.bvh)}, "
                  f"layout words {different})", flush=True)
        done += 1
        tris_total += len(v)
        for b_ in (dv, bvh, tris, order):
            b_.free()
        if done % 25 == 0:
            print(f"{done} scenes, {tris_total} triangles, {packed} + {uploaded} layouts compared, {bad} different, "
                  f"{time.time() - t0:.0f} s", flush=True)
    print(f"soak: {done} scenes, {tris_total} triangles, {packed} built + {uploaded} uploaded layouts compared, {bad} different")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
