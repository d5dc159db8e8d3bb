"""Device BVH build (nova_hip_build_bvh) against the host builder (libnova_scene.so), node for node.
usage: python scripts/gpu_bvh_build.py [max_triangles]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402


def compare(name, verts):
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
    t0 = time.time()
    host = S.build_scene(verts)
    t_host = time.time() - t0
    t0 = time.time()
    bvh, tris, order, stats = capi.build_bvh(verts)
    t_call = time.time() - t0
    ok = True
    if len(bvh) != len(host.bvh):
        print(f"  node count {len(bvh)} != {len(host.bvh)}")
        ok = False
    else:
        for f in ("top_offset_left", "bottom_num_right"):
            same = bvh[f] == host.bvh[f]
            if not same.all():
                bad = np.argwhere(~same.all(axis=1))[:, 0]
                print(f"  {f}: {len(bad)} nodes differ, first {bad[:5]}: {bvh[f][bad[0]]} vs {host.bvh[f][bad[0]]}")
                ok = False
    if not np.array_equal(order, host.tri_order):
        bad = np.flatnonzero(order != host.tri_order)
        print(f"  order: {len(bad)} differ, first {bad[:5]}")
        ok = False
    ht = host.triangles.view(np.float32).reshape(-1, 12)
    if ok and not np.array_equal(tris.view(np.uint32), ht.view(np.uint32)):
        bad = np.flatnonzero((tris.view(np.uint32) != ht.view(np.uint32)).any(axis=1))
        print(f"  triangle data: {len(bad)} differ, first {bad[:3]}: {tris[bad[0]]} vs {ht[bad[0]]}")
        ok = False
    print(f"{name}: {len(verts)} triangles, {len(bvh)} nodes, {stats['levels']} levels: "
          f"{'EQUAL' if ok else 'DIFFERENT'}; device {stats['device_ms']:.2f} ms, call {t_call * 1e3:.0f} ms, "
          f"host builder {t_host * 1e3:.0f} ms, bins peak {stats['peak_bin_bytes'] / 2**20:.0f} MiB", flush=True)
    return ok


def main():
    limit = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    print(capi.device_name(), flush=True)
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cornell_box.npz"))
    cases = [("cornell", z["verts"]), ("tiny", S.soup(5, seed=1, edge=0.3)), ("nine", S.soup(9, seed=2, edge=0.3)),
             ("soup-1k", S.soup(1000, seed=3, edge=0.1)), ("soup-50k", S.soup(50000, seed=4, edge=0.05)),
             ("knot-96", S.knot_mesh(96, 96)), ("flat", np.concatenate([S.soup(3000, seed=5, edge=0.05)[:, :9]]) *
              np.tile(np.array([1, 1, 0], np.float32), 3))]
    room = S.room_scene(n_objects=20, res=9)
    cases.append(("room", room[0] if isinstance(room, tuple) else room))
    if limit >= 871_200:
        cases.append(("knot-660", S.knot_mesh(660, 660)))
    if limit >= 1_000_000:
        cases.append(("soup-1M", S.soup(1_000_000, seed=4, edge=0.01)))
    if limit >= 10_000_000:
        cases.append(("soup-10M", S.soup(10_000_000, seed=4, edge=0.01)))
    bad = 0
    for name, v in cases:
        if not compare(name, v):
            bad += 1
    print("FAILED" if bad else "ALL EQUAL")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
