"""The largest input nova_hip_build_bvh accepts (2^24 triangles: FlatBVHNode keeps offsets in floats) against
the CPU builder, and one more triangle refused."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402

n = 1 << 24
verts = S.soup(n, seed=9, edge=0.008)
capi.init(0)
dv = capi.Buffer(verts)
t0 = time.time()
bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, n)
print(f"{n} triangles -> {n_nodes} nodes, {stats['levels']} levels: device build {stats['device_ms']:.1f} ms, layout "
      f"{stats['pack_ms']:.1f} ms, call {(time.time() - t0) * 1e3:.0f} ms", flush=True)
t0 = time.time()
host = S.build_scene(verts)
print(f"CPU builder {time.time() - t0:.1f} s", flush=True)
got = bvh.read(S.BVH_NODE, n_nodes)
same = n_nodes == len(host.bvh) and all(np.array_equal(got[f], host.bvh[f]) for f in ("top_offset_left", "bottom_num_right"))
same = same and np.array_equal(order.read(np.uint32, n), host.tri_order)
a = tris.read(np.float32, n * 12)
same = same and np.array_equal(a.view(np.uint32), host.triangles.view(np.uint32).reshape(-1))
print("tree, order, matrices:", "EQUAL" if same else "DIFFERENT", "; layout words differing from the host packer:",
      capi.validate_scene_pack(bvh, tris), flush=True)
try:
    capi.build_bvh(np.zeros((n + 1, 9), np.float32))
    print("2^24 + 1 triangles: ACCEPTED (wrong)")
    same = False
except capi.NovaHipError as e:
    print("2^24 + 1 triangles:", e, "(code", e.err(), ")")
    same = same and e.err() == capi.E_TOO_LARGE
sys.exit(0 if same else 1)
