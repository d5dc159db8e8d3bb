"""Scene set-up cost on the GPU box: device BVH build vs the first use of the scene (repack of the tree
into the traversal layout, get_scene in nova_hip.hip).  usage: gpu_upload_time.py <knot|soup> <size>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402

kind, size = sys.argv[1], int(sys.argv[2])
verts = S.knot_mesh(size, size) if kind == "knot" else S.soup(size, seed=4, edge=0.01)
n = len(verts)
capi.init(0)
dv = capi.Buffer(np.ascontiguousarray(verts, np.float32))
for rep in range(2):
    t0 = time.time()
    bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, n)
    t_build = time.time() - t0
    rng = np.random.default_rng(1)
    o = np.zeros((4096, 4), np.float32); o[:, :3] = rng.uniform(-1, 1, (4096, 3)); o[:, 3] = 3.0e38
    d = np.zeros((4096, 4), np.float32); d[:, :3] = rng.normal(size=(4096, 3)); d[:, :3] /= np.linalg.norm(d[:, :3], axis=1, keepdims=True)
    ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=4096 * 16)
    t0 = time.time()
    capi.trace_rays(tris, bvh, ro, rd, hits, 4096)
    t_first = time.time() - t0
    t0 = time.time()
    capi.trace_rays(tris, bvh, ro, rd, hits, 4096)
    t_second = time.time() - t0
    different = capi.validate_scene_pack(bvh, tris) if stats["pack_ms"] > 0 else -1
    print(f"{kind} {n} triangles, {n_nodes} nodes: build call {t_build * 1e3:.1f} ms (device build {stats['device_ms']:.1f} ms, "
          f"device pack {stats['pack_ms']:.1f} ms, words differing from the host packer: {different}), "
          f"first trace (repack + upload) {t_first * 1e3:.1f} ms, second trace {t_second * 1e3:.2f} ms", flush=True)
    for b in (bvh, tris, order, ro, rd, hits):
        b.free()

# the same scene as a caller's arrays (built on the CPU, uploaded): validation + layout at the first launch
host = S.build_scene(verts)
hb, ht = capi.Buffer(host.bvh), capi.Buffer(host.triangles)
ro, rd, hits = capi.Buffer(o), capi.Buffer(d), capi.Buffer(nbytes=4096 * 16)
t0 = time.time()
capi.trace_rays(ht, hb, ro, rd, hits, 4096)
t_first = time.time() - t0
print(f"{kind} {n} triangles uploaded as FlatBVHNode[]: first trace (validate + layout) {t_first * 1e3:.1f} ms, "
      f"words differing from the host packer: {capi.validate_scene_pack(hb, ht)}", flush=True)
