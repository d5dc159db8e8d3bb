"""Device BVH build only (for rocprofv3): python scripts/gpu_bvh_profile.py <knot|soup> <size> [repeats]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402

kind, size = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
verts = S.knot_mesh(size, size) if kind == "knot" else S.soup(size, seed=4, edge=0.01)
for r in range(reps):
    bvh, tris, order, stats = capi.build_bvh(verts)
    print(f"{kind} {len(verts)} triangles -> {len(bvh)} nodes, {stats['levels']} levels, device {stats['device_ms']:.2f} ms, "
          f"bins peak {stats['peak_bin_bytes'] / 2**20:.0f} MiB", flush=True)
