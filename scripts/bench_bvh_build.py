"""Measurement of the device BVH build in the format of bench.py's line: one JSON object
per scene with the whole-build rate, the roofline of the build (algorithmic bytes = 68 B per triangle and
level: 40 B binning + 28 B partition) and the CPU builder timed beside it on the host cores.
usage: python scripts/bench_bvh_build.py [--repeats 5] [--no-10m]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nova_amd import capi, scene as S  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_TRIANGLE_LEVEL = 68.0


def main():
    repeats = int(sys.argv[sys.argv.index("--repeats") + 1]) if "--repeats" in sys.argv else 5
    scenes = [("dragon stand-in (knot), 871200 triangles", lambda: S.knot_mesh(660, 660)),
              ("soup, 1000000 triangles", lambda: S.soup(1_000_000, seed=4, edge=0.01))]
    if "--no-10m" not in sys.argv:
        scenes.append(("soup (BASELINE config 4), 10000000 triangles", lambda: S.soup(10_000_000, seed=4, edge=0.01)))
    capi.init(0)
    for name, make in scenes:
        verts = np.ascontiguousarray(make(), np.float32).reshape(-1, 9)
        n = len(verts)
        dv = capi.Buffer(verts)
        device_ms, pack_ms, call_ms = [], [], []
        for r in range(repeats + 1):
            t0 = time.time()
            bvh, tris, order, n_nodes, stats = capi.build_bvh_device(dv, n)
            t1 = time.time()
            if r:  # the first call pays the allocator's first touch
                device_ms.append(stats["device_ms"])
                pack_ms.append(stats["pack_ms"])
                call_ms.append((t1 - t0) * 1e3)
            different = capi.validate_scene_pack(bvh, tris) if r == repeats else None
            for b in (bvh, tris, order):
                b.free()
        t0 = time.time()
        host = S.build_scene(verts)
        cpu_s = time.time() - t0
        ms = float(np.median(device_ms))
        gbs = n * BYTES_PER_TRIANGLE_LEVEL * stats["levels"] / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "BVH build, triangles/s (device time, vertices and results in HBM)",
            "value": round(n / (ms * 1e-3) / 1e6, 1), "unit": "Mtriangles/s", "n_gpus": 1, "higher_is_better": True,
            "dtype": "f32", "data": "synthetic", "config": {"workload": name, "nodes": n_nodes, "levels": stats["levels"]},
            "build_ms": round(ms, 2), "layout_ms": round(float(np.median(pack_ms)), 2),
            "call_ms": round(float(np.median(call_ms)), 2),
            "same_tree_as_cpu_builder": bool(n_nodes == len(host.bvh)), "layout_words_differing_from_host_packer": different,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes": f"{BYTES_PER_TRIANGLE_LEVEL:.0f} B x triangles x levels"},
            "cpu_baseline": {"value": round(n / cpu_s / 1e6, 2), "unit": "Mtriangles/s", "cores": os.cpu_count() if
                             int(os.environ.get("OMP_NUM_THREADS", "0")) == 0 else int(os.environ["OMP_NUM_THREADS"]),
                             "kind": "port", "sample": f"the whole scene once: {cpu_s * 1e3:.0f} ms "
                             "(nova_amd/csrc/host/scene_build.cpp, OpenMP tasks; includes its Woop-matrix loop)"},
        }), flush=True)
        dv.free()


if __name__ == "__main__":
    main()
