"""BASELINE.json configs[3] as a bench line of its own (not bench.py's: that is configs[1]): 10 M-triangle soup built
on the device, 16.7 M incoherent closest-hit rays per batch through nova_hip_trace_rays, with the roofline of the one
configuration whose working set (0.7 GB of nodes + triangles) exceeds L2 + Infinity Cache.
usage: bench_c4.py [--batches K] [--prof]   (--prof: two batches, nothing else — the workload of the rocprofv3 passes)"""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--prof", action="store_true")
ap.add_argument("--triangles", type=int, default=10_000_000)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--kernel", type=int, default=0, help="NOVA_HIP_OPT_TRAVERSE_KERNEL (A/B)")
ap.add_argument("--sort", type=int, default=0, help="NOVA_HIP_OPT_SORT_RAYS: 0 auto, 1 never, 2 always")
args = ap.parse_args()

capi.init(0)
capi.set_option(capi.OPT_SORT_RAYS, args.sort)
if args.kernel:
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, args.kernel)
n_tri, n = args.triangles, args.rays
vb = capi.Buffer(S.soup(n_tri, seed=4))
bvh, tris, order, n_nodes, stats = capi.build_bvh_device(vb, n_tri)
vb.free()
rng = np.random.default_rng(5)
pt = np.zeros((n, 4), np.float32)
pt[:, :3] = rng.uniform(-1, 1, (n, 3))
pt[:, 3] = S.FLT_MAX
d = rng.standard_normal((n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
df = np.zeros((n, 4), np.float32)
df[:, :3] = d
ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
capi.trace_rays(tris, bvh, ro, rd, hb, n)  # warm-up (and the traversal layout is cached by the build)
if args.prof:
    capi.trace_rays(tris, bvh, ro, rd, hb, n)
    print("ok")
    sys.exit(0)
capi.reset_counters()
capi.synchronize()
t0 = time.perf_counter()
for _ in range(args.batches):
    capi.trace_rays(tris, bvh, ro, rd, hb, n)
capi.synchronize()
dt = time.perf_counter() - t0
k = capi.counters()
launch_s = k["traverse_ms"] / k["traverse_launches"] / 1e3
hit = (hb.read(np.float32).reshape(n, 4)[:, 3].view(np.int32) >= 0).mean()

# algorithmic work per ray: the CPU oracle's ordered, distance-culled BVH2 traversal of the same tree
from oracle import bind as O  # checker / reported baseline only
host = types.SimpleNamespace(bvh=bvh.read(S.BVH_NODE, n_nodes), triangles=tris.read(S.TRIANGLE_DATA, n_tri))
m = 1 << 20
work = np.zeros(8, np.uint64)
O.trace_rays(host, pt[:m], df[:m], ordered=True, counters=work)
n_node, n_tris_per_ray = float(work[0]) / m, float(work[1]) / m  # trace_rays counters: [0] box tests, [1] triangle tests
threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0)), 64))
O.set_threads(threads)
c0 = time.perf_counter()
O.trace_rays(host, pt[:m], df[:m])  # reference-order walk (kernels/raytrace.hpp:38-102)
cdt = time.perf_counter() - c0
bytes_per_ray = 32.0 * n_node + 48.0 * n_tris_per_ray + 64.0
valu_per_ray = 17.0 * n_node + 33.0 * n_tris_per_ray
rates = capi.measure_device()
achieved = bytes_per_ray * n / launch_s / 1e9
roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
        "traffic": None, "kernel": "k_traverse_p", "algorithmic_bytes_per_ray": round(bytes_per_ray, 1),
        "n_node_per_ray": round(n_node, 2), "n_tri_per_ray": round(n_tris_per_ray, 2), "rays_per_launch": n,
        "avg_launch_ms": round(launch_s * 1e3, 4), "launches": int(k["traverse_launches"]),
        "valu_ops_per_ray": round(valu_per_ray, 1),
        "measured": {"hbm_copy_gbs": round(rates["hbm_copy_gbs"], 1), "fp32_fma_tflops": round(rates["fp32_fma_tflops"], 2)},
        "frac_of_measured_copy": round(achieved / rates["hbm_copy_gbs"], 4),
        "valu_frac": round(valu_per_ray * n / launch_s / (157.3e12 / 2), 4),
        "working_set_bytes": int(n_nodes) * 32 + n_tri * 48}
pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic_c4.json")
if os.path.exists(pmc):
    pj = json.load(open(pmc))
    roof["traffic"] = pj.get("hbm_bytes_per_launch")
    roof["traffic_source"] = "profiles/pmc_traffic_c4.json (rocprofv3 FETCH_SIZE/WRITE_SIZE passes of bench_c4.py --prof)"
    if roof["traffic"]:
        roof["hbm_frac_counter"] = round(roof["traffic"] / launch_s / 1e9 / 8000.0, 4)
print(json.dumps({
    "metric": "Mrays/s (incoherent closest-hit batch)", "value": round(n * args.batches / dt / 1e6, 2), "unit": "Mrays/s",
    "n_gpus": 1, "steps": args.batches, "warmup": 1, "ms_per_step": round(dt / args.batches * 1e3, 4),
    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    "config": {"workload": f"BASELINE configs[3]: {n_tri}-triangle soup (mt seed 4; BVH {n_nodes} nodes, built on the device in "
                           f"{stats['device_ms']:.1f} ms), {n} uniform random closest-hit rays per batch (seed 5), hit rate {hit:.3f}"},
    "cpu_baseline": {"value": round(m / cdt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                     "sample": f"the first {m} rays of the batch, reference-order walk, OpenMP: {cdt:.2f} s"},
    "roofline": roof}), flush=True)
