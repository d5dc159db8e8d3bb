"""Benchmark of the hot path: Mrays/s of the MI355X ray-tracing backend.

    python bench.py --gpus N --steps K --warmup W

One "step" = one sample of the frame through the reference's per-sample schedule
(kernel_raytrace -> kernel_accumulate -> copy prev image -> kernel_post_process,
reference src/core/raytracer.cpp:169-262) with all inputs resident in HBM.

Workload at N = 1 (BASELINE.json configs[1]): dragon scene — the real asset is absent from the
reference tree, so its stand-in is used: a seeded 871,200-triangle displaced torus knot with the
light, ground plane, camera and shading defaults of scene.toml's dragon section —
1920x1080, path tracing, 4 bounces (max_depth 5), 1 spp per step.
At N > 1 the default is BASELINE.json configs[2] ("scaling": "strong"): the fireplace_room stand-in
(closed room, ~142 k triangles, one ceiling light; the real asset is absent), a FIXED 3840x2160 frame
cut into 64x64 tiles dealt round-robin over the N ranks, scene replicated, no communication per sample,
and — inside the timed region — ONE RCCL all-gather of the finished tiles (uchar4 pixels; the float4
means + a whole-frame post-process on rank 0 with --anti-aliasing) through the C ABI
(nova_hip_tiles_allgather: pack -> ncclAllGather -> unpack on the library's stream).  After the timed
region rank 0 renders the same K samples of the same frame alone and reports that as
`same_workload_one_gpu`, the one-GPU point of this strong-scaling curve (the driver's own N = 1 run is
configs[1], a different scene).  --scaling weak keeps round 1's run: the dragon stand-in with a frame
that grows with N.

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts the N ranks itself: a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...` made before this process touches the
GPU; its JSON line passes through and its exit code is returned.  With fewer GPUs than ranks the child
runs the rehearsal mode (ranks share devices, collectives through gloo) and says so in the line.

Prints ONE JSON line on rank 0.  At N = 1 the line also carries, measured after the timed region:
`roofline` (+ the counters' view of what binds), `cpu_baseline`, `fused_samples`, `with_readback`,
`frame_16spp` and `configs` (BASELINE configs C1, C3, C4, C5 at full size: ms, Mrays/s and the bit
"production traversal == reference walk").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # 256 CUs x 128 FP32 FMA lanes x 2 flop x 2.4 GHz
SEED = 1
MAX_DEPTH = 5  # primary segment + 4 bounces


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads() -> int:
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    return max(1, min(threads, len(os.sched_getaffinity(0)), 64))


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: run the ranks as a CHILD torchrun (this process has not touched the
    GPU: torch.cuda.device_count() does not initialise it) and return its exit code."""
    import socket
    import subprocess

    import torch

    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus and "NOVA_BENCH_BACKEND" not in env:
        env["NOVA_BENCH_BACKEND"] = "gloo"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {args.gpus} ranks on {n_dev} visible GPU(s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def cpu_baseline_and_work(oracle, sc, params, width, height, time_seed):
    """rank 0, N = 1 only: (a) the CPU restatement timed on a bounded sample of the same frame
    (reference-order traversal, all host threads we are allowed), (b) the algorithmic work per
    ray from its ordered, distance-culled traversal of the same rays."""
    threads = host_threads()
    oracle.set_threads(threads)
    # calibrate on 1/16 of the rows, then size the timed sample for ~12 s of CPU work
    cal = np.zeros(8, np.uint64)
    t0 = time.perf_counter()
    oracle.kernel_raytrace(sc, params, time_seed, width, height, max_depth=MAX_DEPTH, bands=(4, 16),
                           counters=cal)
    cal_dt = max(time.perf_counter() - t0, 1e-3)
    spp = int(np.clip(round(12.0 / (cal_dt * 16.0)), 1, 64))
    cnt = np.zeros(8, np.uint64)
    t0 = time.perf_counter()
    for s in range(spp):
        oracle.kernel_raytrace(sc, params, time_seed + s, width, height, max_depth=MAX_DEPTH, counters=cnt)
    dt = time.perf_counter() - t0
    rays = int(cnt[0] + cnt[1] + cnt[2])
    # (the rays the library traverses: MIS rays that its light pre-test answers and shadow rays of zero light-sample
    # terms are neither walked nor counted)
    work = np.zeros(8, np.uint64)
    oracle.kernel_raytrace(sc, params, time_seed, width, height, max_depth=MAX_DEPTH,
                           ordered=True, count_work=True, counters=work, mis_pretest=True)
    wr = int(work[0] + work[1] + work[2]) - int(work[6]) - int(work[7])
    # the camera rays alone (they run in their own kernel): the same frame without path tracing
    cam = np.zeros(8, np.uint64)
    p_cam = np.array(params, copy=True)
    p_cam["path_tracing"] = 0
    oracle.kernel_raytrace(sc, p_cam, time_seed, width, height, ordered=True, count_work=True, counters=cam)
    assert int(cam[0]) == int(work[0]) and int(cam[1] + cam[2]) == 0
    n_node = float(work[4] - cam[4]) / (wr - int(cam[0]))
    n_tri = float(work[5] - cam[5]) / (wr - int(cam[0]))
    n_node_cam = float(cam[4]) / int(cam[0])
    n_tri_cam = float(cam[5]) / int(cam[0])
    base = {
        "value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model(), "nproc": os.cpu_count(),
        "sample": f"{spp} spp of the whole {width}x{height} frame, max_depth {MAX_DEPTH}: {rays} rays in "
                  f"{dt:.2f} s, the reference's own schedule (every MIS ray traced) and reference-order traversal "
                  "(kernels/raytrace.hpp:38-102), OpenMP",
    }
    return base, n_node, n_tri, n_node_cam, n_tri_cam


def not_traced(c):
    """find_intersection calls of the reference's schedule that the library answers without a BVH walk."""
    return c["rays_mis_pretested"] + c["rays_shadow_skipped"]


def walked_work(capi, rt, params, seed):
    """What the production kernels really walk for one sample of the frame (device counters, NOVA_HIP_OPT_COUNT_WORK):
    64-byte node lines and triangle tests per ray of k_traverse_p — a four-wide node is two lines — found as the
    whole sample minus the camera rays alone (their own kernel, on the box-pair tree)."""
    capi.set_option(capi.OPT_COUNT_WORK, 1)
    try:
        capi.reset_counters()
        rt.kernel_raytrace(seed)
        full = capi.counters()
        p_cam = np.array(params, copy=True)
        p_cam["path_tracing"] = 0
        keep = rt.params
        rt.params = p_cam
        capi.reset_counters()
        rt.kernel_raytrace(seed)
        cam = capi.counters()
        rt.params = keep
    finally:
        capi.set_option(capi.OPT_COUNT_WORK, 0)
    rays = full["rays_bounce"] + full["rays_shadow"] - not_traced(full)
    if rays == 0:
        return None
    lines = (full["node_visits"] - cam["node_visits"]) / rays
    tris = (full["triangle_tests"] - cam["triangle_tests"]) / rays
    return {"node_lines_per_ray": lines, "wide_nodes_per_ray": lines / 2.0, "tri_tests_per_ray": tris,
            "bytes_per_ray": 64.0 * lines + 48.0 * tris + 64.0}


def counters_view(roof, pj, avg_launch_s, rays_per_launch, rates, name="pmc_traffic.json"):
    """What the rocprofv3 counter passes (profiles/pmc_traffic*.json, scripts/pmc_passes.sh on the same workload)
    say about the kernel, next to the algorithmic figures."""
    if not pj:
        return
    roof["traffic"] = pj.get("hbm_bytes_per_launch")
    roof["traffic_source"] = pj.get("source", f"profiles/{name} (rocprofv3 FETCH_SIZE/WRITE_SIZE passes)")
    if roof["traffic"]:
        # HBM-side bytes per second while the kernel runs: how busy HBM really is
        roof["hbm_frac_counter"] = round(roof["traffic"] / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4)
    wi = pj.get("valu_wave_insts_per_launch")
    tc = pj.get("valu_thread_cycles_per_launch")
    if wi:
        # the rate at which the chip issues wave64 FP32 instructions, measured by nova_hip_measure_device's FMA loop
        # (one v_fma_f32 = 64 lane-ops = 128 flop)
        wave_issue_measured = rates["fp32_fma_tflops"] * 1e12 / 128.0
        roof["valu_issue_frac_counter"] = round(wi / avg_launch_s / wave_issue_measured, 4)
        if tc:
            roof["lanes_active"] = round(tc / wi, 2)  # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU: lanes per VALU instruction
            roof["valu_lane_ops_per_ray_counter"] = round(tc / max(1.0, rays_per_launch), 1)
    if pj.get("valu_pipe_busy") is not None:
        roof["valu_pipe_busy"] = pj["valu_pipe_busy"]  # SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES on the SIMDs that hold waves
    # Which roof: HBM only if the counters say HBM is busy.  Where the working set lives in L2 / Infinity Cache, or the
    # rays were binned so that neighbours share lines, the kernel is co-limited by VALU issue and the latency of its
    # dependent loads, and `frac` (requested bytes against the HBM peak) is a normalised rate, not an HBM utilisation.
    hb = roof.get("hbm_frac_counter")
    if hb is not None and hb >= 0.6:
        roof["bound"] = "hbm"
    else:
        roof["bound"] = "valu issue + dependent-load latency (not hbm: see hbm_frac_counter, valu_pipe_busy, lanes_active)"


def load_pmc(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def production_equals_walk(capi, rt, depth, seed=7):
    """The size-independent parity property: the production (ordered, culled, four-wide) traversal and the
    reference walk (fixed order, no culling: kernels/raytrace.hpp:38-102) give the same radiance plane, bitwise."""
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    out = []
    for trav in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, trav)
        rt.kernel_raytrace(seed)
        out.append(rt.temp_color_img1.read().view(np.uint32))
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    return bool((out[0] == out[1]).all())


def frame_config(capi, rt, depth, n, label):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    same = production_equals_walk(capi, rt, depth)
    rt.start()
    rt.raytrace(1, read_back=False)
    capi.reset_counters()
    capi.synchronize()
    t = time.perf_counter()
    for i in range(n):
        rt.raytrace(2 + i, read_back=False)
    capi.synchronize()
    dt = (time.perf_counter() - t) / n
    c = capi.counters()
    rays = (c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"] - not_traced(c)) / n  # traversed
    m = max(n, 6)
    rt.start()
    rt.raytrace_samples([2 + i for i in range(m)], read_back=False)
    rt.start()
    capi.synchronize()
    t = time.perf_counter()
    rt.raytrace_samples([2 + i for i in range(m)], read_back=False)
    capi.synchronize()
    fdt = (time.perf_counter() - t) / m
    return {"workload": label, "samples_timed": n, "ms_per_sample": round(dt * 1e3, 4), "mrays_per_s": round(rays / dt / 1e6, 1),
            "rays_per_sample": int(rays), "traverse_ms": round(c["traverse_ms"] / n, 4), "shade_ms": round(c["shade_ms"] / n, 4),
            "fused_ms_per_sample": round(fdt * 1e3, 4), "production_equals_reference_walk": same}


def other_configs(capi, S, Raytracer, rates, which):
    """BASELINE.json configs[0], [2], [3], [4] at full size on this one GPU, after the timed region (not bench lines:
    the numbers DESIGN.md §7 quotes, now run by whoever runs bench.py)."""
    out = {}
    if "c1" in which:
        sc = S.cornell_box()
        p = S.make_scene_params(S.Camera(width=512, height=512, **S.CORNELL_CAMERA), path_tracing=True)
        rt = Raytracer(0)
        rt.set_scene(sc, p, 512, 512)
        out["c1"] = frame_config(capi, rt, 1, 20, "cornell_box 512x512, path tracing with one segment (primary + shadow + MIS rays)")
        del rt
    if "c3" in which:
        sc = S.room_standin()
        w, h = 3840, 2160
        p = S.make_scene_params(S.Camera(width=w, height=h, **S.ROOM_CAMERA), **S.ROOM_SHADING)
        rt = Raytracer(0)
        rt.set_scene(sc, p, w, h)
        out["c3"] = frame_config(capi, rt, MAX_DEPTH, 8, f"fireplace_room stand-in ({sc.triangles.shape[0]} triangles), 3840x2160, max_depth {MAX_DEPTH}")
        del rt
    if "c5" in which:
        sc = S.aircraft_scene(sky_size=(4096, 2048))
        w, h = 3840, 2160
        p = S.make_scene_params(S.Camera(width=w, height=h, **S.AIRCRAFT_CAMERA), shading_metallic=0.6, shading_roughness=0.3)
        rt = Raytracer(0)
        rt.set_scene(sc, p, w, h, materials_u8=S.aircraft_layers_u8())
        out["c5"] = frame_config(capi, rt, MAX_DEPTH, 8, f"assets/aircraft ({sc.triangles.shape[0]} triangles, its own UVs, {sc.materials.shape[0]} "
                                 f"{sc.materials.shape[2]}x{sc.materials.shape[1]} uchar4 material layers, normal maps, 4096x2048 analytic sky), "
                                 f"3840x2160, max_depth {MAX_DEPTH}")
        del rt
    if "c4" in which:
        n_tri, n = 10_000_000, 1 << 24
        vb = capi.Buffer(S.soup(n_tri, seed=4))
        bvh, tris, order, n_nodes, stats = capi.build_bvh_device(vb, n_tri)
        vb.free()
        rng = np.random.default_rng(5)
        pt = np.zeros((n, 4), np.float32)
        pt[:, :3] = rng.uniform(-1, 1, (n, 3))
        pt[:, 3] = S.FLT_MAX
        d = rng.standard_normal((n, 3), dtype=np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        df = np.zeros((n, 4), np.float32)
        df[:, :3] = d
        ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
        capi.trace_rays(tris, bvh, ro, rd, hb, n)  # warm-up (the traversal layout was cached by the build)
        batches = 4
        capi.reset_counters()
        capi.synchronize()
        t0 = time.perf_counter()
        for _ in range(batches):
            capi.trace_rays(tris, bvh, ro, rd, hb, n)
        capi.synchronize()
        dt = (time.perf_counter() - t0) / batches
        k = capi.counters()
        launch_s = k["traverse_ms"] / k["traverse_launches"] / 1e3
        got = hb.read(np.uint32).reshape(n, 4)
        m = 1 << 20
        capi.set_option(capi.OPT_TRAVERSAL, 1)
        hw = capi.Buffer(nbytes=16 * m)
        capi.trace_rays(tris, bvh, ro, rd, hw, m)
        capi.set_option(capi.OPT_TRAVERSAL, 0)
        same = bool((hw.read(np.uint32).reshape(m, 4) == got[:m]).all())
        # what the four-wide walk really touches (device counters) — SURVEY §8(d)'s BVH2 model over-counts a collapsed tree
        capi.set_option(capi.OPT_COUNT_WORK, 1)
        capi.reset_counters()
        capi.trace_rays(tris, bvh, ro, rd, hw, m)
        kc = capi.counters()
        capi.set_option(capi.OPT_COUNT_WORK, 0)
        lines, tests = kc["node_visits"] / m, kc["triangle_tests"] / m
        walked_bytes = 64.0 * lines + 48.0 * tests + 64.0
        achieved = walked_bytes * n / launch_s / 1e9
        roof = {"bound": None, "kernel": "k_traverse_p (+ k_ray_keys / scan / k_ray_scatter: the batch is binned first)",
                "model": "walked: 128 B x four-wide nodes + 48 B x triangle tests + 64 B per ray, counted on the device",
                "wide_nodes_per_ray": round(lines / 2, 2), "tri_tests_per_ray": round(tests, 2),
                "bytes_per_ray": round(walked_bytes, 1), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(launch_s * 1e3, 4),
                "frac_of_measured_copy": round(achieved / rates["hbm_copy_gbs"], 4),
                "working_set_bytes": int(n_nodes) * 32 + n_tri * 48, "traffic": None}
        counters_view(roof, load_pmc("pmc_traffic_c4.json"), launch_s, n, rates, "pmc_traffic_c4.json")
        # For this block `frac` is what HBM really moved (the counters) over the peak: the requested bytes per second
        # exceed the HBM peak once the binned rays share lines in L2, and a fraction above 1 says nothing.
        roof["requested_frac"] = roof["frac"]
        if roof.get("hbm_frac_counter") is not None:
            roof["frac"] = roof["hbm_frac_counter"]
            roof["frac_how"] = "HBM bytes per launch by the counters (traffic) / launch time / peak; requested_frac = walked bytes instead"
        out["c4"] = {"workload": f"{n_tri}-triangle soup (seed 4; BVH {n_nodes} nodes built on the device in {stats['device_ms']:.1f} ms), "
                                 f"{n} uniform random closest-hit rays per batch (seed 5)",
                     "ms_per_batch": round(dt * 1e3, 3), "mrays_per_s": round(n / dt / 1e6, 1),
                     "hit_rate": round(float((got[:, 3].view(np.int32) >= 0).mean()), 4),
                     "production_equals_reference_walk": same, "checked_rays": m, "roofline": roof}
    capi.set_option(capi.OPT_MAX_DEPTH, MAX_DEPTH)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--grid", type=int, default=660, help="stand-in mesh resolution (2*grid^2 triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true",
                    help="skip the informational runs after the timed region (fused samples, read-back, 16 spp frame, "
                         "other configs): rocprofv3 --stats then averages the timed schedule's launches only")
    ap.add_argument("--configs", default="c1,c3,c4,c5",
                    help="N = 1: BASELINE configs run at full size after the timed region ('' = none)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 only. strong (default): room stand-in, fixed 3840x2160 (BASELINE configs[2]); "
                         "weak: dragon stand-in, frame grows with N")
    ap.add_argument("--anti-aliasing", action="store_true",
                    help="FXAA on: under tiles the float4 means are gathered and rank 0 post-processes the whole frame")
    ap.add_argument("--check-frame", action="store_true",
                    help="N > 1: rank 0 re-renders the whole frame alone and compares it with the gathered one")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NOVA_BENCH_FORCE_DIST=1 (tests): run the distributed path with a world of ONE rank under the launcher, so that
    # the process group, the library's RCCL communicator and the gather through the C ABI are exercised on a one-GPU box
    multi = world > 1 or (os.environ.get("NOVA_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ)
    dist = None
    torch = None
    # NOVA_BENCH_BACKEND=gloo is the rehearsal mode for boxes with fewer GPUs than ranks: ranks share
    # devices (local_rank % device_count) and the collectives go through host memory.  The driver's
    # runs use the default: one rank per GPU, nccl (= RCCL over xGMI).
    backend = os.environ.get("NOVA_BENCH_BACKEND", "nccl")
    dev_index = local_rank
    n_dev = 1
    if multi:
        import torch
        import torch.distributed as dist
        n_dev = max(1, torch.cuda.device_count())
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dev_index = local_rank % n_dev
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"

    from nova_amd import capi, scene as S, tiles
    from nova_amd.raytracer import Raytracer

    capi.init(dev_index)  # fails loudly without the HIP library / a device
    strong = multi and args.scaling == "strong"
    width, height = (3840, 2160) if strong else tiles.weak_scaling_dims(world)
    if args.width and args.height:
        width, height = args.width, args.height

    # The data-path collective: the library's own RCCL communicator (nova_hip_tiles_allgather) when every rank has
    # its own GPU.  In the rehearsal mode ranks share a device, which RCCL refuses, so the stacks go through
    # torch.distributed (gloo).  If the library cannot open librccl on some rank, every rank falls back together to
    # torch.distributed's all_gather (RCCL as well, through torch) and the line says so.
    via, via_note = "torch", None
    rccl_info = None
    if multi:
        if backend == "nccl":
            try:
                info = capi.tiles_comm_info()
                ok = 1 if info["rccl_version"] > 0 else 0
            except capi.NovaHipError as e:
                ok, via_note = 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=coll_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                tiles.init_comm(rank, world)
                via = "capi"
            else:
                via_note = via_note or "librccl not usable by the library on some rank"
        else:
            via_note = f"rehearsal: {world} ranks share {n_dev} GPU(s), RCCL refuses two ranks on one device"
        mine = dict(capi.tiles_comm_info(), rank=rank, local_rank=local_rank, pid=os.getpid(),
                    torch_device=torch.cuda.get_device_name(dev_index))
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        rccl_info = {
            "backend": backend + (" (RCCL)" if backend == "nccl" else " (rehearsal, host memory)"),
            "world_size": dist.get_world_size(), "ranks_seen": len({s["rank"] for s in seen}),
            "device_per_rank": [s["pci_bus_id"] for s in seen],
            "distinct_devices": len({s["pci_bus_id"] for s in seen}),
            "gather": ("nova_hip_tiles_allgather: pack -> ncclAllGather -> unpack on the library's stream"
                       if via == "capi" else "nova_hip_tiles_pack -> torch.distributed.all_gather_into_tensor -> nova_hip_tiles_unpack"),
            "library_comm_ranks": [s["ranks"] for s in seen],
            "rccl_version": seen[0]["rccl_version"], "rccl_library": seen[0]["rccl_path"],
        }
        if via_note:
            rccl_info["note"] = via_note

    if strong:
        sc = S.room_standin()
        cam = S.Camera(width=width, height=height, **S.ROOM_CAMERA)
        params = S.make_scene_params(cam, path_tracing=True, num_samples=args.steps,
                                     anti_aliasing=args.anti_aliasing, **S.ROOM_SHADING)
        scene_name = "fireplace_room stand-in (closed room with seeded clutter, one ceiling light"
    else:
        sc = S.dragon_standin(args.grid, args.grid)
        cam = S.Camera(width=width, height=height, **S.DRAGON_CAMERA)
        params = S.make_scene_params(cam, shading_diffuse=(1.0, 1.0, 1.0), shading_metallic=0.4,
                                     shading_roughness=0.1, path_tracing=True, num_samples=args.steps,
                                     anti_aliasing=args.anti_aliasing)
        scene_name = "dragon stand-in (displaced torus knot incl. ground+light"
    # under a tile partition FXAA needs the whole frame: the ranks skip kernel_post_process per sample and
    # rank 0 runs it once on the gathered means (tiles.finish_frame)
    per_sample_post = not (multi and args.anti_aliasing)
    rt = Raytracer(dev_index)
    rt.set_scene(sc, params, width, height)
    capi.set_option(capi.OPT_MAX_DEPTH, MAX_DEPTH)
    capi.set_option(capi.OPT_TILE_RANK, rank)
    capi.set_option(capi.OPT_TILE_RANKS, world)
    rt.start()

    def barrier():
        capi.synchronize()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        rt.raytrace(rt.step_time(SEED), read_back=False, post_process=per_sample_post)
    if dist is not None:  # warm the collective too
        tiles.finish_frame(rt, rank, world, coll_device, via=via)
    rt.start()
    capi.reset_counters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rt.raytrace(rt.step_time(SEED), read_back=False, post_process=per_sample_post)
    frame = None
    if dist is not None:
        g0 = time.perf_counter()
        tiles.finish_frame(rt, rank, world, coll_device, via=via)  # rank 0's color_img = the whole frame
        gather_s = time.perf_counter() - g0
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None and rank == 0:
        frame = rt.color_img.read()

    k = capi.counters()
    # rays = rays TRAVERSED.  MIS rays that k_shade's light pre-test answered (NOVA_HIP_OPT_MIS_PRETEST; find_intersection
    # calls of the reference's schedule that the library proves cannot contribute) are reported apart, never in `value`.
    local = np.array([k["rays_primary"], k["rays_bounce"] - k["rays_mis_pretested"], k["rays_shadow"] - k["rays_shadow_skipped"],
                      k["traverse_launches"], k["rays_mis_pretested"], k["rays_shadow_skipped"]], np.float64)
    trav_ms = k["traverse_ms"]
    shade_ms = k["shade_ms"]
    if dist is not None:
        tt = torch.tensor([dt, trav_ms, shade_ms, gather_s], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, trav_ms, shade_ms, gather_s = (float(x) for x in tt.cpu())
        cc = torch.tensor(local, dtype=torch.float64, device=coll_device)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total = cc.cpu().numpy()
        rccl_info["gather_ms"] = round(gather_s * 1e3, 3)  # post-process (or not) + pack + all-gather + unpack, max over ranks
    else:
        total = local
    rays_primary, rays_bounce, rays_shadow = (int(x) for x in total[:3])
    rays_pretested, rays_dark = int(total[4]), int(total[5])
    rays_untraced = rays_pretested + rays_dark
    rays = rays_primary + rays_bounce + rays_shadow
    rays_local = int(local[0] + local[1] + local[2])

    # Informational, outside the timed region (N = 1): the same K samples
    #  * through nova_hip_render_samples (one enqueue, one host wait, one post-process; byte-identical images,
    #    tests/test_gpu_render_samples.py),
    #  * with the uchar4 read-back Raytracer::raytrace ends every sample with (raytracer.cpp:253-255), into pinned memory,
    #  * and BASELINE configs[1]'s whole frame: 16 spp, four calls per sample and fused.
    fused = with_readback = frame16 = None
    if not multi and not args.no_fused:
        times = [(SEED + i) & 0xFFFFFFFF for i in range(args.steps)]
        rt.start()
        rt.raytrace_samples(times, read_back=False)  # warm
        rt.start()
        capi.synchronize()
        f0 = time.perf_counter()
        rt.raytrace_samples(times, read_back=False)
        capi.synchronize()
        fdt = time.perf_counter() - f0
        fused = {"entry": "nova_hip_render_samples", "samples": args.steps,
                 "ms_per_sample": round(fdt / args.steps * 1e3, 4),
                 "mrays_per_s": round(rays_local / fdt / 1e6, 2)}
        pinned = capi.PinnedArray((height, width, 4), np.uint8)
        rt.start()
        rt.raytrace(rt.step_time(SEED), read_back=False)
        rt.color_img.read_into(pinned)
        rt.start()
        capi.synchronize()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            rt.raytrace(rt.step_time(SEED), read_back=False)
            rt.color_img.read_into(pinned)  # read_image2D of the uchar4 frame, every sample
        capi.synchronize()
        rdt = time.perf_counter() - r0
        with_readback = {"ms_per_step_with_readback": round(rdt / args.steps * 1e3, 4),
                         "mrays_per_s": round(rays_local / rdt / 1e6, 2),
                         "what": f"the timed schedule + read_image2D of the {width * height * 4} B uchar4 frame into page-locked host "
                                 "memory every sample (src/core/raytracer.cpp:253-255): the whole Raytracer::raytrace"}
        spp = 16
        t16 = [(SEED + i) & 0xFFFFFFFF for i in range(spp)]
        rt.start()
        capi.synchronize()
        a0 = time.perf_counter()
        for _ in range(spp):
            rt.raytrace(rt.step_time(SEED), read_back=False)
            rt.color_img.read_into(pinned)
        capi.synchronize()
        four_call = time.perf_counter() - a0
        rt.start()
        capi.synchronize()
        a0 = time.perf_counter()
        rt.raytrace_samples(t16, read_back=False)
        rt.color_img.read_into(pinned)
        capi.synchronize()
        fused16 = time.perf_counter() - a0
        frame16 = {"spp": spp, "ms_per_frame_16spp": round(four_call * 1e3, 3), "ms_per_frame_16spp_fused": round(fused16 * 1e3, 3),
                   "what": "BASELINE configs[1]'s frame: 16 samples of 1920x1080 incl. read-back (per sample / once)"}
        pinned.free()

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary + incoherent bounce)",
            "value": round(rays / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{scene_name}; {sc.triangles.shape[0]} triangles, BVH {sc.bvh.shape[0]} nodes), "
                            f"{width}x{height}, path tracing, max_depth {MAX_DEPTH} (4 bounces), 1 spp per step, "
                            f"fixed seed {SEED}+sample",
                "frame": [width, height],
                "tiles": "64x64 round-robin over ranks" if multi else "single GPU",
                "schedule": "kernel_raytrace + kernel_accumulate + copy + kernel_post_process per step (four C-ABI calls that "
                            "return once enqueued, NOVA_HIP_OPT_ASYNC_LAUNCH; kernel_raytrace of successive steps on three "
                            "lanes in turn, NOVA_HIP_OPT_CALL_LANES; the timed region ends with nova_hip_synchronize)"
                            + ("" if not multi else "; one all-gather of the finished tiles per run"
                               + (" (float4 means, FXAA + tone-map of the whole frame on rank 0)" if args.anti_aliasing
                                  else " (uchar4 pixels)")),
            },
            "rays_per_step": rays // args.steps,
            "ray_count": "rays traversed (1 ray = 1 BVH walk).  The reference's schedule has "
                         f"{(rays + rays_untraced) // args.steps} find_intersection calls per step; {rays_pretested // args.steps} of them "
                         f"are MIS rays that k_shade's light pre-test answers without a walk and {rays_dark // args.steps} shadow rays "
                         "of light samples whose term is +0 (frame bit-identical either way, NOVA_HIP_OPT_MIS_PRETEST): "
                         "those are NOT counted in `value`",
            "rays_per_step_reference_schedule": (rays + rays_untraced) // args.steps,
            "mis_rays_pretested_per_step": rays_pretested // args.steps,
            "shadow_rays_skipped_per_step": rays_dark // args.steps,
            "mrays_per_s_reference_schedule": round((rays + rays_untraced) / dt / 1e6, 2),
            "mrays_per_s_primary": round(rays_primary / dt / 1e6, 2),
            "mrays_per_s_incoherent": round((rays_bounce + rays_shadow) / dt / 1e6, 2),
            "mrays_per_s_per_gpu": round(rays / dt / 1e6 / world, 2),
            "ms_per_frame": round(dt / args.steps * 1e3, 4),
            "traverse_ms_per_step": round(trav_ms / args.steps, 4),
            "shade_ms_per_step": round(shade_ms / args.steps, 4),
            "device": capi.device_name(),
        }
        if rccl_info is not None:
            out["rccl"] = rccl_info
        roof = {"bound": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "kernel": "k_traverse_p"}
        if not multi and not args.no_cpu_baseline:
            from oracle import bind as O  # checker / reported baseline only; never the measured path
            base, n_node, n_tri, n_node_cam, n_tri_cam = cpu_baseline_and_work(O, sc, params, width, height, SEED)
            out["cpu_baseline"] = base
            # dominant kernel = k_traverse_p: every traversal launch but the camera-ray one
            bytes_per_ray = 32.0 * n_node + 48.0 * n_tri + 64.0
            launches = max(1, int(local[3]) - int(k["camera_traverse_launches"]))
            rays_k = rays_local - int(local[0])
            avg_launch_s = (k["traverse_ms"] - k["camera_traverse_ms"]) / launches / 1e3
            bytes_per_launch = bytes_per_ray * rays_k / launches
            achieved = bytes_per_launch / avg_launch_s / 1e9
            rates = capi.measure_device()  # measured denominators
            rays_per_s = rays_k / launches / avg_launch_s
            valu_ops_per_ray = 17.0 * n_node + 33.0 * n_tri
            valu_rate_measured = rates["fp32_fma_tflops"] * 1e12 / 2.0  # one fma = one VALU lane-op
            roof.update({
                "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                "model": "SURVEY §8(d): 32 B x box tests + 48 B x triangle tests + 64 B per ray, counts from the CPU oracle's "
                         "ordered, distance-culled BVH2 walk of the same rays",
                "algorithmic_bytes_per_ray": round(bytes_per_ray, 1),
                "n_node_per_ray": round(n_node, 2), "n_tri_per_ray": round(n_tri, 2),
                "rays_per_launch": round(rays_k / launches, 1),
                "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": launches,
                "valu_ops_per_ray": round(valu_ops_per_ray, 1),
                "measured": {"hbm_copy_gbs": round(rates["hbm_copy_gbs"], 1),
                             "fp32_fma_tflops": round(rates["fp32_fma_tflops"], 2),
                             "how": "nova_hip_measure_device: 1 GiB float4 copy (read + write bytes), loop of independent v_fma_f32"},
                "frac_of_measured_copy": round(achieved / rates["hbm_copy_gbs"], 4),
                # the VALU side of SURVEY §8d's max(...): algorithmic slab / Woop lane-ops against the FP32 vector rate
                "valu_frac": round(valu_ops_per_ray * rays_per_s / (FP32_VECTOR_PEAK_TFLOPS * 1e12 / 2.0), 4),
                "valu_frac_of_measured": round(valu_ops_per_ray * rays_per_s / valu_rate_measured, 4),
            })
            # The timed region runs kernel_raytrace of up to three samples side by side (NOVA_HIP_OPT_CALL_LANES), so a
            # launch there shares the GPU with other samples' kernels.  The same K steps once more, one sample at a time:
            # the kernel alone on the GPU, which is also how rocprofv3's counter passes see it (they serialise dispatches).
            capi.set_option(capi.OPT_CALL_LANES, 1)
            rt.start()
            rt.raytrace(rt.step_time(SEED), read_back=False)
            rt.start()
            capi.reset_counters()
            i0 = time.perf_counter()
            for _ in range(args.steps):
                rt.raytrace(rt.step_time(SEED), read_back=False)
            capi.synchronize()
            idt = time.perf_counter() - i0
            ki = capi.counters()
            capi.set_option(capi.OPT_CALL_LANES, 0)
            il = max(1, int(ki["traverse_launches"]) - int(ki["camera_traverse_launches"]))
            iso_launch_s = (ki["traverse_ms"] - ki["camera_traverse_ms"]) / il / 1e3
            iso_achieved = bytes_per_launch / iso_launch_s / 1e9
            roof["note"] = ("achieved / frac / avg_launch_ms are the timed region's, where kernel_raytrace of three samples is in flight "
                            "at once and k_traverse_p runs with 3 instead of 5 work-groups per CU to share them: a launch's "
                            "duration there is not the kernel's speed.  one_sample_at_a_time is the kernel alone on the GPU.")
            roof["one_sample_at_a_time"] = {
                "how": "the same K steps with NOVA_HIP_OPT_CALL_LANES = 1: kernels never share the GPU (the counter fields "
                       "below are priced against this launch time: counter passes serialise dispatches)",
                "ms_per_step": round(idt / args.steps * 1e3, 4), "avg_launch_ms": round(iso_launch_s * 1e3, 4),
                "achieved": round(iso_achieved, 1), "frac": round(iso_achieved / HBM_PEAK_GBS, 4),
                "camera_ray_kernel_ms": round(ki["camera_traverse_ms"] / max(1, int(ki["camera_traverse_launches"])), 4),
                "shade_ms_per_step": round(ki["shade_ms"] / args.steps, 4),
            }
            # the same kernel against the bytes of the tree it really walks (four-wide nodes of 128 B)
            ww = walked_work(capi, rt, params, SEED)
            if ww:
                wa = ww["bytes_per_ray"] * rays_k / launches / avg_launch_s / 1e9
                wi = ww["bytes_per_ray"] * rays_k / launches / iso_launch_s / 1e9
                roof["walked"] = {"model": "128 B x four-wide nodes + 48 B x triangle tests + 64 B per ray, counted on the device "
                                           "(NOVA_HIP_OPT_COUNT_WORK, one sample)",
                                  "wide_nodes_per_ray": round(ww["wide_nodes_per_ray"], 2),
                                  "tri_tests_per_ray": round(ww["tri_tests_per_ray"], 2),
                                  "bytes_per_ray": round(ww["bytes_per_ray"], 1), "achieved": round(wa, 1),
                                  "frac": round(wa / HBM_PEAK_GBS, 4),
                                  "frac_one_sample_at_a_time": round(wi / HBM_PEAK_GBS, 4)}
            # what the counters say (profiles/, rocprofv3 passes of scripts/prof_run.py on the same frame)
            pj = load_pmc("pmc_traffic.json")
            counters_view(roof, pj, iso_launch_s, rays_k / launches, rates)
            if roof["bound"] is None:  # no counter file: say what the measured denominators say
                roof["bound"] = "unknown (profiles/pmc_traffic.json missing)"
            cam_pj = pj.get("camera_ray_kernel") if pj else None
            # camera rays: the packet kernel fetches a node once per 64 rays through the scalar cache, so a
            # per-ray byte model says nothing about it; it is bound by instruction issue (VALU + SALU ports)
            cam_timed_s = k["camera_traverse_ms"] / max(1, int(k["camera_traverse_launches"])) / 1e3
            cam_s = ki["camera_traverse_ms"] / max(1, int(ki["camera_traverse_launches"])) / 1e3  # alone on the GPU
            wave_issue_measured = rates["fp32_fma_tflops"] * 1e12 / 128.0
            cam = {"kernel": "k_traverse_packet", "avg_launch_ms": round(cam_s * 1e3, 4),
                   "avg_launch_ms_timed_region": round(cam_timed_s * 1e3, 4),
                   "rays_per_launch": int(local[0]) // max(1, int(k["camera_traverse_launches"])),
                   "bound": "instruction issue + dependent scalar-load latency (wave-uniform node and triangle fetches through the scalar cache)",
                   "algorithmic_valu_frac": round((17.0 * n_node_cam + 33.0 * n_tri_cam) * int(local[0]) /
                                                  max(1, int(k["camera_traverse_launches"])) / max(cam_s, 1e-12) /
                                                  (FP32_VECTOR_PEAK_TFLOPS * 1e12 / 2.0), 4),
                   "n_node_per_ray": round(n_node_cam, 2), "n_tri_per_ray": round(n_tri_cam, 2)}
            if cam_pj and cam_pj.get("valu_wave_insts_per_launch"):
                cam["issue_frac_valu_counter"] = round(cam_pj["valu_wave_insts_per_launch"] / cam_s / wave_issue_measured, 4)
                cam["salu_wave_insts_per_launch"] = cam_pj.get("salu_wave_insts_per_launch")
            roof["camera_ray_kernel"] = cam
            if args.configs and not args.no_fused:
                out["configs"] = other_configs(capi, S, Raytracer, rates, [c for c in args.configs.split(",") if c])
        else:
            out["cpu_baseline"] = None
            # no oracle, no byte model: still say how long the dominant kernel's launches took by the library's HIP
            # events, so that a rocprofv3 --kernel-trace --stats run of this very process can be held against it
            launches = max(1, int(local[3]) - int(k["camera_traverse_launches"]))
            roof.update({"avg_launch_ms": round((k["traverse_ms"] - k["camera_traverse_ms"]) / launches, 4), "launches": launches,
                         "camera_ray_kernel": {"kernel": "k_traverse_packet", "avg_launch_ms_timed_region": round(
                             k["camera_traverse_ms"] / max(1, int(k["camera_traverse_launches"])), 4)}})
        out["roofline"] = roof
        if fused is not None:
            out["fused_samples"] = fused
        if with_readback is not None:
            out["with_readback"] = with_readback
            out["ms_per_step_with_readback"] = with_readback["ms_per_step_with_readback"]
        if frame16 is not None:
            out["frame_16spp"] = frame16
        if multi:
            # the one-GPU point of the same workload (and, with --check-frame, the frame itself): rank 0
            # renders every tile alone, same seeds, while the other ranks wait at the final barrier
            capi.set_option(capi.OPT_TILE_RANK, 0)
            capi.set_option(capi.OPT_TILE_RANKS, 1)
            rt.start()
            rt.raytrace(rt.step_time(SEED), read_back=False)  # frame buffers grow to the whole frame here
            rt.start()
            capi.reset_counters()
            capi.synchronize()
            a0 = time.perf_counter()
            for _ in range(args.steps):
                rt.raytrace(rt.step_time(SEED), read_back=False)
            capi.synchronize()
            adt = time.perf_counter() - a0
            ka = capi.counters()
            arays = ka["rays_primary"] + ka["rays_bounce"] + ka["rays_shadow"] - not_traced(ka)
            out["same_workload_one_gpu"] = {"ms_per_step": round(adt / args.steps * 1e3, 4),
                                            "value": round(arays / adt / 1e6, 2), "unit": "Mrays/s",
                                            "rays_per_step": int(arays // args.steps)}
            if args.check_frame and frame is not None:
                alone = rt.color_img.read()
                out["frame_check"] = bool((alone == frame).all())
                out["frame_bytes_differing"] = int((alone != frame).sum())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()  # rank 0's one-GPU run of the same workload ends here
        if via == "capi":
            capi.tiles_comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
