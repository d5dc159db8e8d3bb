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
and — inside the timed region — ONE RCCL all_gather of the finished tiles to rank 0 (uchar4 pixels; the
float4 means + a whole-frame post-process on rank 0 with --anti-aliasing).  After the timed region rank 0
renders the same K samples of the same frame alone and reports that as `same_workload_one_gpu`, the
one-GPU point of this strong-scaling curve (the driver's own N = 1 run is configs[1], a different scene).
--scaling weak keeps round 1's run: the dragon stand-in with a frame that grows with N.

Prints ONE JSON line on rank 0.
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
WAVE_ISSUE_PEAK = 256 * 4 * 2.4e9  # wave-instructions per second the chip can issue on one port: 1 / SIMD / cycle
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


def cpu_baseline_and_work(scene_mod, oracle, sc, params, width, height, time_seed):
    """rank 0, N = 1 only: (a) the CPU restatement timed on a bounded sample of the same frame
    (reference-order traversal, all host threads we are allowed), (b) the algorithmic work per
    ray from its ordered, distance-culled traversal of the same rays."""
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    threads = max(1, min(threads, len(os.sched_getaffinity(0)), 64))
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
    work = np.zeros(8, np.uint64)
    oracle.kernel_raytrace(sc, params, time_seed, width, height, max_depth=MAX_DEPTH,
                           ordered=True, count_work=True, counters=work)
    wr = int(work[0] + work[1] + work[2])
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
                  f"{dt:.2f} s, reference-order traversal (kernels/raytrace.hpp:38-102), OpenMP",
    }
    return base, n_node, n_tri, n_node_cam, n_tri_cam


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
                    help="skip the informational nova_hip_render_samples run after the timed region "
                         "(rocprofv3 --stats then averages the timed schedule's launches only)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 only. strong (default): room stand-in, fixed 3840x2160 (BASELINE configs[2]); "
                         "weak: dragon stand-in, frame grows with N")
    ap.add_argument("--anti-aliasing", action="store_true",
                    help="FXAA on: under tiles the float4 means are gathered and rank 0 post-processes the whole frame")
    ap.add_argument("--check-frame", action="store_true",
                    help="N > 1: rank 0 re-renders the whole frame alone and compares it with the gathered one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    torch = None
    # NOVA_BENCH_BACKEND=gloo is the rehearsal mode for boxes with fewer GPUs than ranks: ranks share
    # devices (local_rank % device_count) and the collectives go through host memory.  The driver's
    # runs use the default: one rank per GPU, nccl (= RCCL over xGMI).
    backend = os.environ.get("NOVA_BENCH_BACKEND", "nccl")
    dev_index = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dev_index = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend)
    coll_device = "cuda" if backend == "nccl" else "cpu"

    from nova_amd import capi, scene as S, tiles
    from nova_amd.raytracer import Raytracer

    capi.init(dev_index)  # fails loudly without the HIP library / a device
    strong = world > 1 and args.scaling == "strong"
    width, height = (3840, 2160) if strong else tiles.weak_scaling_dims(world)
    if args.width and args.height:
        width, height = args.width, args.height

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
    per_sample_post = not (world > 1 and args.anti_aliasing)
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
        tiles.finish_frame(rt, rank, world, coll_device)
    rt.start()
    capi.reset_counters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rt.raytrace(rt.step_time(SEED), read_back=False, post_process=per_sample_post)
    frame = None
    if dist is not None:
        tiles.finish_frame(rt, rank, world, coll_device)  # rank 0's color_img = the whole frame
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None and rank == 0:
        frame = rt.color_img.read()

    k = capi.counters()
    local = np.array([k["rays_primary"], k["rays_bounce"], k["rays_shadow"], k["traverse_launches"]], np.float64)
    trav_ms = k["traverse_ms"]
    shade_ms = k["shade_ms"]
    if dist is not None:
        tt = torch.tensor([dt, trav_ms, shade_ms], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, trav_ms, shade_ms = (float(x) for x in tt.cpu())
        cc = torch.tensor(local, dtype=torch.float64, device=coll_device)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        total = cc.cpu().numpy()
    else:
        total = local
    rays_primary, rays_bounce, rays_shadow = (int(x) for x in total[:3])
    rays = rays_primary + rays_bounce + rays_shadow
    rays_local = int(local[0] + local[1] + local[2])

    # Informational, outside the timed region: the same K samples through nova_hip_render_samples
    # (one enqueue, one host wait, one post-process; byte-identical images, tests/test_gpu_render_samples.py).
    fused = None
    if world == 1 and not args.no_fused:
        rt.start()
        rt.raytrace_samples([(SEED + i) & 0xFFFFFFFF for i in range(args.steps)], read_back=False)  # warm
        rt.start()
        capi.synchronize()
        f0 = time.perf_counter()
        rt.raytrace_samples([(SEED + i) & 0xFFFFFFFF for i in range(args.steps)], read_back=False)
        capi.synchronize()
        fdt = time.perf_counter() - f0
        fused = {"entry": "nova_hip_render_samples", "samples": args.steps,
                 "ms_per_sample": round(fdt / args.steps * 1e3, 4),
                 "mrays_per_s": round(rays_local / fdt / 1e6, 2)}

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
                "tiles": "64x64 round-robin over ranks" if world > 1 else "single GPU",
                "schedule": "kernel_raytrace + kernel_accumulate + copy + kernel_post_process per step"
                            + ("" if world == 1 else "; one all_gather of the finished tiles per run"
                               + (" (float4 means, FXAA + tone-map of the whole frame on rank 0)" if args.anti_aliasing
                                  else " (uchar4 pixels)")),
            },
            "rays_per_step": rays // args.steps,
            "mrays_per_s_primary": round(rays_primary / dt / 1e6, 2),
            "mrays_per_s_incoherent": round((rays_bounce + rays_shadow) / dt / 1e6, 2),
            "mrays_per_s_per_gpu": round(rays / dt / 1e6 / world, 2),
            "ms_per_frame": round(dt / args.steps * 1e3, 4),
            "traverse_ms_per_step": round(trav_ms / args.steps, 4),
            "shade_ms_per_step": round(shade_ms / args.steps, 4),
            "device": capi.device_name(),
        }
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "kernel": "k_traverse_p"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import bind as O  # checker / reported baseline only; never the measured path
            base, n_node, n_tri, n_node_cam, n_tri_cam = cpu_baseline_and_work(S, O, sc, params, width, height, SEED)
            out["cpu_baseline"] = base
            # dominant kernel = k_traverse_p: every traversal launch but the camera-ray one
            bytes_per_ray = 32.0 * n_node + 48.0 * n_tri + 64.0
            launches = max(1, int(local[3]) - int(k["camera_traverse_launches"]))
            rays_k = rays_local - int(local[0])
            avg_launch_s = (k["traverse_ms"] - k["camera_traverse_ms"]) / launches / 1e3
            bytes_per_launch = bytes_per_ray * rays_k / launches
            achieved = bytes_per_launch / avg_launch_s / 1e9
            cam_bytes = 32.0 * n_node_cam + 48.0 * n_tri_cam + 64.0
            cam_s = k["camera_traverse_ms"] / max(1, int(k["camera_traverse_launches"])) / 1e3
            rates = capi.measure_device()  # measured denominators
            rays_per_s = rays_k / launches / avg_launch_s
            valu_ops_per_ray = 17.0 * n_node + 33.0 * n_tri
            valu_rate_measured = rates["fp32_fma_tflops"] * 1e12 / 2.0  # one fma = one VALU lane-op
            roof.update({
                "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
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
            # what the counters say (profiles/, rocprofv3 passes of scripts/prof_run.py on the same frame)
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    roof["traffic"] = pj.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE/WRITE_SIZE passes)"
                    if roof["traffic"]:
                        # HBM-side bytes per second while the kernel runs: how busy HBM really is.  Far below `frac`
                        # because the scene (51 MB) is served by L1 / L2 / Infinity Cache, not by HBM
                        roof["hbm_frac_counter"] = round(roof["traffic"] / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4)
                    if pj.get("valu_wave_insts_per_launch"):
                        wi = pj["valu_wave_insts_per_launch"]
                        roof["valu_issue_frac_counter"] = round(wi / avg_launch_s / WAVE_ISSUE_PEAK, 4)
                        roof["valu_lane_ops_per_ray_counter"] = round(pj.get("valu_thread_cycles_per_launch", 0) / max(1.0, rays_k / launches), 1)
                    cam_pj = pj.get("camera_ray_kernel")
                except Exception:
                    cam_pj = None
            else:
                cam_pj = None
            # camera rays: the packet kernel fetches a node once per 64 rays through the scalar cache, so a
            # per-ray byte model says nothing about it; it is bound by instruction issue (VALU + SALU ports)
            cam_s = k["camera_traverse_ms"] / max(1, int(k["camera_traverse_launches"])) / 1e3
            cam = {"kernel": "k_traverse_packet", "avg_launch_ms": round(cam_s * 1e3, 4),
                   "rays_per_launch": int(local[0]) // max(1, int(k["camera_traverse_launches"])),
                   "bound": "instruction issue (wave-uniform node and triangle fetches through the scalar cache)",
                   "algorithmic_valu_frac": round((17.0 * n_node_cam + 33.0 * n_tri_cam) * int(local[0]) /
                                                  max(1, int(k["camera_traverse_launches"])) / max(cam_s, 1e-12) /
                                                  (FP32_VECTOR_PEAK_TFLOPS * 1e12 / 2.0), 4),
                   "n_node_per_ray": round(n_node_cam, 2), "n_tri_per_ray": round(n_tri_cam, 2)}
            if cam_pj and cam_pj.get("valu_wave_insts_per_launch"):
                cam["issue_frac_valu_counter"] = round(cam_pj["valu_wave_insts_per_launch"] / cam_s / WAVE_ISSUE_PEAK, 4)
                cam["issue_frac_salu_counter"] = round(cam_pj.get("salu_wave_insts_per_launch", 0) / cam_s / WAVE_ISSUE_PEAK, 4)
            roof["camera_ray_kernel"] = cam
        else:
            out["cpu_baseline"] = None
        out["roofline"] = roof
        if fused is not None:
            out["fused_samples"] = fused
        if world > 1:
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
            arays = ka["rays_primary"] + ka["rays_bounce"] + ka["rays_shadow"]
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
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
