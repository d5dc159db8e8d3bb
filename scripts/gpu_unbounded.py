"""1080p bench frame with the reference's unbounded paths (max_depth 0): four-call schedule against
nova_hip_render_samples.  usage: gpu_unbounded.py [samples]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nova_amd import scene as S, capi  # noqa: E402
from nova_amd.raytracer import Raytracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
p = S.make_scene_params(S.Camera(width=W, height=H, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
for depth in (5, 0):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    rt.start()
    rt.raytrace(1, read_back=False)
    rt.start()
    capi.reset_counters()
    capi.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        rt.raytrace(1 + i, read_back=False)
    capi.synchronize()
    dt = (time.perf_counter() - t0) / n
    c = capi.counters()
    rays = (c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"] - c.get("rays_mis_pretested", 0) - c.get("rays_shadow_skipped", 0)) / n
    four = rt.temp_color_img2.read()
    print(f"max_depth {depth}: four calls {dt * 1e3:.3f} ms per sample, {rays / 1e6:.2f} Mrays, "
          f"{c['traverse_launches'] / n:.1f} traversal launches -> {rays / dt / 1e6:.0f} Mrays/s", flush=True)
    rt.start()
    rt.raytrace_samples([1 + i for i in range(n)], read_back=False)
    rt.start()
    capi.synchronize()
    t0 = time.perf_counter()
    rt.raytrace_samples([1 + i for i in range(n)], read_back=False)
    capi.synchronize()
    fdt = (time.perf_counter() - t0) / n
    same = bool((rt.temp_color_img2.read().view("u4") == four.view("u4")).all())
    print(f"max_depth {depth}: one call   {fdt * 1e3:.3f} ms per sample -> {rays / fdt / 1e6:.0f} Mrays/s; "
          f"mean of {n} samples bitwise equal: {same}", flush=True)
capi.set_option(capi.OPT_MAX_DEPTH, 0)
