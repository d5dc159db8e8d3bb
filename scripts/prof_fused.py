"""Workload for rocprofv3 --kernel-trace: the 1080p frame of bench.py, N samples through
nova_hip_render_samples (lanes from NOVA_HIP_LANES), after one warm call.
usage: prof_fused.py [samples]"""
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
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
rt.start()
rt.raytrace_samples([1 + i for i in range(n)], read_back=False)
rt.start()
capi.synchronize()
t0 = time.perf_counter()
rt.raytrace_samples([1 + i for i in range(n)], read_back=False)
capi.synchronize()
dt = time.perf_counter() - t0
print(f"lanes {os.environ.get('NOVA_HIP_LANES', 'default')}: {n} samples, {dt / n * 1e3:.3f} ms per sample")
