"""Diagnostic (a library built with -DNOVA_COUNT_STEPS): wave steps of the camera-ray packet kernels on the benchmark
frame — node fetches and leaf rounds per 64-ray packet.  usage: NOVA_HIP_LIB=build_ab/libnova_hip_steps.so python gpu_packet_steps.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
p = S.make_scene_params(S.Camera(width=W, height=H, **S.DRAGON_CAMERA), path_tracing=False)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_COUNT_WORK, 1)
packets = W * H / 64
for kernel, name in ((12, "box-pair packet"),):  # (13 = the four-wide packet of commit "packet4": profiles/r03_packet_steps.txt)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
    capi.reset_counters()
    rt.kernel_raytrace(1)
    k = capi.counters()
    nodes = k["node_visits"] / (2 if kernel == 13 else 1)
    print("%s: %.1f node steps and %.1f leaf rounds per packet (%d packets)" % (name, nodes / packets, k["triangle_tests"] / packets, packets))
