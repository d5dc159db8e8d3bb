"""A/B of the 64-byte quantised nodes (default) against the 128-byte nodes (NOVA_HIP_OPT_TRAVERSE_KERNEL 23) on the
benchmark frame: bitwise-equal images, ms per sample, node visits and triangle tests."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer

capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
p = S.make_scene_params(S.Camera(width=W, height=H, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)


def run(kernel):
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
    rt.start()
    for i in range(3):
        rt.raytrace(1 + i, read_back=False)
    capi.synchronize()
    best = 1e9
    for rep in range(3):
        rt.start()
        t = time.perf_counter()
        for i in range(16):
            rt.raytrace(1 + i, read_back=False)
        capi.synchronize()
        best = min(best, (time.perf_counter() - t) / 16)
    img = rt.color_img.read()
    capi.set_option(capi.OPT_COUNT_WORK, 1)
    capi.reset_counters()
    rt.start()
    rt.raytrace(1, read_back=False)
    capi.synchronize()
    k = capi.counters()
    capi.set_option(capi.OPT_COUNT_WORK, 0)
    return best * 1e3, img, k["node_visits"], k["triangle_tests"]


a = run(23)
b = run(0)
print("128-byte nodes: %.4f ms per sample, node lines %d, triangle tests %d" % (a[0], a[2], a[3]))
print(" 64-byte nodes: %.4f ms per sample, node lines %d, triangle tests %d" % (b[0], b[2], b[3]))
print("images equal:", bool(np.array_equal(a[1], b[1])))
