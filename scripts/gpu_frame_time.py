"""Wall time per frame of the benchmark frame (dragon stand-in 1080p, max_depth 5), N frames."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
rt.start()
for i in range(3):
    rt.raytrace(1 + i, read_back=False)
capi.synchronize()
best = 1e9
for rep in range(3):
    rt.start()
    t = time.perf_counter()
    for i in range(n):
        rt.raytrace(1 + i, read_back=False)
    capi.synchronize()
    best = min(best, (time.perf_counter() - t) / n)
print("ms per frame: %.4f" % (best * 1e3))
