"""Work counters of the benchmark frame (dragon stand-in 1080p, max_depth 5): node lines and triangle tests
per ray by the production kernels (NOVA_HIP_OPT_COUNT_WORK), and the frame time without counting."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
rt.start()
rt.raytrace(1, read_back=False)
capi.set_option(capi.OPT_COUNT_WORK, 1)
capi.reset_counters()
rt.raytrace(2, read_back=False)
capi.synchronize()
c = capi.counters()
rays = c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"] - c.get("rays_mis_pretested", 0) - c.get("rays_shadow_skipped", 0)
print({k: v for k, v in c.items()})
capi.set_option(capi.OPT_COUNT_WORK, 0)
best = 1e9
for rep in range(3):
    rt.start()
    capi.synchronize()
    t = time.perf_counter()
    for i in range(16):
        rt.raytrace(1 + i, read_back=False)
    capi.synchronize()
    best = min(best, (time.perf_counter() - t) / 16)
print("rays %d  ms per frame: %.4f" % (rays, best * 1e3))
