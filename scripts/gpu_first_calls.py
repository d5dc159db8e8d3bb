"""Host time of the first calls of the four-call schedule on a fresh context (where the lanes' allocations land)."""
import sys, time
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
rt.start()
for i in range(8):
    t = time.perf_counter()
    rt.raytrace(1 + i, read_back=False)
    t1 = time.perf_counter()
    capi.synchronize()
    print("call %d: returned after %.3f ms, device done after %.3f ms" % (i, (t1 - t) * 1e3, (time.perf_counter() - t) * 1e3), flush=True)
