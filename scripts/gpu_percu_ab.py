import sys, os, time
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
def run(tune):
    capi.set_option(capi.OPT_TUNE, tune)
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
    return best * 1e3
for rep in range(2):
    for per_cu in (5, 3, 2):
        for pk in (8, 6, 4):
            print("per_cu %d packet_per_cu %d: %.4f ms" % (per_cu, pk, run((per_cu << 12) | (0xD << 16) | (pk << 24))), flush=True)
