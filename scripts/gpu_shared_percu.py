"""Four-call and fused ms per sample on the bench frame, C1, C3 and C5 (library defaults; NOVA_HIP_SHARED_BLOCKS_PER_CU from the environment)."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import bench
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
capi.init(0)
def run(name, sc, p, w, h, depth, n=12, u8=None):
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h, materials_u8=u8)
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    best = [1e9, 1e9]
    for rep in range(3):
        rt.start()
        for i in range(3):
            rt.raytrace(1 + i, read_back=False)
        capi.synchronize()
        t = time.perf_counter()
        for i in range(n):
            rt.raytrace(1 + i, read_back=False)
        capi.synchronize()
        best[0] = min(best[0], (time.perf_counter() - t) / n)
        rt.start()
        capi.synchronize()
        t = time.perf_counter()
        rt.raytrace_samples([1 + i for i in range(n)], read_back=False)
        capi.synchronize()
        best[1] = min(best[1], (time.perf_counter() - t) / n)
    print("%s shared_per_cu=%s: four-call %.4f ms, fused %.4f ms" % (name, os.environ.get("NOVA_HIP_SHARED_BLOCKS_PER_CU", "default"), best[0] * 1e3, best[1] * 1e3), flush=True)
W, H = 1920, 1080
run("dragon1080p", S.dragon_standin(), S.make_scene_params(S.Camera(width=W, height=H, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.1), W, H, 5)
run("cornell512", S.cornell_box(), S.make_scene_params(S.Camera(width=512, height=512, **S.CORNELL_CAMERA), path_tracing=True), 512, 512, 1, n=20)
run("room4k", S.room_standin(), S.make_scene_params(S.Camera(width=3840, height=2160, **S.ROOM_CAMERA), **S.ROOM_SHADING), 3840, 2160, 5, n=6)
sc = S.aircraft_scene()
run("aircraft4k", sc, S.make_scene_params(S.Camera(width=3840, height=2160, **S.AIRCRAFT_CAMERA), shading_metallic=0.6, shading_roughness=0.3), 3840, 2160, 5, n=8, u8=S.aircraft_layers_u8())
