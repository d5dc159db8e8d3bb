"""GPU check on the dragon stand-in: culled vs reference-walk vs oracle, timing."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from oracle import bind as O
capi.init(0)
t = time.time(); sc = S.dragon_standin(); print("scene build %.1fs" % (time.time() - t), sc.triangles.shape, flush=True)
W, H = 1920, 1080
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
rt = Raytracer(0)
p0 = S.make_scene_params(cam, path_tracing=False)
t = time.time(); rt.set_scene(sc, p0, W, H); print("upload %.2fs" % (time.time() - t), flush=True)
enc = S.encode_ids_in_kd(sc.meta)
neg_sky = S.constant_sky((-1, -1, -1), w=sc.sky.shape[1], h=sc.sky.shape[0])
rt.set_meta(enc); rt.sky_img.write(neg_sky)
ids = {}
for mode in (0, 1):
    capi.set_option(capi.OPT_TRAVERSAL, mode)
    t = time.time(); rt.kernel_raytrace(7); dt = time.time() - t
    ids[mode] = S.decode_ids(rt.temp_color_img1.read())
    print("mode %d first call %.1f ms" % (mode, dt * 1e3), flush=True)
ref = S.decode_ids(O.kernel_raytrace(sc, p0, 7, W, H, meta=enc, sky=neg_sky)[0])
print("primary ids: culled vs walk %d, culled vs oracle %d, hits %d" % ((ids[0] != ids[1]).sum(), (ids[0] != ref).sum(), (ref >= 0).sum()), flush=True)
capi.set_option(capi.OPT_TRAVERSAL, 0)
rt.set_meta(sc.meta); rt.sky_img.write(sc.sky)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt.params = p
capi.set_option(capi.OPT_MAX_DEPTH, 5)
for mode, kern in ((0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (1, 0)):
    capi.set_option(capi.OPT_TRAVERSAL, mode)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kern)
    print("kernel", kern)
    rt.kernel_raytrace(3)
    for cw in (0,):
        capi.set_option(capi.OPT_COUNT_WORK, cw)
        capi.reset_counters()
        t = time.time(); rt.kernel_raytrace(3); dt = time.time() - t
        k = capi.counters()
        rays = k["rays_primary"] + k["rays_bounce"] + k["rays_shadow"]
        print("mode %d count %d: %.2f ms wall, traverse %.2f ms, shade %.2f ms, rays %d -> %.1f Mrays/s (traverse only %.1f); nodes/ray %.1f tris/ray %.1f" % (
            mode, cw, dt * 1e3, k["traverse_ms"], k["shade_ms"], rays, rays / dt / 1e6, rays / k["traverse_ms"] / 1e3,
            k["node_visits"] / rays, k["triangle_tests"] / rays), flush=True)
    c = rt.temp_color_img1.read()
    if mode == 0 and kern == 0: c0 = c
    if mode == 0 and kern >= 1: print('ww vs batch kernel bit-equal:', bool((c0 == c).all()))
capi.set_option(capi.OPT_TRAVERSAL, 0); capi.set_option(capi.OPT_COUNT_WORK, 0); capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
print("culled vs walk radiance bit-equal:", bool((c0 == c).all()), flush=True)
t = time.time(); co = O.kernel_raytrace(sc, p, 3, W, H, max_depth=5)[0]; print("oracle %.1fs" % (time.time() - t))
d = c0[..., :3].astype(np.float64) - co[..., :3]
print("vs oracle: bit-equal px %.5f relrmse %.3e" % ((c0 == co).all(-1).mean(), np.sqrt((d ** 2).sum() / (co[..., :3].astype(np.float64) ** 2).sum())), flush=True)
