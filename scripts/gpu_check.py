"""Ad-hoc GPU check: HIP path vs the CPU oracle on Cornell box and a small soup."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from oracle import bind as O

def relrmse(a, b):
    a = a[..., :3].astype(np.float64); b = b[..., :3].astype(np.float64)
    return np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30))

print(capi.device_count(), flush=True)
capi.init(0)
print(capi.device_name(), flush=True)
W = H = 256
for name, sc, camkw in [("cornell", S.cornell_box(), S.CORNELL_CAMERA),
                        ("soup", S.build_scene(S.soup(50000, seed=4, edge=0.05)), dict(position=[0,0,3.5], target=[0,0,0], up=[0,1,0], fovy=40.0))]:
    cam = S.Camera(width=W, height=H, **camkw)
    print(name, "tris", sc.triangles.shape[0], "nodes", sc.bvh.shape[0], "depth", sc.max_depth, flush=True)
    rt = Raytracer(0, denoise_available=True)
    # primary ids
    p0 = S.make_scene_params(cam, path_tracing=False)
    rt.set_scene(sc, p0, W, H)
    rt.set_meta(S.encode_ids_in_kd(sc.meta))
    rt.sky_img.write(S.constant_sky((-1, -1, -1)))
    for mode in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, mode)
        rt.kernel_raytrace(7)
        ids = S.decode_ids(rt.temp_color_img1.read())
        ref = S.decode_ids(O.kernel_raytrace(sc, p0, 7, W, H, meta=S.encode_ids_in_kd(sc.meta), sky=S.constant_sky((-1, -1, -1)))[0])
        print("  primary-id mismatches (mode %d):" % mode, int((ids != ref).sum()), "hits", int((ref >= 0).sum()), flush=True)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    # path tracing, several depth caps
    for depth in (1, 5, 0):
        p = S.make_scene_params(cam, path_tracing=True)
        rt.set_scene(sc, p, W, H)
        capi.set_option(capi.OPT_MAX_DEPTH, depth)
        capi.reset_counters()
        t = time.time(); rt.kernel_raytrace(3); dt = time.time() - t
        c = rt.temp_color_img1.read(); a = rt.albedo_img1.read(); n = rt.normal_img1.read()
        cnt = np.zeros(8, np.uint64)
        co, ao, no = O.kernel_raytrace(sc, p, 3, W, H, denoise_available=1, max_depth=depth, counters=cnt)
        k = capi.counters()
        print("  depth %d: relrmse color %.3e albedo %.3e normal %.3e | bit-equal px %.4f | %.1f ms | rays gpu %d/%d/%d cpu %d/%d/%d seg %d/%d" % (
            depth, relrmse(c, co), relrmse(a, ao), relrmse(n, no), (c == co).all(-1).mean(), dt * 1e3,
            k["rays_primary"], k["rays_bounce"], k["rays_shadow"], cnt[0], cnt[1], cnt[2], k["path_segments"], cnt[3]), flush=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    # ray batch
    rng = np.random.default_rng(0)
    n = 200000
    pt = np.zeros((n, 4), np.float32); pt[:, :3] = rng.uniform(-1, 1, (n, 3)); pt[:, 3] = S.FLT_MAX
    pt[::3, 3] = rng.uniform(0.1, 2.0, pt[::3].shape[0])
    d = rng.standard_normal((n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    df = np.zeros((n, 4), np.float32); df[:, :3] = d
    fl = np.zeros(n, np.uint32); fl[::3] = 1
    df[:, 3] = fl.view(np.float32)
    bo, bd, bh = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=n * 16)
    for mode in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, mode)
        capi.trace_rays(rt.triangle_buf, rt.bvh_buf, bo, bd, bh, n)
        hg = bh.read(np.float32).reshape(n, 4)
        ho = O.trace_rays(sc, pt, df)
        closest = fl == 0
        eq = (hg.view(np.uint32) == ho.view(np.uint32)).all(1)
        anyhit_same = ((hg.view(np.int32)[:, 3] >= 0) == (ho.view(np.int32)[:, 3] >= 0))
        print("  rays mode %d: closest bitwise mismatches %d / %d ; any-hit occlusion mismatches %d / %d" % (
            mode, int((~eq[closest]).sum()), int(closest.sum()), int((~anyhit_same[~closest]).sum()), int((~closest).sum())), flush=True)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    # accumulate + post
    rt.start()
    out = rt.raytrace(11)
    p = rt.params
    co, ao, no = O.kernel_raytrace(sc, p, 11, W, H, denoise_available=1)
    z = np.zeros_like(co)
    c2, a2, n2 = O.kernel_accumulate(0, 1, co, ao, no, z, z, z)
    po = O.kernel_post_process(p, c2)
    dd = np.abs(out.astype(int) - po.astype(int))
    print("  post-process: max |diff| %d, frac differing %.5f" % (dd.max(), (dd > 0).any(-1).mean()), flush=True)
print("done")
