"""GPU A/B: traversal kernel variants against the box-pair batch kernel (bitwise) + timing.
usage: gpu_check3.py [kernels comma list, default 0,6]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from tests.conftest import random_rays

# entries: kernel or kernel:tune (tune = NOVA_HIP_OPT_TUNE: grab size | flags << 16)
kernels = [tuple(int(y, 0) for y in (x + ":0").split(":")[:2]) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,6").split(",")]
small = "--no-small" not in sys.argv
capi.init(0)
capi.set_option(capi.OPT_TUNE, int(os.environ.get('NOVA_TUNE', '0')))
ok = True


def batch(sc, pt, df, kern, trav=0):
    n = pt.shape[0]
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    capi.set_option(capi.OPT_TRAVERSAL, trav)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, kern[0])
    capi.set_option(capi.OPT_TUNE, kern[1])
    capi.trace_rays(tb, bb, ro, rd, hb, n)
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
    capi.set_option(capi.OPT_TUNE, 0)
    return hb.read(np.uint32).reshape(n, 4)


# ---- small scenes: random rays incl. axis-parallel ones and bounded / any-hit rays ----------------
scenes = {
    "cornell": S.cornell_box(),
    "soup50k": S.build_scene(S.soup(50000, seed=4, edge=0.05)),
    "knot96": S.dragon_standin(96, 96),
    "onetri": S.build_scene(np.float32([[-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.6, 0]])),
}
rng = np.random.default_rng(0)
many = (rng.uniform(-0.3, 0.3, (300, 3, 3)) * [1, 1, 0.001]).reshape(300, 9).astype(np.float32)
scenes["bigleaf"] = S.build_scene(many)
for name, sc in (scenes.items() if small else []):
    pt, df = random_rays(200000, seed=31, anyhit_every=7, tmax_every=5)
    # axis-parallel directions (zero components, both signs of zero) -> box-pair path inside k_traverse4
    d = df[:, :3].copy()
    d[::11, 0] = 0.0
    d[::13, 1] = -0.0
    d[::17, 2] = 0.0
    d[::11 * 13, :2] = 0.0
    nz = np.linalg.norm(d, axis=1, keepdims=True)
    nz[nz == 0] = 1
    df[:, :3] = d / nz
    df[::11, 0] = 0.0
    df[::13, 1] = -0.0
    pt[::19, :3] = 0.0  # origins at 0: -o/d = -0 * inf = NaN on axis-parallel rays
    walk = batch(sc, pt, df, (0, 0), trav=1)
    anyhit = (df[:, 3].view(np.uint32) & 1) == 1
    for k in kernels:
        got = batch(sc, pt, df, k)
        same = (got[~anyhit] == walk[~anyhit]).all() and ((got[anyhit, 3].view(np.int32) >= 0) == (walk[anyhit, 3].view(np.int32) >= 0)).all()
        ok &= bool(same)
        print("%-8s kernel %s vs reference walk: %s (hits %.3f)" % (name, k, same, (walk[:, 3].view(np.int32) >= 0).mean()), flush=True)

# ---- full size: path tracing on the dragon stand-in ------------------------------------------------
sc = S.dragon_standin()
W, H = 1920, 1080
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
base = None
for k in kernels:
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, k[0])
    capi.set_option(capi.OPT_TUNE, k[1])
    rt.kernel_raytrace(3)
    best = None
    for rep in range(3):
        capi.reset_counters()
        t = time.time(); rt.kernel_raytrace(3); dt = time.time() - t
        c = capi.counters()
        if best is None or c["traverse_ms"] < best[0]:
            best = (c["traverse_ms"], c["shade_ms"], dt)
    rays = c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"]
    capi.set_option(capi.OPT_COUNT_WORK, 1)
    capi.reset_counters(); rt.kernel_raytrace(3); cw = capi.counters()
    capi.set_option(capi.OPT_COUNT_WORK, 0)
    img = rt.temp_color_img1.read()
    if base is None:
        base = img
    same = bool((img == base).all())
    ok &= same
    print("dragon kernel %s: traverse %.2f ms shade %.2f ms wall %.2f ms -> %.0f Mrays/s; node lines/ray %.1f tris/ray %.1f; bit-equal to first: %s" % (
        k, best[0], best[1], best[2] * 1e3, rays / best[2] / 1e6, cw["node_visits"] / rays, cw["triangle_tests"] / rays, same), flush=True)
capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
capi.set_option(capi.OPT_TUNE, 0)

# ---- incoherent batch on a 2M soup --------------------------------------------------------------
sc = S.build_scene(S.soup(2_000_000, seed=4))
n = 4_000_000
pt, df = random_rays(n, seed=5, lo=-1.0, hi=1.0)
tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
base = None
for k in kernels:
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, k[0])
    capi.set_option(capi.OPT_TUNE, k[1])
    capi.trace_rays(tb, bb, ro, rd, hb, n)
    capi.reset_counters()
    capi.trace_rays(tb, bb, ro, rd, hb, n)
    c = capi.counters()
    got = hb.read(np.uint32)
    if base is None:
        base = got
    same = bool((got == base).all())
    ok &= same
    print("soup2M kernel %s: traverse %.2f ms -> %.0f Mrays/s; bit-equal: %s" % (k, c["traverse_ms"], n / c["traverse_ms"] / 1e3, same), flush=True)
capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
capi.set_option(capi.OPT_TUNE, 0)
print("ALL OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
