"""Soak test 2: the whole per-sample schedule (kernel_raytrace -> accumulate -> copy -> post_process)
on the GPU against the CPU oracle, BITWISE, over random configurations: scene, camera, odd frame
sizes, seeds, depth caps (incl. unbounded), shading defaults, textures / normal maps / sky, feature
images, FXAA, exposure, several accumulated samples, tile partitions.
usage: gpu_soak_oracle.py [configs] [seconds budget]"""
import sys, os, time, dataclasses
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from oracle import bind as O

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
t_start = time.time()
capi.init(0)
O.set_threads(min(16, len(os.sched_getaffinity(0))))
rng = np.random.default_rng(77)


def textured(sc):
    mats = rng.uniform(0.05, 1.0, (4, 16, 32, 4)).astype(np.float32)
    mats[3, ..., :3] = 0.5 + 0.3 * rng.standard_normal((16, 32, 3)).astype(np.float32)
    meta = sc.meta.copy()
    n = meta.shape[0]
    uv = rng.uniform(-2, 3, (n, 3, 2)).astype(np.float32)
    meta["texture_coord1"], meta["texture_coord2"], meta["texture_coord3"] = uv[:, 0], uv[:, 1], uv[:, 2]
    tex = rng.uniform(size=n) < 0.7
    meta["diffuse_index"][tex] = 0
    meta["metallic_index"][tex & (rng.uniform(size=n) < 0.5)] = 1
    meta["roughness_index"][tex & (rng.uniform(size=n) < 0.5)] = 2
    meta["normal_index"][tex & (rng.uniform(size=n) < 0.5)] = 3
    t = rng.standard_normal((n, 3)).astype(np.float32)
    meta["tangent1"] = meta["tangent2"] = meta["tangent3"] = t
    meta["bitangent1"] = meta["bitangent2"] = meta["bitangent3"] = np.cross(meta["normal1"], t).astype(np.float32)
    return dataclasses.replace(sc, meta=meta, materials=mats, sky=S.analytic_sky(64, 32))


verts, meta, em = S.room_scene(n_objects=12, res=9)
scenes = {
    "cornell": (S.cornell_box(), (2.0, 5.0)),
    "knot96": (S.dragon_standin(96, 96), (1.2, 3.0)),
    "soup20k": (S.build_scene(S.soup(20000, seed=11, edge=0.06)), (1.2, 3.0)),
    "room": (S.build_scene(verts, meta=meta, emissive=em), (0.1, 0.8)),
}
scenes["knot96 textured"] = (textured(scenes["knot96"][0]), (1.2, 3.0))
names = list(scenes)
bad = 0
done = 0
for c in range(n_cfg):
    if time.time() - t_start > budget:
        break
    name = names[c % len(names)]
    sc, dist = scenes[name]
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 130))
    v = rng.standard_normal(3)
    v /= np.linalg.norm(v)
    cam = S.Camera(width=w, height=h, position=(v * rng.uniform(*dist)).tolist(), target=rng.uniform(-0.2, 0.2, 3).tolist(),
                   up=[0.0, 1.0, 0.0], fovy=float(rng.uniform(25, 80)))
    pt = bool(rng.uniform() < 0.85)
    aa = bool(rng.uniform() < 0.5)
    p = S.make_scene_params(cam, shading_diffuse=tuple(rng.uniform(0.2, 1.0, 3).tolist()), shading_metallic=float(rng.uniform(0, 1)),
                            shading_roughness=float(rng.uniform(0.02, 1)), path_tracing=pt, anti_aliasing=aa,
                            exposure=float(rng.uniform(0.3, 3.0)))
    depth = int(rng.choice([0, 1, 2, 3, 5, 8]))
    denoise = int(rng.uniform() < 0.5)
    samples = int(rng.integers(1, 6))
    seed = int(rng.integers(1, 2**31))
    rt = Raytracer(0, denoise_available=bool(denoise))
    rt.set_scene(sc, p, w, h)
    rt.start()
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    z = np.zeros((h, w, 4), np.float32)
    prev = [z, z, z]
    ok = True
    for s in range(samples):
        pixels = rt.raytrace(seed + s)
        ref = O.kernel_raytrace(sc, p, seed + s, w, h, denoise_available=denoise, max_depth=depth)
        got1 = [rt.temp_color_img1.read()] + ([rt.albedo_img1.read(), rt.normal_img1.read()] if denoise else [])
        for g, r in zip(got1, ref):
            ok &= bool((g.view(np.uint32) == r.view(np.uint32)).all())
        acc = O.kernel_accumulate(s, denoise, ref[0], ref[1], ref[2], prev[0], prev[1], prev[2])
        prev = list(acc)
        ok &= bool((rt.temp_color_img2.read().view(np.uint32) == acc[0].view(np.uint32)).all())
        if denoise:
            ok &= bool((rt.albedo_img2.read().view(np.uint32) == acc[1].view(np.uint32)).all())
            ok &= bool((rt.normal_img2.read().view(np.uint32) == acc[2].view(np.uint32)).all())
        ok &= bool((pixels == O.kernel_post_process(p, acc[0])).all())
    # the same samples as ONE call (nova_hip_render_samples: lanes, ping-pong planes, one post-process)
    last1 = ref
    rt.start()
    if denoise:
        rt.prev_albedo_img.fill([0.0] * 4)
        rt.prev_normal_img.fill([0.0] * 4)
    pixels = rt.raytrace_samples([seed + s for s in range(samples)])
    ok_f = bool((pixels == O.kernel_post_process(p, prev[0])).all())
    pairs = [(rt.temp_color_img1, last1[0]), (rt.temp_color_img2, prev[0]), (rt.prev_color_img, prev[0])]
    if denoise:
        pairs += [(rt.albedo_img1, last1[1]), (rt.normal_img1, last1[2]), (rt.albedo_img2, prev[1]), (rt.normal_img2, prev[2]),
                  (rt.prev_albedo_img, prev[1]), (rt.prev_normal_img, prev[2])]
    for img, r in pairs:
        ok_f &= bool((img.read().view(np.uint32) == r.view(np.uint32)).all())
    if not ok_f:
        print("  (the fused call differs)", flush=True)
    ok &= ok_f
    # the same frame as tiles of 3 ranks, sample 0
    whole = None
    if pt:
        rt.kernel_raytrace(seed)
        whole = rt.temp_color_img1.read().copy()
        merged = np.zeros_like(whole)
        for r in range(3):
            capi.set_option(capi.OPT_TILE_RANK, r)
            capi.set_option(capi.OPT_TILE_RANKS, 3)
            rt.temp_color_img1.fill([0.0, 0.0, 0.0, 0.0])
            rt.kernel_raytrace(seed)
            part = rt.temp_color_img1.read()
            m = (part.view(np.uint32) != 0).any(-1)
            merged[m] = part[m]
        capi.set_option(capi.OPT_TILE_RANK, 0)
        capi.set_option(capi.OPT_TILE_RANKS, 1)
        nz = (whole.view(np.uint32) != 0).any(-1)
        ok &= bool((merged[nz].view(np.uint32) == whole[nz].view(np.uint32)).all())
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH cfg %d: %s %dx%d pt %s aa %s depth %d denoise %d samples %d seed %d" % (c, name, w, h, pt, aa, depth, denoise, samples, seed), flush=True)
    elif c % 10 == 0:
        print("cfg %d ok (%s %dx%d depth %d, %.0f s)" % (c, name, w, h, depth, time.time() - t_start), flush=True)
print("configurations checked: %d, failing: %d" % (done, bad))
print("SOAK OK" if bad == 0 else "SOAK FAILED")
sys.exit(0 if bad == 0 else 1)
