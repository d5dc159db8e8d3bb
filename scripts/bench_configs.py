"""The other BASELINE.json configurations at full size on one GPU: parity properties
(production traversal == reference walk, bitwise) and throughput.  Not bench.py lines — the numbers go
into DESIGN.md §7.  usage: bench_configs.py [c1,c3,c4,c5]"""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer

which = (sys.argv[1] if len(sys.argv) > 1 else "c1,c3,c4,c5").split(",")
capi.init(0)


def frames(rt, n, depth, label):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    rt.start()
    rt.raytrace(1, read_back=False)
    capi.reset_counters()
    capi.synchronize()
    t = time.perf_counter()
    for i in range(n):
        rt.raytrace(2 + i, read_back=False)
    capi.synchronize()
    dt = (time.perf_counter() - t) / n
    c = capi.counters()
    rays = (c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"] - c.get("rays_mis_pretested", 0) - c.get("rays_shadow_skipped", 0)) / n
    print("%s: %.2f ms/frame, %.2f Mrays/frame -> %.0f Mrays/s (traverse %.2f ms, shade %.2f ms per frame)" % (
        label, dt * 1e3, rays / 1e6, rays / dt / 1e6, c["traverse_ms"] / n, c["shade_ms"] / n), flush=True)
    # the same frames through nova_hip_render_samples: n samples per call
    m = max(n, 6)
    rt.start()
    rt.raytrace_samples([2 + i for i in range(m)], read_back=False)
    rt.start()
    capi.reset_counters()
    capi.synchronize()
    t = time.perf_counter()
    rt.raytrace_samples([2 + i for i in range(m)], read_back=False)
    capi.synchronize()
    fdt = (time.perf_counter() - t) / m
    c = capi.counters()
    frays = (c["rays_primary"] + c["rays_bounce"] + c["rays_shadow"] - c.get("rays_mis_pretested", 0) - c.get("rays_shadow_skipped", 0)) / m
    print("%s, %d samples per call: %.2f ms/sample -> %.0f Mrays/s" % (label, m, fdt * 1e3, frays / fdt / 1e6), flush=True)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)


def walk_equals_production(rt, depth, label):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    out = []
    for trav in (0, 1):
        capi.set_option(capi.OPT_TRAVERSAL, trav)
        rt.kernel_raytrace(7)
        out.append(rt.temp_color_img1.read())
    capi.set_option(capi.OPT_TRAVERSAL, 0)
    capi.set_option(capi.OPT_MAX_DEPTH, 0)
    same = bool((out[0] == out[1]).all())
    print("%s: production == reference walk, every pixel bitwise: %s" % (label, same), flush=True)
    return same


ok = True
if "c1" in which:  # Cornell box 512^2, 1 spp, primary + shadow (path tracing with one segment)
    sc = S.cornell_box()
    w = h = 512
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    ok &= walk_equals_production(rt, 1, "C1 cornell 512^2")
    frames(rt, 20, 1, "C1 cornell 512^2, 1 segment")

if "c3" in which:  # fireplace_room stand-in, 3840x2160
    sc = S.room_standin()
    w, h = 3840, 2160
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.ROOM_CAMERA), **S.ROOM_SHADING)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    print("C3 room stand-in: %d triangles, %d lights" % (sc.triangles.shape[0], sc.num_lights))
    ok &= walk_equals_production(rt, 5, "C3 room 3840x2160")
    frames(rt, 4, 5, "C3 room 3840x2160, max_depth 5")
    del rt

if "c4" in which:  # 10 M-triangle soup, 16.7 M incoherent closest-hit rays per batch
    t = time.time()
    sc = S.build_scene(S.soup(10_000_000, seed=4))
    print("C4 soup: %d triangles, %d BVH nodes, host build %.0f s" % (sc.triangles.shape[0], sc.bvh.shape[0], time.time() - t), flush=True)
    n = 1 << 24
    rng = np.random.default_rng(5)
    pt = np.zeros((n, 4), np.float32)
    pt[:, :3] = rng.uniform(-1, 1, (n, 3))
    pt[:, 3] = S.FLT_MAX
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    df = np.zeros((n, 4), np.float32)
    df[:, :3] = d
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    for rep in range(2):  # second pass: rays leave the surface points the first pass hit
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        capi.reset_counters()
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        ms = capi.counters()["traverse_ms"]
        hits = hb.read(np.float32).reshape(n, 4)
        hit = hits[:, 3].view(np.int32) >= 0
        print("C4 pass %d: %d rays in %.2f ms -> %.0f Mrays/s (hit rate %.3f)" % (rep, n, ms, n / ms / 1e3, hit.mean()), flush=True)
        if rep == 0:
            m = 1 << 20  # the reference walk on the first 1 M rays
            capi.set_option(capi.OPT_TRAVERSAL, 1)
            hb2 = capi.Buffer(nbytes=16 * m)
            capi.trace_rays(tb, bb, ro, rd, hb2, m)
            capi.set_option(capi.OPT_TRAVERSAL, 0)
            same = bool((hb2.read(np.uint32) == hits[:m].view(np.uint32).reshape(-1)).all())
            ok &= same
            print("C4: production == reference walk on %d rays: %s" % (m, same), flush=True)
            pt[hit, :3] = pt[hit, :3] + d[hit] * (hits[hit, 0:1] - 1e-3)
            d2 = rng.standard_normal((n, 3)).astype(np.float32)
            d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
            df[:, :3] = d2
            d = d2
            ro, rd = capi.Buffer(pt), capi.Buffer(df)

if "c5" in which:  # textured shading path at 3840x2160 (synthetic textures + analytic sky)
    sc = S.dragon_standin()
    rng = np.random.default_rng(8)
    mats = rng.uniform(0.05, 1.0, (4, 1024, 1024, 4)).astype(np.float32)
    mats[3, ..., :3] = 0.5 + 0.2 * rng.standard_normal((1024, 1024, 3)).astype(np.float32)
    meta = sc.meta.copy()
    n = meta.shape[0]
    uv = rng.uniform(0, 1, (n, 3, 2)).astype(np.float32)
    meta["texture_coord1"], meta["texture_coord2"], meta["texture_coord3"] = uv[:, 0], uv[:, 1], uv[:, 2]
    meta["diffuse_index"][:] = 0
    meta["metallic_index"][:] = 1
    meta["roughness_index"][:] = 2
    meta["normal_index"][:] = 3
    tvec = rng.standard_normal((n, 3)).astype(np.float32)
    meta["tangent1"] = meta["tangent2"] = meta["tangent3"] = tvec
    meta["bitangent1"] = meta["bitangent2"] = meta["bitangent3"] = np.cross(meta["normal1"], tvec).astype(np.float32)
    sc = dataclasses.replace(sc, meta=meta, materials=mats, sky=S.analytic_sky(4096, 2048))
    w, h = 3840, 2160
    p = S.make_scene_params(S.Camera(width=w, height=h, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.3)
    rt = Raytracer(0)
    rt.set_scene(sc, p, w, h)
    frames(rt, 4, 5, "C5 textured stand-in 3840x2160, max_depth 5")

print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
