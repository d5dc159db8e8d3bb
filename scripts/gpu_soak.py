"""Soak test of the exactness claim: on several scenes, for many random cameras, seeds and
ray batches, the production traversal (four-wide tree, ordered, distance-culled, forking, packets) must
give the bits of the reference-order exhaustive walk (NOVA_HIP_OPT_TRAVERSAL = 1).
usage: gpu_soak.py [frames per scene] [seconds budget]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from tests.conftest import random_rays

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 900.0
t_start = time.time()
capi.init(0)
rng = np.random.default_rng(2026)
bad = 0
checked_pixels = 0
checked_rays = 0


def frames(name, sc, w, h, depth, metallic, roughness, dist=(1.2, 3.0)):
    global bad, checked_pixels, n_frames
    rt = Raytracer(0)
    first = True
    for f in range(n_frames):
        if time.time() - t_start > budget:
            print("%s: time budget reached after %d frames" % (name, f), flush=True)
            break
        # camera on a random direction around the scene, looking near the centre
        v = rng.standard_normal(3)
        v /= np.linalg.norm(v)
        pos = (v * rng.uniform(*dist)).tolist()
        target = rng.uniform(-0.2, 0.2, 3).tolist()
        cam = S.Camera(width=w, height=h, position=pos, target=target, up=[0.0, 1.0, 0.0], fovy=float(rng.uniform(25, 70)))
        p = S.make_scene_params(cam, shading_metallic=metallic, shading_roughness=roughness)
        if first:
            rt.set_scene(sc, p, w, h)
            first = False
        else:
            rt.params = np.ascontiguousarray(p)
        seed = int(rng.integers(1, 2**31))
        capi.set_option(capi.OPT_MAX_DEPTH, depth)
        out = []
        for trav in (0, 1):
            capi.set_option(capi.OPT_TRAVERSAL, trav)
            rt.kernel_raytrace(seed)
            out.append(rt.temp_color_img1.read().view(np.uint32).copy())
        capi.set_option(capi.OPT_TRAVERSAL, 0)
        capi.set_option(capi.OPT_MAX_DEPTH, 0)
        diff = int((out[0] != out[1]).any(-1).sum())
        checked_pixels += w * h
        if diff:
            bad += 1
            print("%s frame %d seed %d: %d pixels differ" % (name, f, seed, diff), flush=True)
    print("%s: %d frames of %dx%d at max_depth %d done (%.0f s)" % (name, n_frames, w, h, depth, time.time() - t_start), flush=True)


def batches(name, sc, n, reps):
    global bad, checked_rays
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    for rep in range(reps):
        if time.time() - t_start > budget:
            break
        pt, df = random_rays(n, seed=int(rng.integers(1, 2**31)), anyhit_every=7, tmax_every=5)
        d = df[:, :3].copy()
        d[::11, 0] = 0.0
        d[::13, 1] = -0.0
        d[::17, 2] = 0.0
        nz = np.linalg.norm(d, axis=1, keepdims=True)
        nz[nz == 0] = 1
        df[:, :3] = d / nz
        df[::11, 0] = 0.0
        df[::13, 1] = -0.0
        pt[::19, :3] = 0.0
        ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
        res = []
        for trav in (0, 1):
            capi.set_option(capi.OPT_TRAVERSAL, trav)
            capi.trace_rays(tb, bb, ro, rd, hb, n)
            res.append(hb.read(np.uint32).reshape(n, 4).copy())
        capi.set_option(capi.OPT_TRAVERSAL, 0)
        anyhit = (df[:, 3].view(np.uint32) & 1) == 1
        diff = int((res[0][~anyhit] != res[1][~anyhit]).any(-1).sum())
        diff += int(((res[0][anyhit, 3].view(np.int32) >= 0) != (res[1][anyhit, 3].view(np.int32) >= 0)).sum())
        checked_rays += n
        if diff:
            bad += 1
            print("%s batch %d: %d rays differ" % (name, rep, diff), flush=True)
    print("%s: %d batches of %d rays done (%.0f s)" % (name, reps, n, time.time() - t_start), flush=True)


sc = S.dragon_standin()
frames("knot 871k", sc, 960, 540, 5, 0.4, 0.1)
batches("knot 871k", sc, 1_000_000, 6)
verts, meta, em = S.room_scene()
sc = S.build_scene(verts, meta=meta, emissive=em)
frames("room 142k", sc, 960, 540, 6, 0.1, 0.6, dist=(0.1, 0.8))
batches("room 142k", sc, 1_000_000, 6)
sc = S.cornell_box()
frames("cornell", sc, 512, 512, 8, 0.5, 0.1, dist=(2.0, 5.0))
batches("cornell", sc, 500_000, 6)
sc = S.build_scene(S.soup(500_000, seed=9, edge=0.03))
frames("soup 500k", sc, 640, 360, 4, 0.2, 0.3)
batches("soup 500k", sc, 1_000_000, 6)
# degenerate trees: one triangle; 300 coplanar triangles in one leaf (chained 128-triangle chunks)
batches("one triangle", S.build_scene(np.float32([[-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.6, 0]])), 200_000, 3)
many = (np.random.default_rng(0).uniform(-0.3, 0.3, (300, 3, 3)) * [1, 1, 0.001]).reshape(300, 9).astype(np.float32)
batches("big leaf", S.build_scene(many), 200_000, 3)
# cameras far outside the scene box (cull slack scaled by |origin| / radius) and inside the geometry
sc = S.dragon_standin(200, 200)
n_keep = n_frames
n_frames = max(4, n_frames // 4)
frames("knot 80k, far cameras", sc, 640, 360, 5, 0.4, 0.1, dist=(4.0, 40.0))
frames("knot 80k, cameras inside", sc, 640, 360, 5, 0.9, 0.05, dist=(0.0, 0.9))
n_frames = n_keep
print("checked %d pixels and %d rays; failing frames/batches: %d" % (checked_pixels, checked_rays, bad))
print("SOAK OK" if bad == 0 else "SOAK FAILED")
sys.exit(0 if bad == 0 else 1)
