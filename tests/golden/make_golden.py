"""Generates the vectors under tests/golden/ from oracle/libnova_oracle.so, the CPU restatement of
Nova's hot path (oracle/nova_oracle.cpp).  PARITY UNPINNED: the reference's kernels cannot be built or
run in this image and its own tests hold no vectors for them, so these files are
regression vectors of the restatement — they freeze its behaviour, so that the oracle, the HIP path
and later rounds are all held to the same numbers — not outputs of the reference.

The vectors hold inputs and expected outputs only:
  kat.npz               hash, rand, intersects_aabb, CookTorranceBRDF sample/eval/pdf
  ids_*.npz             primary closest-hit triangle id per pixel (kD-encoding trick, see
                        nova_amd.scene.encode_ids_in_kd) for Cornell box 512^2, a 50k-triangle soup
                        256^2 and the small knot scene 256^2
  rays_*.npz            find_intersection on random ray batches (closest / any-hit / bounded)
  cornell_radiance.npz  kernel_raytrace + kernel_accumulate radiance after 1 and 16 samples at 128^2,
                        first-sample albedo/normal features, kernel_post_process bytes without and
                        with FXAA, SHA-256 of the 512^2 one-sample radiance plane
  fxaa_edge.npz         kernel_post_process with FXAA on a 32^2 step-edge image
Every scene file also stores the SHA-256 of the scene arrays it was rendered from, so a test can
tell a drifted scene builder from a wrong kernel.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from nova_amd import scene as S  # noqa: E402
from oracle import bind as R  # noqa: E402
from tests.conftest import random_rays  # noqa: E402

SEED_TIME = 1


def scene_hash(sc):
    h = hashlib.sha256()
    for a in (sc.bvh, sc.triangles, sc.meta, sc.lights):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def primary_ids(sc, camera_kw, w, h, time):
    cam = S.Camera(width=w, height=h, **camera_kw)
    p = S.make_scene_params(cam, path_tracing=False)
    neg = S.constant_sky((-1, -1, -1), w=sc.sky.shape[1], h=sc.sky.shape[0])
    c = R.kernel_raytrace(sc, p, time, w, h, meta=S.encode_ids_in_kd(sc.meta), sky=neg)[0]
    return S.decode_ids(c).reshape(h, w)


def main():
    L = R.lib()
    rng = np.random.default_rng(2026)

    # ---- known answers of the small device functions --------------------------------------------
    seeds = rng.integers(0, 2 ** 32, 64, dtype=np.uint64).astype(np.uint32)
    seeds[:4] = [0, 1, 61, 0xFFFFFFFF]
    hashes = np.array([L.oracle_hash(int(s)) for s in seeds], np.uint32)
    rand_states = np.array([1, 12345, 0x9E3779B9, 0xFFFFFFFF], np.uint32)
    rand_vals = np.zeros((4, 8), np.float32)
    rand_after = np.zeros((4, 8), np.uint32)
    import ctypes
    for i, s0 in enumerate(rand_states):
        st = np.array([s0], np.uint32)
        for k in range(8):
            rand_vals[i, k] = L.oracle_rand(ctypes.c_void_p(st.ctypes.data))
            rand_after[i, k] = st[0]
    n = 1000
    ab_p = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    ab_d = rng.standard_normal((n, 3)).astype(np.float32)
    ab_d /= np.linalg.norm(ab_d, axis=1, keepdims=True).astype(np.float32)
    ab_d[::50, 0] = 0.0  # axis-parallel rays: 1/0 = inf, NaN slab bounds
    c0 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    c1 = c0 + rng.uniform(0, 1, (n, 3)).astype(np.float32)
    # keep the arrays alive and contiguous per row
    ab_res = np.zeros(n, np.int32)
    for i in range(n):
        rows = [np.ascontiguousarray(a[i]) for a in (ab_p, ab_d, c1, c0)]
        ab_res[i] = L.oracle_intersects_aabb(*[ctypes.c_void_p(r.ctypes.data) for r in rows])
    br_out = rng.standard_normal((n, 3)).astype(np.float32)
    br_out /= np.linalg.norm(br_out, axis=1, keepdims=True).astype(np.float32)
    br_n = br_out + 0.6 * rng.standard_normal((n, 3)).astype(np.float32)
    br_n /= np.linalg.norm(br_n, axis=1, keepdims=True).astype(np.float32)
    br_dif = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    br_met = rng.uniform(0, 1, n).astype(np.float32)
    br_rough = rng.uniform(0.01, 1, n).astype(np.float32)
    br_rng = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    br_res = np.zeros((n, 9), np.float32)
    br_rng_after = br_rng.copy()
    for i in range(n):
        o, nn, dd = (np.ascontiguousarray(a[i]) for a in (br_out, br_n, br_dif))
        st = br_rng_after[i:i + 1]
        out = np.zeros(9, np.float32)
        L.oracle_brdf(ctypes.c_void_p(o.ctypes.data), ctypes.c_void_p(nn.ctypes.data),
                   ctypes.c_void_p(dd.ctypes.data), float(br_met[i]), float(br_rough[i]),
                   ctypes.c_void_p(st.ctypes.data), ctypes.c_void_p(out.ctypes.data))
        br_res[i] = out
    save("kat.npz", hash_in=seeds, hash_out=hashes, rand_state=rand_states, rand_val=rand_vals,
         rand_after=rand_after, aabb_point=ab_p, aabb_dir=ab_d, aabb_top=c1, aabb_bottom=c0,
         aabb_hit=ab_res, brdf_out_dir=br_out, brdf_normal=br_n, brdf_diffuse=br_dif,
         brdf_metallic=br_met, brdf_roughness=br_rough, brdf_rng=br_rng, brdf_rng_after=br_rng_after,
         brdf_result=br_res)

    # ---- scenes ------------------------------------------------------------------------------------
    cornell = S.cornell_box()
    soup = S.build_scene(S.soup(50000, seed=4, edge=0.05))
    knot = S.dragon_standin(96, 96)
    soup_cam = dict(position=[0, 0, 3.5], target=[0, 0, 0], up=[0, 1, 0], fovy=40.0)

    save("ids_cornell_512.npz", ids=primary_ids(cornell, S.CORNELL_CAMERA, 512, 512, SEED_TIME),
         time=SEED_TIME, scene_sha256=scene_hash(cornell))
    save("ids_soup_256.npz", ids=primary_ids(soup, soup_cam, 256, 256, SEED_TIME), time=SEED_TIME,
         scene_sha256=scene_hash(soup))
    save("ids_knot_256.npz", ids=primary_ids(knot, S.DRAGON_CAMERA, 256, 256, SEED_TIME), time=SEED_TIME,
         scene_sha256=scene_hash(knot))

    for name, sc in (("soup", soup), ("knot", knot), ("cornell", cornell)):
        pt, df = random_rays(20000, seed=11, anyhit_every=3, tmax_every=4)
        hits = R.trace_rays(sc, pt, df)
        save(f"rays_{name}.npz", point_tmax=pt, dir_flags=df, hits_bits=hits.view(np.uint32),
             scene_sha256=scene_hash(sc))

    # ---- Cornell radiance: the reference's per-sample schedule, restated (src/core/raytracer.cpp:169-262) ----
    w = h = 128
    cam = S.Camera(width=w, height=h, **S.CORNELL_CAMERA)
    p = S.make_scene_params(cam, path_tracing=True)
    prev = [np.zeros((h, w, 4), np.float32) for _ in range(3)]
    first = None
    for s in range(16):
        c1_, a1, n1 = R.kernel_raytrace(cornell, p, SEED_TIME + s, w, h, denoise_available=1)
        c2, a2, n2 = R.kernel_accumulate(s, 1, c1_, a1, n1, *prev)
        prev = [c2, a2, n2]
        if s == 0:
            first = (c2.copy(), a2.copy(), n2.copy())
    px = R.kernel_post_process(p, prev[0])
    p_aa = S.make_scene_params(cam, path_tracing=True, anti_aliasing=True)
    px_aa = R.kernel_post_process(p_aa, prev[0])
    big = S.Camera(width=512, height=512, **S.CORNELL_CAMERA)
    c512 = R.kernel_raytrace(cornell, S.make_scene_params(big, path_tracing=True), SEED_TIME, 512, 512)[0]
    save("cornell_radiance.npz", color_1spp=first[0], albedo_1spp=first[1], normal_1spp=first[2],
         color_16spp=prev[0], albedo_16spp=prev[1], normal_16spp=prev[2], pixels_16spp=px,
         pixels_16spp_fxaa=px_aa, sha256_color_512_1spp=hashlib.sha256(c512.tobytes()).hexdigest(),
         time=SEED_TIME, scene_sha256=scene_hash(cornell))

    # ---- FXAA on a synthetic step edge -----------------------------------------------------------
    img = np.zeros((32, 32, 4), np.float32)
    yy, xx = np.mgrid[0:32, 0:32]
    img[..., :3] = np.where((xx * 0.7 + yy * 0.45) > 17.0, 0.9, 0.08)[..., None]
    img[..., 1] *= 0.8
    img[..., 3] = 1.0
    cam32 = S.Camera(width=32, height=32, **S.CORNELL_CAMERA)
    save("fxaa_edge.npz", image=img,
         pixels_fxaa=R.kernel_post_process(S.make_scene_params(cam32, anti_aliasing=True, exposure=1.5), img),
         pixels_plain=R.kernel_post_process(S.make_scene_params(cam32, anti_aliasing=False, exposure=1.5), img))


if __name__ == "__main__":
    main()
