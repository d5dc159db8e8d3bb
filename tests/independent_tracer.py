"""An independent second reading of the reference's path tracer, for checking the oracle.

Written from the reference's source alone (kernels/kernels.cu:9-40, raytrace.hpp:104-298, brdf.hpp,
light.hpp, material.hpp:15-62 without textures, transforms.hpp, random.hpp, intersection.hpp:14-37,
types.hpp), in pure Python float64, sharing no code with oracle/nova_oracle.cpp: different language,
different precision, brute force over all triangles instead of a BVH walk.  Random numbers are the
reference's (uint32 Wang hash + xorshift32, divided by 2^32 after rounding to float32), drawn in the
reference's order, so a pixel whose discrete decisions (hit / miss, lobe choice, light choice,
Russian roulette) come out the same must agree with the float32 oracle to ~1e-5; a misread formula
or a draw taken in another order shows up in (nearly) every pixel.

Only what the small test scenes need: no material textures, no normal maps, no box tests (the
Woop test of every triangle; on well-conditioned scenes the boxes change nothing).
"""
from __future__ import annotations

import math

import numpy as np

M32 = 0xFFFFFFFF
RAY_EPSILON = 1e-2
FLT_MAX = 3.4028234663852886e38
NON_ZERO = 1e-7


def wang_hash(s):  # random.hpp:11-18
    s = ((s ^ 61) ^ (s >> 16)) & M32
    s = (s * 9) & M32
    s = (s ^ (s >> 4)) & M32
    s = (s * 0x27D4EB2D) & M32
    s = (s ^ (s >> 15)) & M32
    return s


class Rng:  # random.hpp:21-26
    def __init__(self, state):
        self.s = state & M32

    def rand(self):
        s = self.s
        s ^= (s << 13) & M32
        s ^= s >> 17
        s ^= (s << 5) & M32
        self.s = s
        # uint -> float conversion rounds to nearest float32; (float)UINT_MAX is 2^32
        return float(np.float32(s)) / 4294967296.0


def v(*a):
    return np.array(a, np.float64)


def normalize(a):
    return a / math.sqrt(float(a @ a))


def nz(x):  # make_non_zero, transforms.hpp:50-56
    return np.maximum(x, NON_ZERO)


def create_basis(n):  # transforms.hpp:32-40: rows u, v(=normal), w
    vec = v(1.0, 0.0, 0.0) if abs(n[1]) > 0.1 else v(0.0, 1.0, 0.0)
    u = normalize(np.cross(n, vec))
    w = np.cross(u, n)
    return np.stack([u, n, w])


def spherical(theta, phi):  # transforms.hpp:46-48
    return v(math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))


class Brdf:  # brdf.hpp:20-181 (+ the light variant :183-203)
    def __init__(self, out_dir, normal, diffuse, metallic, roughness, in_dir=None):
        self.out_dir, self.normal, self.diffuse = out_dir, normal, diffuse
        self.metallic, self.roughness = metallic, roughness
        self.done = False
        if in_dir is not None:
            self.in_dir = in_dir
            self.m_normal = normalize(in_dir + out_dir)

    @staticmethod
    def fresnel(c, f0):
        return f0 + (1.0 - f0) * max(1.0 - c, 0.0) ** 5.0

    def sample(self, rng):
        xi0, xi1, xi2 = rng.rand(), rng.rand(), rng.rand()
        f0 = 0.04 + (self.diffuse - 0.04) * self.metallic  # mix(0.04, diffuse, metallic)
        o_dot_n = max(float(self.out_dir @ self.normal), 0.0)
        ks = self.fresnel(o_dot_n, f0)
        kd = (1.0 - ks) * (1.0 - self.metallic)
        ratio = np.clip(ks / (ks + kd), 0.0, 1.0)
        est_t = max(float(ratio.sum()) / 3.0, 0.25)
        basis_t = create_basis(self.normal).T  # transpose(create_basis(normal)) * vec
        phi = 2.0 * math.pi * xi1
        self.near_flip = abs(xi0 - est_t) < 1e-5
        if xi0 <= est_t:
            theta = math.atan(self.roughness * math.sqrt(xi2 / (1.0 - xi2))) if xi2 < 1.0 else math.pi / 2
            self.m_normal = basis_t @ spherical(theta, phi)
            i = -self.out_dir
            self.in_dir = i - 2.0 * float(self.m_normal @ i) * self.m_normal  # reflect(-out, m)
        else:
            theta = math.acos(math.sqrt(xi2))
            self.in_dir = basis_t @ spherical(theta, phi)
            self.m_normal = normalize(self.in_dir + self.out_dir)
        return self.in_dir

    def _inter(self):
        if self.done:
            return
        self.i_dot_m = float(self.in_dir @ self.m_normal)
        self.i_dot_n = float(self.in_dir @ self.normal)
        self.o_dot_n = max(float(self.out_dir @ self.normal), 0.0)
        self.n_dot_m = max(float(self.normal @ self.m_normal), 0.0)
        alpha = self.roughness * self.roughness
        a2 = alpha * alpha
        x = self.n_dot_m * self.n_dot_m * (a2 - 1.0) + 1.0
        self.D = a2 / max(math.pi * x * x, NON_ZERO)
        f0 = 0.04 + (self.diffuse - 0.04) * self.metallic
        self.F = self.fresnel(self.i_dot_m, f0)
        k = (self.roughness + 1.0) ** 2 / 8.0
        g = lambda c: c / max(c * (1.0 - k) + k, NON_ZERO)
        self.G = g(self.i_dot_n) * g(self.o_dot_n)
        self.k_d = (1.0 - self.F) * (1.0 - self.metallic)
        self.t = np.clip(self.F / (self.F + self.k_d), 0.0, 1.0)
        self.done = True

    def eval(self):
        self._inter()
        if self.i_dot_m <= 0.0 or self.i_dot_n <= 0.0:
            return v(0.0, 0.0, 0.0)
        return self.k_d * (self.diffuse / math.pi) + self.D * self.F * self.G / max(4.0 * self.o_dot_n * self.i_dot_n, NON_ZERO)

    def pdf(self):
        self._inter()
        return nz((1.0 - self.t) * self.i_dot_n / math.pi + self.t * self.D * self.n_dot_m / max(4.0 * self.i_dot_m, NON_ZERO))

    def light_pdf(self, l_normal, distance, area):
        return distance * distance / max(abs(float(l_normal @ self.in_dir)) * area, NON_ZERO)


def power_heuristic(p, s):  # brdf.hpp:205-210
    p2, s2 = p * p, s * s
    return p2 / (p2 + s2)


class Scene:
    def __init__(self, sc):
        t = sc.triangles
        self.M = np.stack([t["x"], t["y"], t["z"]], 1).astype(np.float64)  # [n, 3, 4]
        self.meta = sc.meta
        self.lights = sc.lights
        self.num_lights = sc.num_lights
        self.sky = sc.sky
        # set when a discrete decision of the current pixel was within rounding of flipping (a
        # float32 evaluation may then legitimately decide the other way): the caller skips the pixel
        self.ambiguous = False

    def find(self, origin, direction, length=FLT_MAX, anyhit=False):
        """intersection.hpp:14-37 on every triangle, in index order (ties -> lowest index)."""
        o = origin + direction * RAY_EPSILON  # types.hpp:16
        best = (length, -1, None)
        tol = 1e-4
        oz = self.M[:, 2, :3] @ o + self.M[:, 2, 3]
        dz = self.M[:, 2, :3] @ direction
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -oz / dz
        ok = np.isfinite(t) & (t >= 0.0)
        for i in np.flatnonzero(ok):
            ti = float(t[i])
            u = float(self.M[i, 0, :3] @ o + self.M[i, 0, 3]) + ti * float(self.M[i, 0, :3] @ direction)
            w = float(self.M[i, 1, :3] @ o + self.M[i, 1, 3]) + ti * float(self.M[i, 1, :3] @ direction)
            bary = (1.0 - u - w, u, w)
            if min(bary) > -tol and abs(ti - best[0]) <= tol * max(1.0, abs(ti)):
                # a tie with the ray's end (a shadow ray ending on the sampled light's own triangle) or
                # between two triangles; coplanar overlapping triangles with the same shading data (the
                # stand-in's ground plane has them) give the same result whichever wins
                if best[1] < 0 or self.meta[i].tobytes() != self.meta[best[1]].tobytes():
                    self.ambiguous = True
            if ti >= best[0]:
                continue
            if abs(min(bary)) <= tol:
                self.ambiguous = True
            if min(bary) < 0.0:
                continue
            best = (ti, int(i), bary)
            if anyhit:
                break
        return o, best


def read_sky(sky, d):
    """material.hpp:31-35 with the reference's float-coordinate sampler (kernels/backend/opencl/image.hpp:
    14-21: normalised coordinates, CLK_FILTER_LINEAR, CLK_ADDRESS_REPEAT) as OpenCL 1.2 §8.2 defines it."""
    h, w = sky.shape[:2]
    s = math.atan2(d[2], d[0]) / (2.0 * math.pi) + 0.5
    t = math.asin(max(-1.0, min(1.0, d[1]))) / math.pi + 0.5
    u = (s - math.floor(s)) * w
    vv = (t - math.floor(t)) * h
    i0, j0 = math.floor(u - 0.5), math.floor(vv - 0.5)
    a, b = (u - 0.5) - i0, (vv - 0.5) - j0
    i0, i1, j0, j1 = i0 % w, (i0 + 1) % w, j0 % h, (j0 + 1) % h
    tex = lambda i, j: np.asarray(sky[j, i, :3], np.float64)
    return (1 - a) * (1 - b) * tex(i0, j0) + a * (1 - b) * tex(i1, j0) + (1 - a) * b * tex(i0, j1) + a * b * tex(i1, j1)


def light_geometry(L):  # light.hpp:12-44
    a, b, c = (np.asarray(L[k], np.float64) for k in ("a", "b", "c"))
    if int(L["type"]) == 0:  # RECT: position, normal, dims
        return "rect", normalize(b), float(c[0] * c[1]), (a, b, c)
    n = np.cross(b - a, c - a)
    return "tri", normalize(n), 0.5 * math.sqrt(float(n @ n)), (a, b, c)


def sample_light(kind, geo, rng):
    a, b, c = geo
    if kind == "rect":
        r = v(rng.rand(), rng.rand())
        off = c[:2] * (r * 2.0 - 1.0)
        basis = create_basis(normalize(b))  # light_basis * vec = rows dotted with vec
        return a + basis @ v(off[0], 0.0, off[1])
    r1 = math.sqrt(rng.rand())
    r2 = rng.rand()
    return (1.0 - r1) * a + r1 * (1.0 - r2) * b + r1 * r2 * c


def trace_pixel(S, P, px, py, width, time, max_depth=0):
    """kernel_raytrace + trace_ray for one pixel.  Returns (color, n_segments, decisions, ambiguous,
    optional_terms): `color` without the optional light-sample terms (see the shadow ray below),
    `decisions` the discrete outcomes (hit ids, light choices, roulette), `ambiguous` whether any
    other decision was within rounding of flipping."""
    rng = Rng(wang_hash((py * width + px + wang_hash(time)) & M32))
    S.ambiguous = False
    jx, jy = rng.rand(), rng.rand()
    cs, cd = np.asarray(P["coord_scale"], np.float64), np.asarray(P["coord_dims"], np.float64)
    ab = cs * (v(px, py) - cd + v(jx, jy))
    frame = np.stack([np.asarray(P[k], np.float64) for k in ("frame_x", "frame_y", "frame_z")])
    ray_dir = normalize(frame.T @ v(ab[0], -ab[1], -1.0))
    ray_pos = np.asarray(P["eye_pos"], np.float64)
    color = v(0.0, 0.0, 0.0)
    thr = v(1.0, 1.0, 1.0)
    direct = True
    dec = []
    optional_terms = []
    seg = 0
    while True:
        o, (t, tri, bary) = S.find(ray_pos, ray_dir)
        dec.append(tri)
        if tri < 0:
            if direct:
                color = read_sky(S.sky, ray_dir)
            break
        seg += 1
        m = S.meta[tri]
        point = o + ray_dir * t
        kD = np.asarray(m["kD"], np.float64)
        diffuse = np.asarray(P["shading_diffuse"], np.float64) * kD
        metallic = float(P["shading_metallic"]) if float(m["metallic"]) == -1.0 else float(m["metallic"])
        roughness = float(P["shading_roughness"]) if float(m["roughness"]) == -1.0 else float(m["roughness"])
        li_here = int(m["light_index"])
        if not int(P["path_tracing"]):
            color = color + (np.asarray(S.lights[li_here]["intensity"], np.float64) if li_here != -1 else diffuse)
            break
        normal = normalize(sum(bary[k] * np.asarray(m["normal%d" % (k + 1)], np.float64) for k in range(3)))
        out_dir = -ray_dir
        if direct and li_here != -1:
            color = color + thr * np.asarray(S.lights[li_here]["intensity"], np.float64)
        direct_color = v(0.0, 0.0, 0.0)
        if S.num_lights > 1 or (S.num_lights == 1 and li_here == -1):
            while True:
                lr = rng.rand() * S.num_lights
                S.ambiguous |= abs(lr - round(lr)) < 1e-5
                li = min(int(lr), S.num_lights - 1)
                if li != li_here:
                    break
            dec.append(("light", li))
            L = S.lights[li]
            kind, l_normal, l_area, geo = light_geometry(L)
            l_pos = sample_light(kind, geo, rng)
            l_dir = normalize(l_pos - point)
            l_dist = math.sqrt(float((l_pos - point) @ (l_pos - point)))
            # The shadow ray ends, up to rounding, ON the sampled light's own triangle
            # (raytrace.hpp:216-220: length = distance - epsilon from an origin moved by epsilon), so
            # `t >= length` there is decided by the last bit — in the reference too.  Occlusion by
            # anything nearer is decided with a shortened ray; if nothing nearer blocks, the term is
            # recorded as optional and the caller accepts either outcome.
            keep = S.ambiguous
            _, (_, occ, _) = S.find(point, l_dir, l_dist - RAY_EPSILON - 1e-3, anyhit=True)
            near_amb = S.ambiguous
            S.ambiguous = keep or near_amb
            optional = False
            if occ < 0:
                keep = S.ambiguous
                _, (_, occ_full, _) = S.find(point, l_dir, l_dist - RAY_EPSILON, anyhit=True)
                optional = S.ambiguous != keep or occ_full >= 0
                S.ambiguous = keep
            dec.append(("shadow", occ >= 0, optional))
            if occ < 0:
                b = Brdf(out_dir, normal, diffuse, metallic, roughness, in_dir=l_dir)
                f = b.eval()
                lp = b.light_pdf(l_normal, l_dist, l_area)
                w = power_heuristic(v(lp, lp, lp), b.pdf())
                term = S.num_lights * w * np.asarray(L["intensity"], np.float64) * f / lp * max(float(normal @ l_dir), 0.0)
                if optional:
                    optional_terms.append(thr * term)
                else:
                    direct_color = direct_color + term
            b = Brdf(out_dir, normal, diffuse, metallic, roughness, in_dir=v(0.0, 0.0, 0.0) + 0.0 * out_dir)
            b.m_normal = normalize(out_dir)  # normalize({} + out_dir), overwritten by sample()
            m_dir = b.sample(rng)
            S.ambiguous |= b.near_flip
            mo, (mt, mtri, _) = S.find(point, m_dir)
            dec.append(("mis", mtri))
            if mtri >= 0:
                lm = int(S.meta[mtri]["light_index"])
                if lm != -1 and lm != li_here:
                    L2 = S.lights[lm]
                    _, l_normal2, l_area2, _ = light_geometry(L2)
                    f = b.eval()
                    lp = b.light_pdf(l_normal2, mt, l_area2)
                    bp = b.pdf()
                    w = power_heuristic(bp, v(lp, lp, lp))
                    direct_color = direct_color + w * np.asarray(L2["intensity"], np.float64) * f / bp * max(float(normal @ m_dir), 0.0)
        color = color + thr * direct_color
        if max_depth and seg >= max_depth:  # the backend's segment cap, not in the reference
            break
        b = Brdf(out_dir, normal, diffuse, metallic, roughness)
        in_dir = b.sample(rng)
        S.ambiguous |= b.near_flip
        f = b.eval()
        p = b.pdf()
        thr = thr * f / p * max(float(normal @ in_dir), 0.0)
        if not direct:
            q = min(float(thr.max()), 1.0)
            r = rng.rand()
            S.ambiguous |= abs(r - q) < 1e-5
            dec.append(("rr", r > q))
            if r > q:
                break
            thr = thr * (1.0 / q)
        ray_pos, ray_dir = point, in_dir
        direct = False
    return color, seg, tuple(dec), S.ambiguous, optional_terms


def matches(color, optional_terms, ref, tol=1e-4):
    """Is `ref` equal (relative tol) to `color` plus some subset of the optional terms?"""
    import itertools
    scale = max(float(np.abs(ref).max()), 1e-3)
    for k in range(len(optional_terms) + 1):
        for sub in itertools.combinations(optional_terms, k):
            c = color + (sum(sub) if sub else 0.0)
            if float(np.abs(c - ref).max()) <= tol * scale:
                return True
    return False
