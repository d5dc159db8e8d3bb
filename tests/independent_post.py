"""A second, independent reading of the reference's image kernels — kernel_accumulate,
kernel_post_process and fxaa — in plain Python float64 (test infrastructure only).

Written from the reference's source text alone (kernels/kernels.cu:42-116, kernels/anti_aliasing.hpp:
14-150, kernels/transforms.hpp:14-44, kernels/constants.hpp:13-16, the float sampler declared in
kernels/backend/opencl/image.hpp:14-21 with OpenCL 1.2 §8.2's filter equations), not from oracle/.
It shares nothing with the oracle but the inputs.  Because it computes in float64 and the kernels in
float32, every DISCRETE decision on the way (a comparison, a truncation to a byte) is recorded with
its margin: the caller skips the pixels whose smallest margin is within rounding.
"""
import math

EDGE_THRESHOLD_MIN = 0.0312      # constants.hpp:13
EDGE_THRESHOLD_MAX = 0.125       # constants.hpp:14
EDGE_SEARCH_ITERATIONS = 12      # constants.hpp:15
SUBPIXEL_QUALITY = 0.75          # constants.hpp:16
STEP_MOD = (1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 2.0, 2.0, 2.0, 2.0, 4.0, 8.0)  # anti_aliasing.hpp:99-100


class Margin:
    """Smallest distance of any decision taken so far from the point where it would flip."""

    def __init__(self):
        self.value = math.inf

    def ge(self, a, b):
        self.value = min(self.value, abs(a - b))
        return a >= b

    def lt(self, a, b):
        self.value = min(self.value, abs(a - b))
        return a < b


def texel(img, i, j):
    """CLK_ADDRESS_CLAMP: outside the image reads the border colour, (0,0,0,0) for an RGBA image."""
    h, w = img.shape[:2]
    if i < 0 or j < 0 or i >= w or j >= h:
        return (0.0, 0.0, 0.0)
    p = img[j, i]
    return (float(p[0]), float(p[1]), float(p[2]))


def sample(img, s, t):
    """CLK_NORMALIZED_COORDS_TRUE | CLK_FILTER_LINEAR | CLK_ADDRESS_CLAMP (OpenCL 1.2 §8.2):
    u = s*w, i0 = floor(u - 0.5), a = frac(u - 0.5); T = (1-a)(1-b)T00 + a(1-b)T10 + (1-a)b T01 + ab T11."""
    h, w = img.shape[:2]
    u, v = s * w, t * h
    i0, j0 = math.floor(u - 0.5), math.floor(v - 0.5)
    a, b = (u - 0.5) - i0, (v - 0.5) - j0
    t00, t10, t01, t11 = texel(img, i0, j0), texel(img, i0 + 1, j0), texel(img, i0, j0 + 1), texel(img, i0 + 1, j0 + 1)
    return tuple((1 - a) * (1 - b) * t00[c] + a * (1 - b) * t10[c] + (1 - a) * b * t01[c] + a * b * t11[c]
                 for c in range(3))


def luma(c):  # transforms.hpp:28-30
    return 0.299 * c[0] + 0.587 * c[1] + 0.114 * c[2]


def coords_to_uv(x, y, w, h):  # transforms.hpp:42-44
    return (x + 0.5) / w, (y + 0.5) / h


def fxaa(img, inv, uv, m):  # anti_aliasing.hpp:14-150
    def lum(du, dv):
        return luma(sample(img, uv[0] + du * inv[0], uv[1] + dv * inv[1]))

    center_color = sample(img, uv[0], uv[1])
    center = luma(center_color)
    down, left, right, up = lum(0, -1), lum(-1, 0), lum(1, 0), lum(0, 1)
    luma_max = max(center, up, left, right, down)
    luma_min = min(center, up, left, right, down)
    luma_range = luma_max - luma_min
    if m.lt(luma_range, max(EDGE_THRESHOLD_MIN, luma_max * EDGE_THRESHOLD_MAX)):
        return center_color
    down_left, down_right, up_left, up_right = lum(-1, -1), lum(1, -1), lum(-1, 1), lum(1, 1)
    down_up, left_right = down + up, left + right
    left_corners, down_corners = down_left + up_left, down_left + down_right
    right_corners, up_corners = down_right + up_right, up_left + up_right
    edge_h = abs(left_corners - 2 * left) + 2 * abs(down_up - 2 * center) + abs(right_corners - 2 * right)
    edge_v = abs(up_corners - 2 * up) + 2 * abs(left_right - 2 * center) + abs(down_corners - 2 * down)
    horizontal = m.ge(edge_h, edge_v)
    l0, l1 = (down, up) if horizontal else (left, right)
    g0, g1 = l0 - center, l1 - center
    gradient_scaled = 0.25 * max(abs(g0), abs(g1))
    step_length = inv[1] if horizontal else inv[0]
    if m.ge(abs(g0), abs(g1)):
        step_length = -step_length
        local_avg = 0.5 * (l0 + center)
    else:
        local_avg = 0.5 * (l1 + center)
    u, v = uv
    if horizontal:
        v += 0.5 * step_length
        off = (inv[0], 0.0)
    else:
        u += 0.5 * step_length
        off = (0.0, inv[1])
    uv1, uv2 = [u - off[0], v - off[1]], [u + off[0], v + off[1]]
    end1 = end2 = False
    e1 = e2 = 0.0
    for i in range(EDGE_SEARCH_ITERATIONS):
        if not end1:
            e1 = luma(sample(img, uv1[0], uv1[1])) - local_avg
            end1 = m.ge(abs(e1), gradient_scaled)
            uv1[0] -= STEP_MOD[i] * off[0]
            uv1[1] -= STEP_MOD[i] * off[1]
        if not end2:
            e2 = luma(sample(img, uv2[0], uv2[1])) - local_avg
            end2 = m.ge(abs(e2), gradient_scaled)
            uv2[0] += STEP_MOD[i] * off[0]
            uv2[1] += STEP_MOD[i] * off[1]
        if end1 and end2:
            break
    d1 = (uv[0] - uv1[0]) if horizontal else (uv[1] - uv1[1])
    d2 = (uv2[0] - uv[0]) if horizontal else (uv2[1] - uv[1])
    # the distances are whole numbers of steps; with equal counts, which end decides is left to the
    # last bit of the float32 uv sums, and it matters only when the two ends disagree in sign
    unit = inv[0] if horizontal else inv[1]
    if abs(d1 - d2) < 0.25 * unit and (e1 < 0.0) != (e2 < 0.0):
        m.value = 0.0
    dist_final = min(d1, d2)
    pixel_offset = -dist_final / (d1 + d2) + 0.5
    center_smaller = m.lt(center, local_avg)
    end_luma = e1 if d1 < d2 else e2
    correct_variation = m.lt(end_luma, 0.0) != center_smaller
    final_offset = pixel_offset if correct_variation else 0.0
    luma_avg = (1.0 / 12.0) * (2 * (down_up + left_right) + left_corners + right_corners)
    s1 = min(max(abs(luma_avg - center) / luma_range, 0.0), 1.0)
    s2 = (-2 * s1 + 3) * s1 * s1
    final_offset = max(final_offset, s2 * s2 * SUBPIXEL_QUALITY)
    fu, fv = uv
    if horizontal:
        fv += final_offset * step_length
    else:
        fu += final_offset * step_length
    return sample(img, fu, fv)


def post_process_pixel(img, x, y, exposure, anti_aliasing):
    """kernels.cu:94-116.  Returns ((r, g, b) before truncation, scaled by 255, decision margin)."""
    h, w = img.shape[:2]
    uv = coords_to_uv(x, y, w, h)
    m = Margin()
    color = fxaa(img, (1.0 / w, 1.0 / h), uv, m) if anti_aliasing else sample(img, uv[0], uv[1])
    out = []
    for c in color:
        mapped = 1.0 - math.exp(-c * exposure)                   # tone_map, transforms.hpp:24
        if mapped < 0.0:
            out.append(math.nan)                                  # pow of a negative base: undefined
            continue
        out.append(255.0 * mapped ** (1.0 / 2.2))                 # gamma_correct :26, float3_to_uchar3 :14
    return out, m.value


def accumulate_pixel(sample_index, denoise, color, albedo, normal, prev_color, prev_albedo, prev_normal):
    """kernels.cu:42-92 for one pixel (every image is read at its own texel: the linear sampler at a
    texel centre returns that texel).  Inputs are 3-tuples; returns (color, albedo, normal)."""
    if not denoise:
        albedo = normal = (0.0, 0.0, 0.0)
    acc_c, acc_a, acc_n = color, albedo, normal
    if sample_index != 0:
        n1 = float(sample_index + 1)
        acc_c = tuple((p * sample_index + c) / n1 for p, c in zip(prev_color, color))
        if denoise:
            acc_a = tuple((p * sample_index + c) / n1 for p, c in zip(prev_albedo, albedo))
            t = tuple((p * sample_index + c) / n1 for p, c in zip(prev_normal, normal))
            length = math.sqrt(sum(q * q for q in t))
            acc_n = tuple(q / length for q in t) if length > 0 else (math.nan,) * 3
    return acc_c, acc_a, acc_n
