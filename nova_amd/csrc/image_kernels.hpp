// Streaming image kernels for gfx950: kernel_accumulate and kernel_post_process (+ FXAA)
// (reference kernels/kernels.cu:42-116, kernels/anti_aliasing.hpp:14-150,
// kernels/transforms.hpp:14-30).  Images are linear float4 / uchar4 buffers; one 16-byte access
// per lane, consecutive lanes on consecutive texels (1 KiB per wave instruction).
//
// The reference reads every input through a LINEAR, CLAMP sampler at the pixel centre
// (kernels.cu:59-64,103-110).  That is the texel itself except where (x+0.5)/W*W rounds off x+0.5
// and where a neighbouring texel is NaN/inf (NaN * 0 = NaN), so the same software sampler is used
// here: the three extra taps hit the lines the neighbouring lanes fetch anyway.
#pragma once

#include "shade.hpp"

namespace nova_hip {

struct AccumulateArgs {
  int32_t sample_index;
  uint32_t denoise;
  const float4 *color1, *albedo1, *normal1, *prev_color, *prev_albedo, *prev_normal;
  float4 *color2, *albedo2, *normal2;
  uint32_t width, height;
};

__global__ __launch_bounds__(256) void k_accumulate(AccumulateArgs A) {
  const float si = static_cast<float>(A.sample_index), si1 = static_cast<float>(A.sample_index + 1);
  const uint32_t n_pixels = A.width * A.height;
  const int w = static_cast<int>(A.width), h = static_cast<int>(A.height);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x) {
    const uint32_t y = i / A.width, x = i - y * A.width;
    const float u = (static_cast<float>(static_cast<int>(x)) + 0.5f) / static_cast<float>(A.width);
    const float v = (static_cast<float>(static_cast<int>(y)) + 0.5f) / static_cast<float>(A.height);
    auto ld = [&](const float4* p) { return sample_linear<false>(ImageView { p, w, h, 1 }, u, v, 0); };
    V3 c = ld(A.color1);
    V3 a { 0.f, 0.f, 0.f }, n { 0.f, 0.f, 0.f };
    if (A.denoise) {
      a = ld(A.albedo1);
      n = ld(A.normal1);
    }
    if (A.sample_index != 0) {
      c = fma3(ld(A.prev_color), si, c) / si1;
      if (A.denoise) {
        a = fma3(ld(A.prev_albedo), si, a) / si1;
        n = normalize(fma3(ld(A.prev_normal), si, n) / si1);
      }
    }
    A.color2[i] = make_float4(c.x, c.y, c.z, 1.0f);
    if (A.denoise) {
      A.albedo2[i] = make_float4(a.x, a.y, a.z, 1.0f);
      A.normal2[i] = make_float4(n.x, n.y, n.z, 1.0f);
    }
  }
}

NV_DEV float luma(V3 c) { return dot(c, { 0.299f, 0.587f, 0.114f }); }

// kernels/anti_aliasing.hpp:14-150
NV_DEV V3 fxaa(const ImageView& im, float inv_w, float inv_h, float u, float v) {
  auto tap = [&](float s, float t) { return sample_linear<false>(im, s, t, 0); };
  const V3 center_color = tap(u, v);
  const float center = luma(center_color);
  const float down = luma(tap(u + 0.0f * inv_w, v + -1.0f * inv_h));
  const float left = luma(tap(u + -1.0f * inv_w, v + 0.0f * inv_h));
  const float right = luma(tap(u + 1.0f * inv_w, v + 0.0f * inv_h));
  const float up = luma(tap(u + 0.0f * inv_w, v + 1.0f * inv_h));
  const float lmax = max_(center, max_(up, max_(left, max_(right, down))));
  const float lmin = min_(center, min_(up, min_(left, min_(right, down))));
  const float range = lmax - lmin;
  if (range < max_(0.0312f, lmax * 0.125f)) return center_color;
  const float dl = luma(tap(u + -1.0f * inv_w, v + -1.0f * inv_h));
  const float dr = luma(tap(u + 1.0f * inv_w, v + -1.0f * inv_h));
  const float ul = luma(tap(u + -1.0f * inv_w, v + 1.0f * inv_h));
  const float ur = luma(tap(u + 1.0f * inv_w, v + 1.0f * inv_h));
  const float down_up = down + up, left_right = left + right;
  const float left_c = dl + ul, down_c = dl + dr, right_c = dr + ur, up_c = ul + ur;
  const float edge_h = fma_(2.0f, __builtin_fabsf(fma_(-2.0f, center, down_up)), __builtin_fabsf(fma_(-2.0f, left, left_c))) +
                       __builtin_fabsf(fma_(-2.0f, right, right_c));
  const float edge_v = fma_(2.0f, __builtin_fabsf(fma_(-2.0f, center, left_right)), __builtin_fabsf(fma_(-2.0f, up, up_c))) +
                       __builtin_fabsf(fma_(-2.0f, down, down_c));
  const bool horiz = edge_h >= edge_v;
  const float l1 = horiz ? down : left, l2 = horiz ? up : right;
  const float g1 = l1 - center, g2 = l2 - center;
  const float grad_scaled = 0.25f * max_(__builtin_fabsf(g1), __builtin_fabsf(g2));
  float step = horiz ? inv_h : inv_w;
  float local_avg;
  if (__builtin_fabsf(g1) >= __builtin_fabsf(g2)) {
    step = -step;
    local_avg = 0.5f * (l1 + center);
  } else {
    local_avg = 0.5f * (l2 + center);
  }
  float cu = u, cv = v;
  if (horiz) cv = fma_(0.5f, step, cv); else cu = fma_(0.5f, step, cu);
  const float ox = horiz ? inv_w : 0.0f, oy = horiz ? 0.0f : inv_h;
  float u1 = cu - ox, v1 = cv - oy, u2 = cu + ox, v2 = cv + oy;
  bool end1 = false, end2 = false;
  float le1 = 0.f, le2 = 0.f;
  // step multipliers of the 12 search iterations (anti_aliasing.hpp:99-100)
  for (int i = 0; i < 12; i++) {
    const float mod = i < 5 ? 1.0f : (i == 5 ? 1.5f : (i < 10 ? 2.0f : (i == 10 ? 4.0f : 8.0f)));
    if (!end1) {
      le1 = luma(tap(u1, v1)) - local_avg;
      end1 = __builtin_fabsf(le1) >= grad_scaled;
      u1 = fma_(-mod, ox, u1);
      v1 = fma_(-mod, oy, v1);
    }
    if (!end2) {
      le2 = luma(tap(u2, v2)) - local_avg;
      end2 = __builtin_fabsf(le2) >= grad_scaled;
      u2 = fma_(mod, ox, u2);
      v2 = fma_(mod, oy, v2);
    }
    if (end1 && end2) break;
  }
  const float d1 = horiz ? (u - u1) : (v - v1);
  const float d2 = horiz ? (u2 - u) : (v2 - v);
  const float dmin = min_(d1, d2);
  const float thick = d1 + d2;
  const float pix_off = -dmin / thick + 0.5f;
  const bool center_smaller = center < local_avg;
  const bool correct = ((d1 < d2 ? le1 : le2) < 0.0f) != center_smaller;
  float final_off = correct ? pix_off : 0.0f;
  const float lavg = (1.0f / 12.0f) * (fma_(2.0f, down_up + left_right, left_c) + right_c);
  const float s1 = min_(max_(__builtin_fabsf(lavg - center) / range, 0.0f), 1.0f);
  const float s2 = (fma_(-2.0f, s1, 3.0f) * s1) * s1;
  const float sfin = (s2 * s2) * 0.75f;
  final_off = max_(final_off, sfin);
  float fu = u, fv = v;
  if (horiz) fv = fma_(final_off, step, fv); else fu = fma_(final_off, step, fu);
  return tap(fu, fv);
}

struct PostArgs {
  ImageView color2;
  uchar4* pixels;
  uint32_t width, height;
  float exposure;
  uint32_t anti_aliasing;
};

__global__ __launch_bounds__(256) void k_post_process(PostArgs A) {
  const uint32_t n = A.width * A.height;
  const float inv_w = 1.0f / static_cast<float>(A.width), inv_h = 1.0f / static_cast<float>(A.height);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    V3 c;
    const uint32_t y = i / A.width, x = i - y * A.width;
    const float u = (static_cast<float>(static_cast<int>(x)) + 0.5f) / static_cast<float>(A.width);
    const float v = (static_cast<float>(static_cast<int>(y)) + 0.5f) / static_cast<float>(A.height);
    if (A.anti_aliasing) {
      c = fxaa(A.color2, inv_w, inv_h, u, v);
    } else {
      c = sample_linear<false>(A.color2, u, v, 0);
    }
    // tone_map, gamma_correct, float3_to_uchar3 (kernels/transforms.hpp:14-26)
    const float g = 1.0f / 2.2f;
    const V3 m = { 1.0f - nova_expf(-c.x * A.exposure), 1.0f - nova_expf(-c.y * A.exposure), 1.0f - nova_expf(-c.z * A.exposure) };
    const V3 o = { nova_powf(m.x, g) * 255.0f, nova_powf(m.y, g) * 255.0f, nova_powf(m.z, g) * 255.0f };
    uchar4 px;
    px.x = static_cast<unsigned char>(static_cast<int>(o.x));
    px.y = static_cast<unsigned char>(static_cast<int>(o.y));
    px.z = static_cast<unsigned char>(static_cast<int>(o.z));
    px.w = 255;
    A.pixels[i] = px;
  }
}

// dst[i] = pattern (1..16 bytes), for Buffer::fill / fill_image2D
template <typename T>
__global__ void k_fill(T* dst, T value, size_t count) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < count;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    dst[i] = value;
  }
}

}
