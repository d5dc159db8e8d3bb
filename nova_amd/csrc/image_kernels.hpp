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
  const uint8_t* mask; // tile partition: per pixel, bit 0 = own tile, bit 1 = halo pixel kept as well; null = every pixel
  bool edge;           // ImageView::edge of the images read
};

// sample_linear<false> at a pixel centre under a tile partition: texels this rank does not keep (mask 0) read
// as 0 without being loaded.  Every tap with a non-zero weight is kept by construction (TilePlan), so for
// finite images this is sample_linear<false> bit for bit — and whatever another rank's pixels hold in this
// rank's memory (never written here; possibly NaN) cannot reach its own pixels through a zero weight.
NV_DEV V3 sample_kept(const float4* p, const uint8_t* mask, int w, int h, float s, float t, bool edge) {
  const float u = s * static_cast<float>(w), v = t * static_cast<float>(h);
  const int i0 = static_cast<int>(__builtin_floorf(u - 0.5f)), j0 = static_cast<int>(__builtin_floorf(v - 0.5f));
  const int i1 = i0 + 1, j1 = j0 + 1;
  const float a = (u - 0.5f) - __builtin_floorf(u - 0.5f);
  const float b = (v - 0.5f) - __builtin_floorf(v - 0.5f);
  auto tx = [&](int i, int j) {
    if (i < 0 || j < 0 || i >= w || j >= h) {
      if (!edge) return make_float4(0.f, 0.f, 0.f, 0.f);
      i = min(max(i, 0), w - 1);
      j = min(max(j, 0), h - 1);
    }
    const size_t k = static_cast<size_t>(j) * w + i;
    return mask[k] ? p[k] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const float4 t00 = tx(i0, j0), t10 = tx(i1, j0), t01 = tx(i0, j1), t11 = tx(i1, j1);
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  return { fma_(t11.x, w11, fma_(t01.x, w01, fma_(t00.x, w00, t10.x * w10))),
           fma_(t11.y, w11, fma_(t01.y, w01, fma_(t00.y, w00, t10.y * w10))),
           fma_(t11.z, w11, fma_(t01.z, w01, fma_(t00.z, w00, t10.z * w10))) };
}

__global__ __launch_bounds__(256) void k_accumulate(AccumulateArgs A) {
  const float si = static_cast<float>(A.sample_index), si1 = static_cast<float>(A.sample_index + 1);
  const uint32_t n_pixels = A.width * A.height;
  const int w = static_cast<int>(A.width), h = static_cast<int>(A.height);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x) {
    if (A.mask && A.mask[i] == 0) continue; // another rank's pixel
    const uint32_t y = i / A.width, x = i - y * A.width;
    const float u = (static_cast<float>(static_cast<int>(x)) + 0.5f) / static_cast<float>(A.width);
    const float v = (static_cast<float>(static_cast<int>(y)) + 0.5f) / static_cast<float>(A.height);
    auto ld = [&](const float4* p) {
      return A.mask ? sample_kept(p, A.mask, w, h, u, v, A.edge)
                    : sample_linear<false>(ImageView { p, w, h, 1, nullptr, nullptr, A.edge }, u, v, 0);
    };
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
  const uint8_t* mask; // as in AccumulateArgs; only own pixels (bit 0) are written
};

__global__ __launch_bounds__(256) void k_post_process(PostArgs A) {
  const uint32_t n = A.width * A.height;
  const float inv_w = 1.0f / static_cast<float>(A.width), inv_h = 1.0f / static_cast<float>(A.height);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (A.mask && (A.mask[i] & 1u) == 0) continue;
    V3 c;
    const uint32_t y = i / A.width, x = i - y * A.width;
    const float u = (static_cast<float>(static_cast<int>(x)) + 0.5f) / static_cast<float>(A.width);
    const float v = (static_cast<float>(static_cast<int>(y)) + 0.5f) / static_cast<float>(A.height);
    if (A.anti_aliasing) {
      c = fxaa(A.color2, inv_w, inv_h, u, v);
    } else if (A.mask) {
      c = sample_kept(A.color2.texels, A.mask, A.color2.w, A.color2.h, u, v, A.color2.edge);
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

// Denoise hand-off (src/core/raytracer.cpp:211-242): flatten<float4, float, 3> and pack<float4, float, 3>
// (src/vector/vector_utils.hpp:8-51) on the device.  Four pixels per thread: 64 bytes in, 48 bytes out as three
// aligned 16-byte stores (and the reverse), the remainder of a plane pixel by pixel.
__global__ __launch_bounds__(256) void k_flatten3(const float4* __restrict__ in, float* __restrict__ out, uint32_t n_pixels) {
  const uint32_t quads = n_pixels / 4u;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
    const float4 a = in[4u * q], b = in[4u * q + 1u], c = in[4u * q + 2u], d = in[4u * q + 3u];
    float4* o = reinterpret_cast<float4*>(out) + 3u * static_cast<size_t>(q);
    o[0] = make_float4(a.x, a.y, a.z, b.x);
    o[1] = make_float4(b.y, b.z, c.x, c.y);
    o[2] = make_float4(c.z, d.x, d.y, d.z);
  }
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_pixels - 4u * quads) {
    const uint32_t i = 4u * quads + t;
    const float4 a = in[i];
    out[3u * static_cast<size_t>(i)] = a.x; out[3u * static_cast<size_t>(i) + 1u] = a.y; out[3u * static_cast<size_t>(i) + 2u] = a.z;
  }
}
// w = 0: pack's vector is value-initialised and only x, y, z are assigned (vector_utils.hpp:35-47)
__global__ __launch_bounds__(256) void k_pack3(const float* __restrict__ in, float4* __restrict__ out, uint32_t n_pixels) {
  const uint32_t quads = n_pixels / 4u;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
    const float4* s = reinterpret_cast<const float4*>(in) + 3u * static_cast<size_t>(q);
    const float4 a = s[0], b = s[1], c = s[2];
    out[4u * q] = make_float4(a.x, a.y, a.z, 0.0f);
    out[4u * q + 1u] = make_float4(a.w, b.x, b.y, 0.0f);
    out[4u * q + 2u] = make_float4(b.z, b.w, c.x, 0.0f);
    out[4u * q + 3u] = make_float4(c.y, c.z, c.w, 0.0f);
  }
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_pixels - 4u * quads) {
    const uint32_t i = 4u * quads + t;
    out[i] = make_float4(in[3u * static_cast<size_t>(i)], in[3u * static_cast<size_t>(i) + 1u], in[3u * static_cast<size_t>(i) + 2u], 0.0f);
  }
}

// Tile stacks for the multi-GPU gather: rank r's tiles (tile id = r + k * ranks, 64x64 texels each, zero
// beyond the image border and beyond the last tile) as one contiguous [k][64][64] array, and back.
template <typename T>
__global__ __launch_bounds__(256) void k_tiles_pack(const T* image, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                    uint32_t n_tiles, uint32_t rank, uint32_t ranks, uint32_t per_rank, T* stack) {
  const size_t total = static_cast<size_t>(per_rank) * 4096u;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const uint32_t k = static_cast<uint32_t>(i >> 12), q = static_cast<uint32_t>(i & 4095u);
    const uint32_t tile = rank + k * ranks;
    T v = {};
    if (tile < n_tiles) {
      const uint32_t x = (tile % tiles_x) * 64u + (q & 63u), y = (tile / tiles_x) * 64u + (q >> 6);
      if (x < width && y < height) v = image[static_cast<size_t>(y) * width + x];
    }
    stack[i] = v;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_tiles_unpack(T* image, uint32_t width, uint32_t height, uint32_t tiles_x,
                                                      uint32_t n_tiles, uint32_t ranks, uint32_t per_rank, const T* stacks) {
  const size_t total = static_cast<size_t>(ranks) * per_rank * 4096u;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const uint32_t q = static_cast<uint32_t>(i & 4095u);
    const size_t rk = i >> 12;
    const uint32_t r = static_cast<uint32_t>(rk / per_rank), k = static_cast<uint32_t>(rk - static_cast<size_t>(r) * per_rank);
    const uint32_t tile = r + k * ranks;
    if (tile >= n_tiles) continue;
    const uint32_t x = (tile % tiles_x) * 64u + (q & 63u), y = (tile / tiles_x) * 64u + (q >> 6);
    if (x < width && y < height) image[static_cast<size_t>(y) * width + x] = stacks[i];
  }
}

// Denominators of the roofline, measured on the box: a float4 copy (HBM read + write) and a
// loop of independent v_fma_f32 chains (FP32 vector rate).
__global__ __launch_bounds__(256) void k_measure_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  // four independent 16-byte loads in flight per lane, a block's 4 KiB pieces next to each other
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4u;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) * 4u + threadIdx.x; i + 3u * blockDim.x < n; i += stride) {
    const float4 a = src[i], b = src[i + blockDim.x], c = src[i + 2u * blockDim.x], d = src[i + 3u * blockDim.x];
    dst[i] = a; dst[i + blockDim.x] = b; dst[i + 2u * blockDim.x] = c; dst[i + 3u * blockDim.x] = d;
  }
}
__global__ __launch_bounds__(256) void k_measure_fma(float* out, int iterations, float seed) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = seed + static_cast<float>(threadIdx.x + k);
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < iterations; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = fma_(a[k], m, c);
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) s += a[k];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s; // keeps the chains alive, never true in practice
}

// dst[i] = pattern (1..16 bytes), for Buffer::fill / fill_image2D
// dst[i] = src[i] for the pixels this rank renders (tile-plan mask != 0); the others keep what they hold
__global__ __launch_bounds__(256) void k_copy_kept(float4* __restrict__ dst, const float4* __restrict__ src,
                                                   const uint8_t* __restrict__ mask, uint32_t n_pixels) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += gridDim.x * blockDim.x)
    if (mask[i]) dst[i] = src[i];
}

template <typename T>
__global__ void k_fill(T* dst, T value, size_t count) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < count;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    dst[i] = value;
  }
}

}
