// Layout guard for every struct that is passed to a kernel BY VALUE.  hipcc compiles this header twice — once for
// the host, once for gfx950 — and a kernel argument is written by the first pass and read by the second, so a
// member whose size depends on the pass shifts everything behind it.  That is how the one GPU fault of round 2
// happened: an `address_space(3)` pointer in ImageView is 4 bytes in the device pass and 8 in the host pass
// (DESIGN.md §7 "k_shade kept four words ..."; fixed in 99904a0 by keeping the member a generic pointer and casting
// at the use, shade.hpp `texel`).  The numbers below are the same literal in both passes: a pass that disagrees
// with them — and therefore with the other pass — fails to compile.
#pragma once

#include <cstddef>

#include "image_kernels.hpp"
#include "shade.hpp"
#include "traverse.hpp"

namespace nova_hip {

#define NOVA_ARG_LAYOUT(T, SIZE, ALIGN)                                                                   \
  static_assert(sizeof(T) == (SIZE) && alignof(T) == (ALIGN), #T ": kernel-argument layout differs from " \
                                                              "the one both compiler passes must agree on")

NOVA_ARG_LAYOUT(ImageView, 48, 8);
NOVA_ARG_LAYOUT(PathPlanes, 48, 8);
NOVA_ARG_LAYOUT(FrameArgs, 344, 8);
NOVA_ARG_LAYOUT(TraverseArgs, 160, 8);
NOVA_ARG_LAYOUT(AccumulateArgs, 104, 8);
NOVA_ARG_LAYOUT(PostArgs, 80, 8);
NOVA_ARG_LAYOUT(LeafRec, 32, 4);
static_assert(offsetof(ImageView, u8) == 24 && offsetof(ImageView, lut) == 32 && offsetof(ImageView, edge) == 40,
              "ImageView members");
static_assert(offsetof(FrameArgs, materials) == 176 && offsetof(FrameArgs, sky) == 224 && offsetof(FrameArgs, out_color) == 280 &&
              offsetof(FrameArgs, n_mis) == 328 && offsetof(FrameArgs, light_tris) == 336,
              "FrameArgs members");
static_assert(offsetof(TraverseArgs, wide) == 112 && offsetof(TraverseArgs, perm) == 144 && offsetof(TraverseArgs, n_extra_ptr) == 152,
              "TraverseArgs members");

#undef NOVA_ARG_LAYOUT

}
