// Per-image loss terms shared by the streaming loss kernel (image_loss.hip) and the one-pass SSIM + loss tile kernel
// (ssim.hip): main_train_dimo.py:331-372, src/loss.py:64-106.
#pragma once
#include "common.hpp"

namespace dimo {

constexpr int LOSS_MAX_B = 64;
struct LossParams {
  float w_mse[LOSS_MAX_B];  // lambda_mse * (1 or 0.5) / (3 H W) per image
  float w_mask;             // lambda_mask * share / (B H W)
  float w_smooth_x, w_smooth_y;   // lambda_smooth * share / (B H (W-1)) , / (B (H-1) W)
  float w_bilat_x, w_bilat_y;     // lambda_bilateral * share / (3 B H (W-1)) , / (3 B (H-1) W)
  // optional per-image base pointers of the targets / masks (null: one contiguous gt / mask tensor): the batch's
  // targets live in a resident pool, one tensor per image -- no stacking copies at the head of the step
  const float *gt_image[LOSS_MAX_B];
  const float *mask_image[LOSS_MAX_B];
};

struct Px {
  float c[3], d, n[3];  // clamped colour, depth, normal
};
struct Grad {
  float c[3], d, n[3];
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float sgn(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }
// lane l <- lane l - 1 / lane l + 1 of the wave (DPP wave shifts: no LDS, one VALU move each)
__device__ __forceinline__ float from_lane_below(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// One neighbouring pair (A = left / top, B = right / bottom): returns the pair's loss and the gradient w.r.t. A's
// values in t; the gradient w.r.t. B's values is -t (every term is a function of A - B).
//   edge-aware depth smoothness   w_sm |dA - dB| exp(-gI)                         src/loss.py:64-83
//   bilateral normal smoothness   w_bl sqrt(1 + (|nA - nB| exp(-3 gI))^2)         src/loss.py:86-106
// with gI = mean_c |cA - cB|.  v_rsq_f32 (1 ulp) replaces sqrt + division.
template <bool DEPTH, bool NORMAL>
__device__ __forceinline__ float pair_term(const Px &A, const Px &B, float w_sm, float w_bl, Grad &t) {
  const float dc0 = A.c[0] - B.c[0], dc1 = A.c[1] - B.c[1], dc2 = A.c[2] - B.c[2];
  const float gI = (fabsf(dc0) + fabsf(dc1) + fabsf(dc2)) * (1.0f / 3.0f);
  float loss = 0.0f, dL_dgI = 0.0f;
  const float e1 = __expf(-gI);
  t.d = 0.0f, t.n[0] = t.n[1] = t.n[2] = 0.0f;
  if (DEPTH) {
    const float dd = A.d - B.d;
    const float l = w_sm * fabsf(dd) * e1;
    loss += l;
    t.d = w_sm * sgn(dd) * e1;
    dL_dgI -= l;
  }
  if (NORMAL) {
    const float e3 = e1 * e1 * e1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float an = (A.n[k] - B.n[k]) * e3;  // signed |dn| e3
      const float s = 1.0f + an * an;
      const float r = __builtin_amdgcn_rsqf(s);
      loss += w_bl * (s * r);                    // sqrt(s)
      const float q = an * r;                    // signed a / root
      t.n[k] = w_bl * q * e3;
      dL_dgI -= 3.0f * w_bl * q * an;
    }
  }
  const float g = dL_dgI * (1.0f / 3.0f);
  t.c[0] = g * sgn(dc0), t.c[1] = g * sgn(dc1), t.c[2] = g * sgn(dc2);
  return loss;
}
__device__ __forceinline__ void add(Grad &g, const Grad &t, float s) {
#pragma unroll
  for (int k = 0; k < 3; ++k) g.c[k] += s * t.c[k], g.n[k] += s * t.n[k];
  g.d += s * t.d;
}

// scalar base + 32-bit byte offset per lane: the addressing mode of global_load / global_store with an SGPR base
// (a 64-bit address per lane and plane cost two VALU instructions per access)
__device__ __forceinline__ float ld(const float *base, unsigned off) {
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off);
}
__device__ __forceinline__ void st(float *base, unsigned off, float v) {
  *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) = v;
}

}  // namespace dimo
