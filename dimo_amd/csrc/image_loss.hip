// Fused image losses of one motion's batch of renders and their gradients w.r.t. the rasterizer outputs
// (main_train_dimo.py:331-372, src/loss.py:64-106):
//   weighted per-image MSE on clamp(image, 0, 1)     main_train_dimo.py:331-337  (+ the clamp of latent_gs_renderer.py:1279)
//   mask MSE on alpha                                main_train_dimo.py:350
//   edge-aware depth smoothness                      src/loss.py:64-83
//   bilateral normal smoothness                      src/loss.py:86-106
//   + an optional dL/d(clamped image) term coming from the fused SSIM backward.
// In the reference (and in a PyTorch restatement) this is ~60 elementwise / reduction kernels forward and as
// many backward per motion, each streaming [B,C,H,W] tensors; the losses are closed-form, so one pass reads
// the eight rendered planes + targets (with the 4-neighbour stencil served from cache) and writes the four
// gradient images the rasterizer backward consumes.  HBM-bound: 12 planes in, 8 planes out per image.
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
  float c[3], d, n[3];
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
__device__ __forceinline__ float sgn(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }

template <bool DEPTH, bool NORMAL>
__device__ __forceinline__ Px load_px(const float *img, const float *depth, const float *normal, size_t HW,
                                      size_t pix) {
  Px p;
  p.c[0] = clamp01(img[pix]), p.c[1] = clamp01(img[HW + pix]), p.c[2] = clamp01(img[2 * HW + pix]);
  p.d = DEPTH ? depth[pix] : 0.0f;
  if (NORMAL) p.n[0] = normal[pix], p.n[1] = normal[HW + pix], p.n[2] = normal[2 * HW + pix];
  else p.n[0] = p.n[1] = p.n[2] = 0.0f;
  return p;
}

// One neighbouring pair (A = left/top, B = right/bottom).  `side` = +1 if the calling pixel is A, -1 if B.
// Adds the calling pixel's share of the gradients; returns the pair's loss (counted by the A side only).
template <bool DEPTH, bool NORMAL>
__device__ __forceinline__ float pair_term(const Px &A, const Px &B, float side, float w_sm, float w_bl, float *gc,
                                           float &gd, float *gn) {
  const float dc0 = A.c[0] - B.c[0], dc1 = A.c[1] - B.c[1], dc2 = A.c[2] - B.c[2];
  const float gI = (fabsf(dc0) + fabsf(dc1) + fabsf(dc2)) * (1.0f / 3.0f);
  float loss = 0.0f, dL_dgI = 0.0f;
  if (DEPTH) {
    const float e1 = __expf(-gI);
    const float dd = A.d - B.d;
    loss += w_sm * fabsf(dd) * e1;
    gd += side * w_sm * sgn(dd) * e1;
    dL_dgI += -w_sm * fabsf(dd) * e1;
  }
  if (NORMAL) {
    const float e3 = __expf(-3.0f * gI);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float dn = A.n[k] - B.n[k];
      const float a = fabsf(dn) * e3;
      const float root = sqrtf(1.0f + a * a);
      const float q = a / root;
      loss += w_bl * root;
      gn[k] += side * w_bl * q * e3 * sgn(dn);
      dL_dgI += -3.0f * w_bl * q * a;
    }
  }
  const float s = side * dL_dgI * (1.0f / 3.0f);
  gc[0] += s * sgn(dc0), gc[1] += s * sgn(dc1), gc[2] += s * sgn(dc2);
  return loss;
}

template <bool DEPTH, bool NORMAL>
__global__ void __launch_bounds__(256) image_loss_kernel(
    int H, int W, int n_images, LossParams prm, const float *__restrict__ image, const float *__restrict__ depth,
    const float *__restrict__ normal, const float *__restrict__ alpha, const float *__restrict__ gt,
    const float *__restrict__ mask, size_t mask_stride, const float *__restrict__ ssim_grad,
    float *__restrict__ loss_out, float *__restrict__ g_image, float *__restrict__ g_depth,
    float *__restrict__ g_normal, float *__restrict__ g_alpha, float *__restrict__ g_dot) {
  __shared__ float s_red[4];
  const size_t HW = (size_t)H * W;
  float loss = 0.0f;
  // a workgroup walks 32x8 pixel tiles (image, ty, tx) with stride gridDim.x: ONE atomic per workgroup, spread over
  // 16 words in 16 cache lines (same-address atomics drain at ~7 ns each: 4096 of them were most of this kernel's
  // 67 us, 1024 still 4 of 40; with the spread the grid size no longer matters -- 35 us for 8 images = 5 TB/s)
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  for (int tile = blockIdx.x; tile < tiles_x * tiles_y * n_images; tile += gridDim.x) {
  const int b = tile / (tiles_x * tiles_y);
  const int x = (tile % tiles_x) * 32 + (threadIdx.x & 31), y = ((tile / tiles_x) % tiles_y) * 8 + (threadIdx.x >> 5);
  const float *img = image + (size_t)b * 3 * HW, *dep = DEPTH ? depth + (size_t)b * HW : nullptr;
  const float *nrm = NORMAL ? normal + (size_t)b * 3 * HW : nullptr;
  if (x < W && y < H) {
    const size_t pix = (size_t)y * W + x;
    const Px P = load_px<DEPTH, NORMAL>(img, dep, nrm, HW, pix);
    float gc[3] = {0, 0, 0}, gd = 0.0f, gn[3] = {0, 0, 0};
    // per-pixel terms
    const float wm = prm.w_mse[b];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float *gtb = prm.gt_image[b] ? prm.gt_image[b] : gt + (size_t)b * 3 * HW;
      const float e = P.c[k] - gtb[k * HW + pix];
      loss += wm * e * e;
      gc[k] += 2.0f * wm * e;
    }
    const float a = alpha[(size_t)b * HW + pix];
    const float em = a - (prm.mask_image[b] ? prm.mask_image[b][pix] : mask[(size_t)b * mask_stride + pix]);
    loss += prm.w_mask * em * em;
    g_alpha[(size_t)b * HW + pix] = 2.0f * prm.w_mask * em;
    float dot = 2.0f * prm.w_mask * em * a;  // sum over the channels of gradient x rendered value (see g_dot)
    // stencil terms.  The four neighbours are loaded UNCONDITIONALLY (clamped to the pixel itself at the image
    // border, where the pair's weights are zero): a branch around each neighbour's loads made the compiler wait for
    // them one after the other -- four dependent memory round trips per pixel.
    if (DEPTH || NORMAL) {
      const bool vr = x + 1 < W, vd = y + 1 < H, vl = x > 0, vu = y > 0;
      const Px QR = load_px<DEPTH, NORMAL>(img, dep, nrm, HW, vr ? pix + 1 : pix);
      const Px QD = load_px<DEPTH, NORMAL>(img, dep, nrm, HW, vd ? pix + W : pix);
      const Px QL = load_px<DEPTH, NORMAL>(img, dep, nrm, HW, vl ? pix - 1 : pix);
      const Px QU = load_px<DEPTH, NORMAL>(img, dep, nrm, HW, vu ? pix - W : pix);
      loss += pair_term<DEPTH, NORMAL>(P, QR, 1.0f, vr ? prm.w_smooth_x : 0.0f, vr ? prm.w_bilat_x : 0.0f, gc, gd, gn);
      loss += pair_term<DEPTH, NORMAL>(P, QD, 1.0f, vd ? prm.w_smooth_y : 0.0f, vd ? prm.w_bilat_y : 0.0f, gc, gd, gn);
      (void)pair_term<DEPTH, NORMAL>(QL, P, -1.0f, vl ? prm.w_smooth_x : 0.0f, vl ? prm.w_bilat_x : 0.0f, gc, gd, gn);
      (void)pair_term<DEPTH, NORMAL>(QU, P, -1.0f, vu ? prm.w_smooth_y : 0.0f, vu ? prm.w_bilat_y : 0.0f, gc, gd, gn);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float g = gc[k];
      if (ssim_grad) g += ssim_grad[(size_t)b * 3 * HW + k * HW + pix];
      const float raw = img[k * HW + pix];
      g = (raw >= 0.0f && raw <= 1.0f) ? g : 0.0f;  // clamp backward
      g_image[(size_t)b * 3 * HW + k * HW + pix] = g;
      dot += g * raw;
    }
    if (g_depth) g_depth[(size_t)b * HW + pix] = gd, dot += gd * P.d;
    if (NORMAL) {
#pragma unroll
      for (int k = 0; k < 3; ++k) g_normal[(size_t)b * 3 * HW + k * HW + pix] = gn[k], dot += gn[k] * P.n[k];
    }
    if (g_dot) g_dot[(size_t)b * HW + pix] = dot;
  }
  }  // tiles
  float v = loss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  // 16 accumulator words in 16 cache lines (DIMO_LOSS_WORDS = 16 x 32 floats): the caller sums them
  if (threadIdx.x == 0) atomicAdd(loss_out + (blockIdx.x & 15) * 32, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_image_loss(int B, int H, int W, const float *image, const float *depth, const float *normal,
                               const float *alpha, const float *gt, const float *mask, int mask_per_image,
                               const float *w_mse_host, float w_mask, float w_smooth_x, float w_smooth_y,
                               float w_bilat_x, float w_bilat_y, const float *ssim_grad, float *loss_accum,
                               float *g_image, float *g_depth, float *g_normal, float *g_alpha, float *g_dot,
                               const float *const *gt_images_host, const float *const *mask_images_host,
                               void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (B < 0 || B > LOSS_MAX_B || H <= 0 || W <= 0) return DIMO_E_ARG;
  if (B == 0) return DIMO_OK;
  if (!image || !alpha || (!gt && !gt_images_host) || (!mask && !mask_images_host) || !w_mse_host || !loss_accum ||
      !g_image || !g_alpha)
    return DIMO_E_ARG;
  if ((depth == nullptr) != (g_depth == nullptr) || (normal == nullptr) != (g_normal == nullptr)) return DIMO_E_ARG;
  LossParams prm;
  for (int b = 0; b < LOSS_MAX_B; ++b) {
    prm.w_mse[b] = b < B ? w_mse_host[b] : 0.0f;
    prm.gt_image[b] = (gt_images_host && b < B) ? gt_images_host[b] : nullptr;
    prm.mask_image[b] = (mask_images_host && b < B) ? mask_images_host[b] : nullptr;
    if (b < B && ((gt_images_host && !prm.gt_image[b]) || (mask_images_host && !prm.mask_image[b]))) return DIMO_E_ARG;
  }
  prm.w_mask = w_mask, prm.w_smooth_x = w_smooth_x, prm.w_smooth_y = w_smooth_y;
  prm.w_bilat_x = w_bilat_x, prm.w_bilat_y = w_bilat_y;
  const long tiles = (long)((W + 31) / 32) * ((H + 7) / 8) * B;
  const dim3 grid((unsigned)(tiles < 1024 ? tiles : 1024)), block(256);
  const size_t mstride = mask_per_image ? (size_t)H * W : 0;
  ScopedTimer tm(T_LOSS, stream);
#define DIMO_LAUNCH_LOSS(D, N)                                                                                  \
  hipLaunchKernelGGL((image_loss_kernel<D, N>), grid, block, 0, stream, H, W, B, prm, image, depth, normal, alpha, \
                     gt, mask, mstride, ssim_grad, loss_accum, g_image, g_depth, g_normal, g_alpha, g_dot)
  if (depth && normal) DIMO_LAUNCH_LOSS(true, true);
  else if (depth) DIMO_LAUNCH_LOSS(true, false);
  else if (normal) DIMO_LAUNCH_LOSS(false, true);
  else DIMO_LAUNCH_LOSS(false, false);
#undef DIMO_LAUNCH_LOSS
  return check_launch();
}
