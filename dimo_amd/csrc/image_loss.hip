// Fused image losses of one motion's batch of renders and their gradients w.r.t. the rasterizer outputs
// (main_train_dimo.py:331-372, src/loss.py:64-106):
//   weighted per-image MSE on clamp(image, 0, 1)     main_train_dimo.py:331-337  (+ the clamp of latent_gs_renderer.py:1279)
//   mask MSE on alpha                                main_train_dimo.py:350
//   edge-aware depth smoothness                      src/loss.py:64-83
//   bilateral normal smoothness                      src/loss.py:86-106
//   + an optional dL/d(clamped image) term coming from the fused SSIM backward.
// In the reference (and in a PyTorch restatement) this is ~60 elementwise / reduction kernels forward and as
// many backward per motion, each streaming [B,C,H,W] tensors; the losses are closed-form, so one pass reads
// the eight rendered planes + targets once and writes the four gradient images the rasterizer backward consumes
// (+ the per-pixel gradient x value plane).  HBM-bound: 15 planes in, 9 planes out per image.
#include "loss_terms.hpp"

namespace dimo {

// A WAVE owns LOSS_COLS columns x LOSS_ROWS rows of one image and walks down its rows: lane l holds pixel
// (x0 - 1 + l, y) -- lanes 0 and 63 are halo columns, the first and last row visited are halo rows.  Every
// neighbouring pair is evaluated ONCE per wave (the first version evaluated the four pairs of every pixel: each pair
// twice, 790 VALU instructions per pixel and VALU-bound at 3 TB/s): the horizontal pair (x - 1, x) by lane x, its
// other half handed to lane x - 1 with a DPP wave shift; the vertical pair (y - 1, y) when row y arrives, its other
// half completing row y - 1, which is then written.  A pixel's planes are loaded once (the row below is the next
// iteration's own row), not once per neighbour.
constexpr int LOSS_COLS = 62, LOSS_ROWS = 8;  // (4 rows: 30.4 us per 4 images, 8: 27.9, 16: 31.2)

template <bool DEPTH, bool NORMAL>
struct RowLoad {  // one pixel's inputs as loaded (the next row's are in flight while this row is processed)
  float raw[3], d, n[3], gt[3], a, m, sg[3];
};

template <bool DEPTH, bool NORMAL>
__global__ void __launch_bounds__(256) image_loss_kernel(
    int H, int W, int n_images, LossParams prm, const float *__restrict__ image, const float *__restrict__ depth,
    const float *__restrict__ normal, const float *__restrict__ alpha, const float *__restrict__ gt,
    const float *__restrict__ mask, size_t mask_stride, const float *__restrict__ ssim_grad,
    float *__restrict__ loss_out, float *__restrict__ g_image, float *__restrict__ g_depth,
    float *__restrict__ g_normal, float *__restrict__ g_alpha, float *__restrict__ g_dot) {
  __shared__ float s_red[4];
  const int HW = H * W;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the image pointers stay in SGPRs
  const int strips = (W + LOSS_COLS - 1) / LOSS_COLS, blocks = (H + LOSS_ROWS - 1) / LOSS_ROWS;
  const int units = strips * blocks * n_images;
  float loss = 0.0f;
  for (int unit = blockIdx.x * 4 + wave; unit < units; unit += gridDim.x * 4) {
    const int b = unit / (strips * blocks), rem = unit - b * (strips * blocks);
    const int blk = rem / strips, strip = rem - blk * strips;
    const int x = strip * LOSS_COLS - 1 + lane, y0 = blk * LOSS_ROWS;
    const int y_end = min(y0 + LOSS_ROWS, H);                       // useful rows: y0 .. y_end - 1
    const bool own_x = lane >= 1 && lane <= LOSS_COLS && x < W;     // this lane's pixels are written by this wave
    const int xc = min(max(x, 0), W - 1);
    const bool pair_x = x >= 1 && x < W;                            // the pair (x - 1, x) exists
    const float w_sx = pair_x ? prm.w_smooth_x : 0.0f, w_bx = pair_x ? prm.w_bilat_x : 0.0f;
    const float *img = image + (size_t)b * 3 * HW, *dep = DEPTH ? depth + (size_t)b * HW : nullptr;
    const float *nrm = NORMAL ? normal + (size_t)b * 3 * HW : nullptr;
    const float *gtb = prm.gt_image[b] ? prm.gt_image[b] : gt + (size_t)b * 3 * HW;
    const float *mkb = prm.mask_image[b] ? prm.mask_image[b] : mask + (size_t)b * mask_stride;
    const float *alp = alpha + (size_t)b * HW;
    const float *sgb = ssim_grad ? ssim_grad + (size_t)b * 3 * HW : nullptr;
    const float wm = prm.w_mse[b];

    auto load_row = [&](int r, RowLoad<DEPTH, NORMAL> &L) {
      const unsigned off = 4u * (unsigned)(min(max(r, 0), H - 1) * W + xc);
      L.raw[0] = ld(img, off), L.raw[1] = ld(img + HW, off), L.raw[2] = ld(img + 2 * HW, off);
      if (DEPTH) L.d = ld(dep, off);
      if (NORMAL) L.n[0] = ld(nrm, off), L.n[1] = ld(nrm + HW, off), L.n[2] = ld(nrm + 2 * HW, off);
      if (r >= y0 && r < y_end) {  // wave-uniform: the per-pixel terms only exist for the rows this wave writes
        L.gt[0] = ld(gtb, off), L.gt[1] = ld(gtb + HW, off), L.gt[2] = ld(gtb + 2 * HW, off);
        L.a = ld(alp, off), L.m = ld(mkb, off);
        if (sgb) L.sg[0] = ld(sgb, off), L.sg[1] = ld(sgb + HW, off), L.sg[2] = ld(sgb + 2 * HW, off);
      }
    };

    Px Pp;            // previous row: pixel, accumulated gradient, what its write-out still needs
    Grad gp;
    float raw_p[3], ga_p = 0.0f, a_p = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) Pp.c[k] = Pp.n[k] = gp.c[k] = gp.n[k] = raw_p[k] = 0.0f;
    Pp.d = gp.d = 0.0f;

    auto process = [&](int r, const RowLoad<DEPTH, NORMAL> &cur) {
      const bool own_row = r >= y0 && r < y_end;
      Px P;
      P.c[0] = clamp01(cur.raw[0]), P.c[1] = clamp01(cur.raw[1]), P.c[2] = clamp01(cur.raw[2]);
      P.d = DEPTH ? cur.d : 0.0f;
      P.n[0] = NORMAL ? cur.n[0] : 0.0f, P.n[1] = NORMAL ? cur.n[1] : 0.0f, P.n[2] = NORMAL ? cur.n[2] : 0.0f;
      Grad g;
      g.c[0] = g.c[1] = g.c[2] = g.d = g.n[0] = g.n[1] = g.n[2] = 0.0f;
      float ga = 0.0f;
      if (own_row) {
        float l = 0.0f;
        // per-pixel terms: weighted MSE on the clamped image, mask MSE on alpha, the SSIM gradient
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float e = P.c[k] - cur.gt[k];
          l += wm * e * e;
          g.c[k] = 2.0f * wm * e + (sgb ? cur.sg[k] : 0.0f);
        }
        const float em = cur.a - cur.m;
        l += prm.w_mask * em * em;
        ga = 2.0f * prm.w_mask * em;
        if (DEPTH || NORMAL) {
          // horizontal pair (x - 1, x): this lane is B, lane - 1 is A
          Px A;
#pragma unroll
          for (int k = 0; k < 3; ++k) A.c[k] = from_lane_below(P.c[k]), A.n[k] = NORMAL ? from_lane_below(P.n[k]) : 0.0f;
          A.d = DEPTH ? from_lane_below(P.d) : 0.0f;
          Grad t;
          l += pair_term<DEPTH, NORMAL>(A, P, w_sx, w_bx, t);  // (counted where B is an own pixel: each pair once)
          add(g, t, -1.0f);
          Grad ta;  // lane + 1's pair has this lane as A
#pragma unroll
          for (int k = 0; k < 3; ++k) ta.c[k] = from_lane_above(t.c[k]), ta.n[k] = NORMAL ? from_lane_above(t.n[k]) : 0.0f;
          ta.d = DEPTH ? from_lane_above(t.d) : 0.0f;
          add(g, ta, 1.0f);
        }
        if (own_x) loss += l;
      }
      if ((DEPTH || NORMAL) && r >= y0) {
        // vertical pair (r - 1, r): completes row r - 1
        const bool pair_y = r >= 1 && r < H && x >= 0 && x < W;
        Grad t;
        const float l = pair_term<DEPTH, NORMAL>(Pp, P, pair_y ? prm.w_smooth_y : 0.0f, pair_y ? prm.w_bilat_y : 0.0f, t);
        if (own_row && own_x) loss += l;  // counted with its lower row
        add(gp, t, 1.0f);
        add(g, t, -1.0f);
      }
      if (r > y0 && own_x) {  // row r - 1 is an own row and complete
        const unsigned off = 4u * (unsigned)((r - 1) * W + x);
        float dot = ga_p * a_p;  // sum over the channels of gradient x rendered value (see g_dot)
        st(g_alpha + (size_t)b * HW, off, ga_p);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float gk = (raw_p[k] >= 0.0f && raw_p[k] <= 1.0f) ? gp.c[k] : 0.0f;  // clamp backward
          st(g_image + (size_t)b * 3 * HW + k * HW, off, gk);
          dot += gk * raw_p[k];
        }
        if (g_depth) st(g_depth + (size_t)b * HW, off, gp.d), dot += gp.d * Pp.d;
        if (NORMAL) {
#pragma unroll
          for (int k = 0; k < 3; ++k) st(g_normal + (size_t)b * 3 * HW + k * HW, off, gp.n[k]), dot += gp.n[k] * Pp.n[k];
        }
        if (g_dot) st(g_dot + (size_t)b * HW, off, dot);
      }
      Pp = P, gp = g, ga_p = ga, a_p = cur.a;
#pragma unroll
      for (int k = 0; k < 3; ++k) raw_p[k] = cur.raw[k];
    };

    // rows y0 - 1 .. y_end through a ring of three buffers: two rows are in flight while one is processed (a wave
    // holds ~4 KB per row; with one row ahead the chip had < 9 MB in flight and the kernel ran in bursts at 3.3 TB/s).
    // The ring is unrolled by hand: moving a buffer whose loads are in flight would wait for them.
    RowLoad<DEPTH, NORMAL> L0{}, L1{}, L2{};
    load_row(y0 - 1, L0);
    load_row(y0, L1);
    for (int r = y0 - 1; r <= y_end; r += 3) {
      if (r + 2 <= y_end) load_row(r + 2, L2);
      process(r, L0);
      if (r + 1 > y_end) break;
      if (r + 3 <= y_end) load_row(r + 3, L0);
      process(r + 1, L1);
      if (r + 2 > y_end) break;
      if (r + 4 <= y_end) load_row(r + 4, L1);
      process(r + 2, L2);
    }
  }  // units
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
  if ((long)H * W > (1L << 28)) return DIMO_E_ARG;  // byte offsets inside an image are 32-bit
  const long units = (long)((W + LOSS_COLS - 1) / LOSS_COLS) * ((H + LOSS_ROWS - 1) / LOSS_ROWS) * B;  // one per wave
  const long wgs = (units + 3) / 4;
  const dim3 grid((unsigned)(wgs < 2048 ? wgs : 2048)), block(256);
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
