// Fused SSIM loss (11x11 Gaussian window, sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2):
// the drop-in for fused_ssim.fused_ssim (main_test_dimo.py:979) and for the pure-PyTorch
// src/loss.py:132-175 `ssim` the trainer uses (main_train_dimo.py:343), which costs five grouped
// convolutions plus ~15 elementwise kernels per call.
//
// One workgroup = one 16x16 output tile of one (batch, channel) plane.  The 26x26 halo of both
// images is staged in LDS once; the window is applied separably (11 horizontal taps into LDS, 11
// vertical taps into registers), so each pixel costs 2 x 11 x 5 FMAs instead of 121 x 5.
// Forward emits the SUM of the SSIM map (one atomic per workgroup) and three partial-derivative
// planes; backward convolves those with the same window -> dL/dimg1.  HBM-bound: 2 planes in,
// 3 planes out (forward); 5 planes in, 1 out (backward).
#include <math.h>

#include "common.hpp"

namespace dimo {

constexpr int ST = 16;         // output tile edge
constexpr int SR = 5;          // window radius
constexpr int SH_ = ST + 2 * SR;  // 26
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

struct Window {
  float w[11];
};

// host: same construction as src/loss.py:126-129 (double exp -> fp32 -> normalise in fp32)
static Window make_window() {
  Window win;
  float sum = 0.0f;
  for (int x = 0; x < 11; ++x) {
    win.w[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5));
    sum += win.w[x];
  }
  for (int x = 0; x < 11; ++x) win.w[x] = win.w[x] / sum;
  return win;
}

__device__ __forceinline__ float maybe_clamp(float v, int on) { return on ? fminf(fmaxf(v, 0.0f), 1.0f) : v; }

__global__ void __launch_bounds__(ST *ST) ssim_fwd_kernel(int H, int W, int n_planes, int clamp1, Window win,
                                                          const float *__restrict__ img1,
                                                          const float *__restrict__ img2,
                                                          float *__restrict__ ssim_sum, float *__restrict__ partials,
                                                          size_t plane_stride_total) {
  __shared__ float s_x[SH_][SH_ + 1];
  __shared__ float s_y[SH_][SH_ + 1];
  __shared__ float s_h[5][SH_][ST + 1];
  __shared__ float s_red[ST * ST / 64];

  // A workgroup walks tiles (plane, ty, tx) with stride gridDim.x and keeps its part of the SSIM sum in registers:
  // one atomic per workgroup at the end.  With one tile (and one atomic) per workgroup the 12288 same-address
  // atomics of a 4 x 3 x 512^2 batch were the kernel's critical path (~170 us for ~25 us of arithmetic).
  const int tid = threadIdx.y * ST + threadIdx.x;
  const int tiles_x = (W + ST - 1) / ST, tiles_y = (H + ST - 1) / ST;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  float m_acc = 0.0f;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
  const int plane = tile / (tiles_x * tiles_y);
  const int x0 = (tile % tiles_x) * ST, y0 = ((tile / tiles_x) % tiles_y) * ST;
  const float *p1 = img1 + (size_t)plane * H * W;
  const float *p2 = img2 + (size_t)plane * H * W;
  __syncthreads();  // the previous tile's LDS has been consumed

  for (int t = tid; t < SH_ * SH_; t += ST * ST) {
    const int ly = t / SH_, lx = t % SH_;
    const int gy = y0 + ly - SR, gx = x0 + lx - SR;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s_x[ly][lx] = in ? maybe_clamp(p1[(size_t)gy * W + gx], clamp1) : 0.0f;
    s_y[ly][lx] = in ? p2[(size_t)gy * W + gx] : 0.0f;
  }
  __syncthreads();
  // horizontal pass: 26 rows x 16 columns
  for (int t = tid; t < SH_ * ST; t += ST * ST) {
    const int ly = t / ST, lx = t % ST;
    float a = 0, b = 0, c = 0, d = 0, e = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float x = s_x[ly][lx + k], y = s_y[ly][lx + k], w = win.w[k];
      a += w * x, b += w * y, c += w * x * x, d += w * y * y, e += w * x * y;
    }
    s_h[0][ly][lx] = a, s_h[1][ly][lx] = b, s_h[2][ly][lx] = c, s_h[3][ly][lx] = d, s_h[4][ly][lx] = e;
  }
  __syncthreads();
  // vertical pass
  float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = win.w[k];
    mu1 += w * s_h[0][threadIdx.y + k][threadIdx.x];
    mu2 += w * s_h[1][threadIdx.y + k][threadIdx.x];
    e11 += w * s_h[2][threadIdx.y + k][threadIdx.x];
    e22 += w * s_h[3][threadIdx.y + k][threadIdx.x];
    e12 += w * s_h[4][threadIdx.y + k][threadIdx.x];
  }
  const int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
  const bool in = gx < W && gy < H;
  float m = 0.0f;
  if (in) {
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A1 = 2.0f * mu12 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
    const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
    const float inv = 1.0f / (B1 * B2);
    m = A1 * A2 * inv;
    if (partials) {
      // total derivatives w.r.t. mu1, E[x^2], E[xy] (sigma terms expanded)
      const float dm_dmu1 = (2.0f * mu2 * (A2 - A1) - m * 2.0f * mu1 * (B2 - B1)) * inv;
      const float dm_de11 = -m / B2;
      const float dm_de12 = 2.0f * A1 * inv;
      const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
      partials[o] = dm_dmu1;
      partials[plane_stride_total + o] = dm_de11;
      partials[2 * plane_stride_total + o] = dm_de12;
    }
  }
  m_acc += m;
  }  // tiles
  // block sum -> one atomic
  float v = m_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(ST *ST) ssim_bwd_kernel(int H, int W, int clamp1, Window win,
                                                          const float *__restrict__ img1,
                                                          const float *__restrict__ img2,
                                                          const float *__restrict__ partials,
                                                          size_t plane_stride_total,
                                                          const float *__restrict__ dL_dmean, float inv_numel,
                                                          float *__restrict__ dL_dimg1) {
  __shared__ float s_p[3][SH_][SH_ + 1];
  __shared__ float s_h[3][SH_][ST + 1];
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
  const int tid = threadIdx.y * ST + threadIdx.x;
  for (int t = tid; t < SH_ * SH_; t += ST * ST) {
    const int ly = t / SH_, lx = t % SH_;
    const int gy = y0 + ly - SR, gx = x0 + lx - SR;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
#pragma unroll
    for (int q = 0; q < 3; ++q) s_p[q][ly][lx] = in ? partials[q * plane_stride_total + o] : 0.0f;
  }
  __syncthreads();
  for (int t = tid; t < SH_ * ST; t += ST * ST) {
    const int ly = t / ST, lx = t % ST;
    float a = 0, b = 0, c = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = win.w[k];
      a += w * s_p[0][ly][lx + k], b += w * s_p[1][ly][lx + k], c += w * s_p[2][ly][lx + k];
    }
    s_h[0][ly][lx] = a, s_h[1][ly][lx] = b, s_h[2][ly][lx] = c;
  }
  __syncthreads();
  float a = 0, b = 0, c = 0;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = win.w[k];
    a += w * s_h[0][threadIdx.y + k][threadIdx.x];
    b += w * s_h[1][threadIdx.y + k][threadIdx.x];
    c += w * s_h[2][threadIdx.y + k][threadIdx.x];
  }
  const int gx = x0 + threadIdx.x, gy = y0 + threadIdx.y;
  if (gx < W && gy < H) {
    const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
    const float x = maybe_clamp(img1[o], clamp1), y = img2[o];
    dL_dimg1[o] = (a + 2.0f * x * b + y * c) * (dL_dmean[0] * inv_numel);
  }
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_ssim_forward(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                                 float *ssim_sum, float *partials, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (B < 0 || C < 0 || H <= 0 || W <= 0 || !ssim_sum) return DIMO_E_ARG;
  if (hipMemsetAsync(ssim_sum, 0, sizeof(float), stream) != hipSuccess) return DIMO_E_LAUNCH;
  const long planes = (long)B * C;
  if (planes == 0) return DIMO_OK;
  if (!img1 || !img2 || planes > 65535) return DIMO_E_ARG;
  static const Window win = make_window();
  const long tiles = (long)((W + ST - 1) / ST) * ((H + ST - 1) / ST) * planes;
  const dim3 grid((unsigned)(tiles < 2048 ? tiles : 2048)), block(ST, ST);
  ScopedTimer tm(T_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fwd_kernel, grid, block, 0, stream, H, W, (int)planes, clamp_img1, win, img1, img2, ssim_sum,
                     partials, (size_t)planes * H * W);
  return check_launch();
}

extern "C" int dimo_ssim_backward(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                                  const float *partials, const float *dL_dmean, float *dL_dimg1, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (B < 0 || C < 0 || H <= 0 || W <= 0) return DIMO_E_ARG;
  const long planes = (long)B * C;
  if (planes == 0) return DIMO_OK;
  if (!img1 || !img2 || !partials || !dL_dmean || !dL_dimg1 || planes > 65535) return DIMO_E_ARG;
  static const Window win = make_window();
  const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, (unsigned)planes), block(ST, ST);
  ScopedTimer tm(T_SSIM_BWD, stream);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, stream, H, W, clamp_img1, win, img1, img2, partials,
                     (size_t)planes * H * W, dL_dmean, 1.0f / (float)((double)planes * H * W), dL_dimg1);
  return check_launch();
}
