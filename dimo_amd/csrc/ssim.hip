// Fused SSIM loss (11x11 Gaussian window, sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2):
// the drop-in for fused_ssim.fused_ssim (main_test_dimo.py:979) and for the pure-PyTorch
// src/loss.py:132-175 `ssim` the trainer uses (main_train_dimo.py:343), which costs five grouped
// convolutions plus ~15 elementwise kernels per call.
//
// A workgroup owns 32x32 output tiles of (batch, channel) planes.  The 42x42 halo of both images is staged in
// LDS once; the window is applied separably with register tiling (11 horizontal taps into LDS, 11 vertical taps
// into registers), so each pixel costs 2 x 11 x 5 FMAs instead of 121 x 5.
// Forward emits the SUM of the SSIM map (one atomic per workgroup) and three partial-derivative
// planes; backward convolves those with the same window -> dL/dimg1.  HBM-bound: 2 planes in,
// 3 planes out (forward); 5 planes in, 1 out (backward).
#include <math.h>

#include "common.hpp"

namespace dimo {

constexpr int TS = 32;            // output tile edge
constexpr int SR = 5;             // window radius
constexpr int HS = TS + 2 * SR;   // 42: halo edge
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

// Register-tiled separable window.  A workgroup owns a 32x32 output tile (42x42 halo in LDS, 1.7x the outputs
// instead of 2.6x for 16x16 tiles).  Horizontal pass: a thread produces 8 adjacent outputs of one halo row from 18
// inputs held in registers (11 x 8 FMAs per map instead of 11 x 8 x (2 LDS reads + FMA)); vertical pass: a thread
// produces 4 vertically adjacent outputs of one column from 14 values.  The first version read LDS for every tap:
// it was VALU + LDS bound at 61 us for a 4 x 3 x 512^2 batch against ~8 us of HBM time.
// The window as six VGPRs (it is symmetric: w[k] == w[10 - k] bit for bit).  As a kernel argument the weights live in
// SGPRs, and on gfx950 a VALU instruction with an SGPR source issues in 4 cycles instead of 2
// (profiles/r02_valu_issue_rates.txt): every one of the 11-tap FMAs paid that.  The asm keeps the compiler from
// folding the copies back into scalar operands.
struct WindowV {
  float w[6];
};
__device__ __forceinline__ WindowV window_to_vgprs(const Window &win) {
  WindowV v;
#pragma unroll
  for (int k = 0; k < 6; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(v.w[k]) : "s"(win.w[k]));
  return v;
}
template <int NOUT, int NIN>
__device__ __forceinline__ void taps(const WindowV &win, const float (&in)[NIN], float (&out)[NOUT]) {
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 11; ++k) a += win.w[k < 6 ? k : 10 - k] * in[o + k];
    out[o] = a;
  }
}

template <int NOUT, int NIN>
__device__ __forceinline__ void taps(const Window &win, const float (&in)[NIN], float (&out)[NOUT]) {
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 11; ++k) a += win.w[k] * in[o + k];
    out[o] = a;
  }
}

__global__ void __launch_bounds__(256) ssim_fwd_kernel(int H, int W, int n_planes, int clamp1, Window win,
                                                       const float *__restrict__ img1,
                                                       const float *__restrict__ img2,
                                                       float *__restrict__ ssim_sum, float *__restrict__ partials,
                                                       size_t plane_stride_total) {
  __shared__ float s_x[HS][HS + 1];
  __shared__ float s_y[HS][HS + 1];
  __shared__ float s_h[HS][TS + 1];  // ONE map at a time (20 KB of LDS in all: the kernel shares CUs with the renders)
  __shared__ float s_red[4];

  // A workgroup walks tiles (plane, ty, tx) with stride gridDim.x and keeps its part of the SSIM sum in registers:
  // one atomic per workgroup at the end (12288 same-address atomics were the first version's critical path).
  const int tid = threadIdx.x;
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  const int c = tid & (TS - 1), r0 = (tid >> 5) * 4;            // vertical pass: column c, rows r0 .. r0 + 3
  const bool hrow = tid < HS * (TS / 8);                         // horizontal pass: 42 rows x 4 groups of 8 columns
  const int ly = hrow ? tid / (TS / 8) : 0, c0 = (tid % (TS / 8)) * 8;
  float m_acc = 0.0f;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int plane = tile / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * TS, y0 = ((tile / tiles_x) % tiles_y) * TS;
    const float *p1 = img1 + (size_t)plane * H * W;
    const float *p2 = img2 + (size_t)plane * H * W;
    __syncthreads();  // the previous tile's LDS has been consumed
    for (int t = tid; t < HS * HS; t += 256) {
      const int hy = t / HS, hx = t - hy * HS;
      const int gy = y0 + hy - SR, gx = x0 + hx - SR;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      s_x[hy][hx] = in ? maybe_clamp(p1[(size_t)gy * W + gx], clamp1) : 0.0f;
      s_y[hy][hx] = in ? p2[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    float x[18], y[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) x[i] = s_x[ly][c0 + i], y[i] = s_y[ly][c0 + i];
    float st[5][4];
#pragma unroll
    for (int q = 0; q < 5; ++q) {  // maps: x, y, x^2, y^2, x y
      if (hrow) {
        float v[18], o[8];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = q == 0 ? x[i] : q == 1 ? y[i] : q == 2 ? x[i] * x[i] : q == 3 ? y[i] * y[i] : x[i] * y[i];
        taps<8, 18>(win, v, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) s_h[ly][c0 + i] = o[i];
      }
      __syncthreads();
      float col[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) col[i] = s_h[r0 + i][c];
      taps<4, 14>(win, col, st[q]);
      __syncthreads();
    }
    const int gx = x0 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i;
      if (gx < W && gy < H) {
        const float mu1 = st[0][i], mu2 = st[1][i], e11 = st[2][i], e22 = st[3][i], e12 = st[4][i];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float A1 = 2.0f * mu12 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
        const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
        const float inv = 1.0f / (B1 * B2);
        const float m = A1 * A2 * inv;
        m_acc += m;
        if (partials) {
          // total derivatives w.r.t. mu1, E[x^2], E[xy] (sigma terms expanded)
          const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
          partials[o] = (2.0f * mu2 * (A2 - A1) - m * 2.0f * mu1 * (B2 - B1)) * inv;
          partials[plane_stride_total + o] = -m / B2;
          partials[2 * plane_stride_total + o] = 2.0f * A1 * inv;
        }
      }
    }
  }  // tiles
  // block sum -> one atomic
  float v = m_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(int H, int W, int n_planes, int clamp1, Window win,
                                                       const float *__restrict__ img1,
                                                       const float *__restrict__ img2,
                                                       const float *__restrict__ partials,
                                                       size_t plane_stride_total,
                                                       const float *__restrict__ dL_dmean, float inv_numel,
                                                       float *__restrict__ dL_dimg1) {
  __shared__ float s_p[HS][HS + 1];
  __shared__ float s_h[HS][TS + 1];
  const int tid = threadIdx.x;
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  const float scale = dL_dmean[0] * inv_numel;
  const int c = tid & (TS - 1), r0 = (tid >> 5) * 4;
  const bool hrow = tid < HS * (TS / 8);
  const int ly = hrow ? tid / (TS / 8) : 0, c0 = (tid % (TS / 8)) * 8;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int plane = tile / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * TS, y0 = ((tile / tiles_x) % tiles_y) * TS;
    float st[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {  // one partial-derivative map at a time: 13 KB of LDS
      __syncthreads();
      for (int t = tid; t < HS * HS; t += 256) {
        const int hy = t / HS, hx = t - hy * HS;
        const int gy = y0 + hy - SR, gx = x0 + hx - SR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_p[hy][hx] = in ? partials[q * plane_stride_total + (size_t)plane * H * W + (size_t)gy * W + gx] : 0.0f;
      }
      __syncthreads();
      if (hrow) {
        float v[18], o[8];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = s_p[ly][c0 + i];
        taps<8, 18>(win, v, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) s_h[ly][c0 + i] = o[i];
      }
      __syncthreads();
      float col[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) col[i] = s_h[r0 + i][c];
      taps<4, 14>(win, col, st[q]);
    }
    const int gx = x0 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i;
      if (gx < W && gy < H) {
        const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
        const float x = maybe_clamp(img1[o], clamp1), y = img2[o];
        dL_dimg1[o] = (st[0][i] + 2.0f * x * st[1][i] + y * st[2][i]) * scale;
      }
    }
  }
}


// ---- forward value + gradient in ONE launch ----------------------------------------------------------------------
// The training loss needs both the SSIM mean and dL/dimg1 of every batch, and the gradient's upstream factor is a
// constant (-lambda / numel), so nothing forces the three derivative planes through HBM (3 planes written, then read
// back with their halos: 76 MB per 4 x 3 x 512^2 batch).  A workgroup computes the window statistics on the 42x42
// halo of its 32x32 tile from a 52x52 input halo (2.6x the pixels, all VALU), keeps the derivative planes in LDS,
// and applies the window a second time.  Positions outside the image carry zero derivatives (the zero padding of
// the backward convolution).
constexpr int IS = HS + 2 * SR;  // 52: input halo edge
// Optional per-image base pointers of img2 (the targets of a batch live in a resident pool, one tensor per image:
// stacking them cost two copy kernels per motion at the head of every step).  n == 0: img2 is one contiguous tensor.
constexpr int SSIM_MAX_IMAGES = 32;
struct ImagePtrs {
  int n, channels;
  const float *p[SSIM_MAX_IMAGES];
};
__global__ void __launch_bounds__(256, 4) ssim_fused_kernel(int H, int W, int n_planes, int clamp1, Window win,
                                                         const float *__restrict__ img1,
                                                         const float *__restrict__ img2, ImagePtrs img2_images,
                                                         const float *__restrict__ dL_dmean, float inv_numel,
                                                         float *__restrict__ ssim_sum, float *__restrict__ dL_dimg1) {
  // 31 KB of LDS per workgroup (53 KB in round 1: three workgroups then owned ALL of a CU's LDS, and in the
  // two-stream schedule no blend workgroup of the other motion could join them on that CU -- the kernel showed 200+ us
  // there against 77 us alone).  The derivative planes live over the input planes, which are dead once the five maps
  // are filtered; x and y of the tile's own pixels are re-read from global at the end.  128 VGPRs: FOUR workgroups
  // per CU (the phases of a tile are short and separated by barriers, so the kernel lives on the other workgroups'
  // waves: 80 -> 71 us for 8 x 3 x 512^2 against three per CU with x / y held in registers).
  __shared__ float s_xy[2][IS][IS + 1];
  __shared__ float s_h[IS][HS + 1];      // horizontally filtered map (52 x 42); reused as 42 x 32 in the second pass
  __shared__ float s_red[4];
  static_assert(3 * HS * (HS + 1) <= 2 * IS * (IS + 1), "derivative planes must fit over the input planes");
  float (*s_x)[IS + 1] = s_xy[0], (*s_y)[IS + 1] = s_xy[1];
  const WindowV winv = window_to_vgprs(win);
  float (*s_p)[HS][HS + 1] = reinterpret_cast<float (*)[HS][HS + 1]>(&s_xy[0][0][0]);  // derivative planes on the halo
  const int tid = threadIdx.x;
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  const float scale = dL_dmean[0] * inv_numel;
  // first pass, horizontal: 52 rows x 3 groups of 14 columns; vertical: 42 columns x 6 groups of 7 rows
  const bool h1 = tid < IS * 3;
  const int h1_row = h1 ? tid / 3 : 0, h1_c0 = (tid % 3) * 14;
  const bool v1 = tid < HS * 6;
  const int v1_c = v1 ? tid % HS : 0, v1_r0 = v1 ? (tid / HS) * 7 : 0;
  // second pass (as ssim_bwd_kernel): horizontal 42 rows x 4 groups of 8; vertical column c, rows r0 .. r0 + 3
  const bool h2 = tid < HS * (TS / 8);
  const int h2_row = h2 ? tid / (TS / 8) : 0, h2_c0 = (tid % (TS / 8)) * 8;
  const int c = tid & (TS - 1), r0 = (tid >> 5) * 4;
  float m_acc = 0.0f;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int plane = tile / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * TS, y0 = ((tile / tiles_x) % tiles_y) * TS;
    const float *p1 = img1 + (size_t)plane * H * W;
    const float *p2 = img2_images.n ? img2_images.p[plane / img2_images.channels] +
                                          (size_t)(plane % img2_images.channels) * H * W
                                    : img2 + (size_t)plane * H * W;
    __syncthreads();  // the previous tile's LDS has been consumed
    for (int t = tid; t < IS * IS; t += 256) {
      const int hy = t / IS, hx = t - hy * IS;
      const int gy = y0 + hy - 2 * SR, gx = x0 + hx - 2 * SR;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const size_t o = in ? (size_t)gy * W + gx : 0;
      const float a = p1[o], b = p2[o];
      s_x[hy][hx] = in ? maybe_clamp(a, clamp1) : 0.0f;
      s_y[hy][hx] = in ? b : 0.0f;
    }
    __syncthreads();
    float st[5][7];
#pragma unroll
    for (int q = 0; q < 5; ++q) {  // maps: x, y, x^2, y^2, x y
      if (h1) {
        float v[24], o[14];
#pragma unroll
        for (int i = 0; i < 24; ++i) {  // (re-read from LDS per map: holding x and y cost 48 VGPRs = one wave per SIMD)
          const float a = (q == 1 || q == 3) ? s_y[h1_row][h1_c0 + i] : s_x[h1_row][h1_c0 + i];
          v[i] = q < 2 ? a : q < 4 ? a * a : a * s_y[h1_row][h1_c0 + i];
        }
        taps<14, 24>(winv, v, o);
#pragma unroll
        for (int i = 0; i < 14; ++i) s_h[h1_row][h1_c0 + i] = o[i];
      }
      __syncthreads();
      float col[17];
#pragma unroll
      for (int i = 0; i < 17; ++i) col[i] = s_h[v1_r0 + i][v1_c];
      taps<7, 17>(winv, col, st[q]);
      __syncthreads();
    }
    if (v1) {
      const int gx = x0 + v1_c - SR;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int hy = v1_r0 + i, gy = y0 + hy - SR;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const float mu1 = st[0][i], mu2 = st[1][i], e11 = st[2][i], e22 = st[3][i], e12 = st[4][i];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float A1 = 2.0f * mu12 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
        const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
        const float inv = 1.0f / (B1 * B2);
        const float m = A1 * A2 * inv;
        const bool own = hy >= SR && hy < SR + TS && v1_c >= SR && v1_c < SR + TS;  // this tile's 32 x 32 outputs
        if (in && own) m_acc += m;
        s_p[0][hy][v1_c] = in ? (2.0f * mu2 * (A2 - A1) - m * 2.0f * mu1 * (B2 - B1)) * inv : 0.0f;
        s_p[1][hy][v1_c] = in ? -m / B2 : 0.0f;
        s_p[2][hy][v1_c] = in ? 2.0f * A1 * inv : 0.0f;
      }
    }
    float g[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __syncthreads();  // s_p complete (q = 0) / s_h free again
      if (h2) {
        float v[18], o[8];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = s_p[q][h2_row][h2_c0 + i];
        taps<8, 18>(winv, v, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) s_h[h2_row][h2_c0 + i] = o[i];
      }
      __syncthreads();
      float col[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) col[i] = s_h[r0 + i][c];
      taps<4, 14>(winv, col, g[q]);
    }
    const int gx = x0 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i;
      if (gx < W && gy < H) {
        const size_t o = (size_t)gy * W + gx;
        const float xv = maybe_clamp(p1[o], clamp1), yv = p2[o];
        dL_dimg1[(size_t)plane * H * W + o] = (g[0][i] + 2.0f * xv * g[1][i] + yv * g[2][i]) * scale;
      }
    }
  }  // tiles
  float v = m_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
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
  const long tiles = (long)((W + TS - 1) / TS) * ((H + TS - 1) / TS) * planes;
  const dim3 grid((unsigned)(tiles < 4096 ? tiles : 4096)), block(256);
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
  const long tiles = (long)((W + TS - 1) / TS) * ((H + TS - 1) / TS) * planes;
  const dim3 grid((unsigned)(tiles < 4096 ? tiles : 4096)), block(256);
  ScopedTimer tm(T_SSIM_BWD, stream);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, stream, H, W, (int)planes, clamp_img1, win, img1, img2, partials,
                     (size_t)planes * H * W, dL_dmean, 1.0f / (float)((double)planes * H * W), dL_dimg1);
  return check_launch();
}

static int ssim_forward_backward_impl(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                                      const float *const *img2_images_host, const float *dL_dmean, float *ssim_sum,
                                      float *dL_dimg1, void *stream_);
extern "C" int dimo_ssim_forward_backward(int B, int C, int H, int W, int clamp_img1, const float *img1,
                                          const float *img2, const float *dL_dmean, float *ssim_sum, float *dL_dimg1,
                                          void *stream_) {
  return ssim_forward_backward_impl(B, C, H, W, clamp_img1, img1, img2, nullptr, dL_dmean, ssim_sum, dL_dimg1, stream_);
}
extern "C" int dimo_ssim_forward_backward_images(int B, int C, int H, int W, int clamp_img1, const float *img1,
                                                 const float *const *img2_images_host, const float *dL_dmean,
                                                 float *ssim_sum, float *dL_dimg1, void *stream_) {
  if (!img2_images_host || B > SSIM_MAX_IMAGES) return DIMO_E_ARG;
  return ssim_forward_backward_impl(B, C, H, W, clamp_img1, img1, nullptr, img2_images_host, dL_dmean, ssim_sum,
                                    dL_dimg1, stream_);
}
static int ssim_forward_backward_impl(int B, int C, int H, int W, int clamp_img1, const float *img1, const float *img2,
                                      const float *const *img2_images_host, const float *dL_dmean, float *ssim_sum,
                                      float *dL_dimg1, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (B < 0 || C < 0 || H <= 0 || W <= 0 || !ssim_sum) return DIMO_E_ARG;
  // bit 1 of clamp_img1: the caller zeroed *ssim_sum together with its other accumulators (a 4-byte memset is a
  // launch of its own on the step's critical path)
  const bool prezeroed = (clamp_img1 & 2) != 0;
  clamp_img1 &= 1;
  if (!prezeroed && hipMemsetAsync(ssim_sum, 0, sizeof(float), stream) != hipSuccess) return DIMO_E_LAUNCH;
  const long planes = (long)B * C;
  if (planes == 0) return DIMO_OK;
  if (!img1 || (!img2 && !img2_images_host) || !dL_dmean || !dL_dimg1 || planes > 65535) return DIMO_E_ARG;
  ImagePtrs ptrs;
  ptrs.n = img2_images_host ? B : 0, ptrs.channels = C > 0 ? C : 1;
  for (int b = 0; b < SSIM_MAX_IMAGES; ++b) ptrs.p[b] = (img2_images_host && b < B) ? img2_images_host[b] : nullptr;
  for (int b = 0; b < ptrs.n; ++b)
    if (!ptrs.p[b]) return DIMO_E_ARG;
  static const Window win = make_window();
  const long tiles = (long)((W + TS - 1) / TS) * ((H + TS - 1) / TS) * planes;
  const dim3 grid((unsigned)(tiles < 4096 ? tiles : 4096)), block(256);
  ScopedTimer tm(T_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fused_kernel, grid, block, 0, stream, H, W, (int)planes, clamp_img1, win, img1, img2, ptrs,
                     dL_dmean, 1.0f / (float)((double)planes * H * W), ssim_sum, dL_dimg1);
  return check_launch();
}
