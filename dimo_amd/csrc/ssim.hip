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
#include <stdlib.h>

#include "common.hpp"
#include "loss_terms.hpp"

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

// SSIM of one pixel from the filtered maps mu1 = E[x], mu2 = E[y], e_sq = E[x^2 + y^2], e12 = E[xy], and its total
// derivatives w.r.t. mu1, E[x^2] and E[xy] (sigma terms expanded; src/loss.py:144-175).  The reciprocals are
// v_rcp_f32 (1 ulp) instead of IEEE divisions (10 instructions each, twice per pixel of the halo).
struct SsimPoint {
  float m, d_mu1, d_e11, d_e12;
};
__device__ __forceinline__ SsimPoint ssim_point(float mu1, float mu2, float e_sq, float e12) {
  const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
  const float s12 = e12 - mu12;
  const float A1 = 2.0f * mu12 + SSIM_C1, A2 = 2.0f * s12 + SSIM_C2;
  const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = (e_sq - mu1_sq - mu2_sq) + SSIM_C2;
  const float inv = __builtin_amdgcn_rcpf(B1 * B2);
  SsimPoint r;
  r.m = A1 * A2 * inv;
  r.d_mu1 = (2.0f * mu2 * (A2 - A1) - r.m * 2.0f * mu1 * (B2 - B1)) * inv;
  r.d_e11 = -r.m * __builtin_amdgcn_rcpf(B2);
  r.d_e12 = 2.0f * A1 * inv;
  return r;
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
    float st[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // maps: x, y, x^2 + y^2, x y (the two variances only ever appear as their sum)
      if (hrow) {
        float v[18], o[8];
#pragma unroll
        for (int i = 0; i < 18; ++i) v[i] = q == 0 ? x[i] : q == 1 ? y[i] : q == 2 ? x[i] * x[i] + y[i] * y[i] : x[i] * y[i];
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
        const SsimPoint sp = ssim_point(st[0][i], st[1][i], st[2][i], st[3][i]);
        m_acc += sp.m;
        if (partials) {
          const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
          partials[o] = sp.d_mu1;
          partials[plane_stride_total + o] = sp.d_e11;
          partials[2 * plane_stride_total + o] = sp.d_e12;
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
//
// The kernel is bound by the LDS pipe, not by its FMAs (profiles/r03_ssim_bisection.txt: with every tap removed
// it still took 58 of 66 us; with the global loads removed as well, 50).  So the first window pass runs VERTICALLY
// straight from registers: a thread loads 21 rows of one input column from global memory (lanes along the row:
// coalesced), filters the four maps and writes 11 rows of each -- the inputs never visit the LDS.  The horizontal pass
// then reads rows with ds_read_b128 (row stride 52 floats = 13 x 16 B: the 16 lanes of a read group hold 16 different
// rows and fall on 16 different 16-byte slots) and keeps the statistics of its 8 pixels in registers.  LDS cycles per
// tile (tools/lds_rate.hip): ~2400 against ~5100 for the version that staged the inputs and filtered horizontally
// first; barriers per tile 9 against 16.  66 -> 51 us for 4 x 3 x 512^2, 103 -> 92 us for 8 x 3 x 512^2.
constexpr int IS = HS + 2 * SR;  // 52: input halo edge
constexpr int MS = IS;           // row stride of a vertically filtered map (42 rows x 52 columns)
constexpr int MAP_WORDS = HS * MS;
constexpr int PS = 52;           // row stride of a derivative plane on the 42 x 42 halo (same slot argument)
constexpr int V_ROWS = 11;       // rows of the 42 a thread of the vertical pass produces (from 21 input rows)
// THREE workgroups per CU, persistent (grid = 3 x 256 CUs).  Round 3's kernel needed 168 VGPRs (at 128, four per CU,
// its vertical pass spilled 61: 83 us against 51); with the tile body as a device function (round 5) it needs 98, and
// four per CU -- all of a CU's LDS -- measured the same 44 us alone and 0.3 % less in the step
// (profiles/r05_one_pass_losses.txt): three stay.  One tile per workgroup instead of the persistent loop: 59 us.
constexpr int SSIM_WGS_PER_CU = 3;
// Optional per-image base pointers of img2 (the targets of a batch live in a resident pool, one tensor per image:
// stacking them cost two copy kernels per motion at the head of every step).  n == 0: img2 is one contiguous tensor.
constexpr int SSIM_MAX_IMAGES = 32;
struct ImagePtrs {
  int n, channels;
  const float *p[SSIM_MAX_IMAGES];
};
// Per-thread index maps of the fused tile body (constant over the tiles a workgroup walks).
struct SsimThread {
  bool v1, h1, h2;
  int v1_c, v1_r0, h1_row, h1_c0, h2_row, h2_c0, c, r0;
};
__device__ __forceinline__ SsimThread ssim_thread(int tid) {
  SsimThread t;
  // first pass, vertical: 52 columns x 4 row groups (rows 0-10, 11-21, 22-32, 31-41 of the 42: the last overlaps)
  t.v1 = tid < IS * 4;
  t.v1_c = tid % IS, t.v1_r0 = min((tid / IS) * V_ROWS, HS - V_ROWS);
  // first pass, horizontal: 42 rows x 6 groups of 8 columns (the last group: columns 40, 41 and six unused ones)
  t.h1 = tid < HS * 6;
  t.h1_row = tid % HS, t.h1_c0 = (tid / HS) * 8;
  // second pass, horizontal: 42 rows x 4 groups of 8; vertical: column c, rows r0 .. r0 + 3
  t.h2 = tid < HS * (TS / 8);
  t.h2_row = tid % HS, t.h2_c0 = (tid / HS) * 8;
  t.c = tid & (TS - 1), t.r0 = (tid >> 5) * 4;
  return t;
}

// One 32 x 32 tile of one (image, channel) plane: SSIM map sum of the tile's pixels added to `m_acc`, and the gradient
// of the map's sum w.r.t. img1 at this thread's four pixels (column x0 + c, rows y0 + r0 .. + 3) returned in `grad`
// (unscaled), with the pixels' own x (clamped if asked) and y in `xo` / `yo`.  LDS: s_m (four maps, then the derivative
// planes over them) and s_h; the caller's NEXT use of either must come after a barrier.
__device__ __forceinline__ void ssim_tile(const SsimThread &t, const WindowV &winv, float *s_m, float (*s_h)[TS + 1],
                                          int H, int W, int x0, int y0, const float *__restrict__ p1,
                                          const float *__restrict__ p2, int clamp1, float &m_acc, float (&grad)[4],
                                          float (&xo)[4], float (&yo)[4]) {
  float *const s_p = s_m;  // [3][HS][PS]
  float xs[2 * V_ROWS - 1], ys[2 * V_ROWS - 1];  // 21 input rows of this thread's column
  // Requests the inputs of the tile: address = scalar row base + one 32-bit lane offset for all 21 rows.  Tiles whose
  // 52 x 52 halo lies inside the image (three in four at 512^2) load unconditionally; the others mask the lanes
  // outside off, and nothing inside the masked region may USE a loaded value (a use makes the loads wait for one
  // another: 5 us for the 42 -- the clamp of img1 is applied when the vertical pass reads the registers).
  if (t.v1) {
    const int gx = x0 + t.v1_c - 2 * SR, gy0 = y0 + t.v1_r0 - 2 * SR;
    const unsigned off = 4u * (unsigned)(t.v1_r0 * W + gx);
    const bool interior = x0 >= 2 * SR && x0 + TS + 2 * SR <= W && y0 >= 2 * SR && y0 + TS + 2 * SR <= H;
    if (interior) {
#pragma unroll
      for (int i = 0; i < 2 * V_ROWS - 1; ++i) {
        const ptrdiff_t row = (ptrdiff_t)(y0 + i - 2 * SR) * W;
        xs[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p1 + row) + off);
        ys[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p2 + row) + off);
      }
    } else {
      const bool in_x = gx >= 0 && gx < W;
#pragma unroll
      for (int i = 0; i < 2 * V_ROWS - 1; ++i) {
        const ptrdiff_t row = (ptrdiff_t)(y0 + i - 2 * SR) * W;
        float a = 0.0f, b = 0.0f;
        if (in_x && gy0 + i >= 0 && gy0 + i < H) {
          a = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p1 + row) + off);
          b = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p2 + row) + off);
        }
        xs[i] = a, ys[i] = b;
      }
    }
  }
  lds_barrier();  // the previous tile's LDS has been consumed
  if (t.v1) {
#pragma unroll
    for (int i = 0; i < 2 * V_ROWS - 1; ++i) xs[i] = maybe_clamp(xs[i], clamp1);  // (0 outside the image stays 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // maps: x, y, x^2 + y^2, x y (the two variances only ever appear as their sum)
      float v[2 * V_ROWS - 1], o[V_ROWS];
#pragma unroll
      for (int i = 0; i < 2 * V_ROWS - 1; ++i)
        v[i] = q == 0 ? xs[i] : q == 1 ? ys[i] : q == 2 ? xs[i] * xs[i] + ys[i] * ys[i] : xs[i] * ys[i];
      taps<V_ROWS, 2 * V_ROWS - 1>(winv, v, o);
#pragma unroll
      for (int j = 0; j < V_ROWS; ++j) s_m[q * MAP_WORDS + (t.v1_r0 + j) * MS + t.v1_c] = o[j];
    }
  }
  lds_barrier();
  float d[3][8];  // derivative planes of this thread's 8 halo pixels
  if (t.h1) {
    float st[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float in[20];
      const float4 *row = reinterpret_cast<const float4 *>(&s_m[q * MAP_WORDS + t.h1_row * MS + t.h1_c0]);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float4 v4 = row[i];
        in[4 * i] = v4.x, in[4 * i + 1] = v4.y, in[4 * i + 2] = v4.z, in[4 * i + 3] = v4.w;
      }
      taps<8, 20>(winv, in, st[q]);
    }
    const int gy = y0 + t.h1_row - SR;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int hx = t.h1_c0 + i, gx = x0 + hx - SR;
      const bool in = hx < HS && gx >= 0 && gx < W && gy >= 0 && gy < H;
      const SsimPoint sp = ssim_point(st[0][i], st[1][i], st[2][i], st[3][i]);
      const bool own = t.h1_row >= SR && t.h1_row < SR + TS && hx >= SR && hx < SR + TS;  // this tile's 32 x 32 outputs
      if (in && own) m_acc += sp.m;
      d[0][i] = in ? sp.d_mu1 : 0.0f, d[1][i] = in ? sp.d_e11 : 0.0f, d[2][i] = in ? sp.d_e12 : 0.0f;
    }
  }
  lds_barrier();  // every map has been read: the derivative planes may overwrite them
  if (t.h1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float4 *row = reinterpret_cast<float4 *>(&s_p[(k * HS + t.h1_row) * PS + t.h1_c0]);
      row[0] = make_float4(d[k][0], d[k][1], d[k][2], d[k][3]);
      row[1] = make_float4(d[k][4], d[k][5], d[k][6], d[k][7]);
    }
  }
  // x and y of this thread's four output pixels (used after the second pass): the statistics' registers are free now
  // (requesting right after the vertical pass spilled: 48 -> 65 us)
  const int gx = x0 + t.c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gy = y0 + t.r0 + i;
    const int o = (gx < W && gy < H) ? gy * W + gx : 0;
    xo[i] = p1[o], yo[i] = p2[o];
  }
  float g[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q) {  // (sharing the barriers between the planes, with a buffer each, changed nothing)
    lds_barrier();  // s_p complete (q = 0) / s_h free again
    if (t.h2) {
      float in[20], o[8];
      const float4 *row = reinterpret_cast<const float4 *>(&s_p[(q * HS + t.h2_row) * PS + t.h2_c0]);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float4 v4 = row[i];
        in[4 * i] = v4.x, in[4 * i + 1] = v4.y, in[4 * i + 2] = v4.z, in[4 * i + 3] = v4.w;
      }
      taps<8, 20>(winv, in, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) s_h[t.h2_row][t.h2_c0 + i] = o[i];
    }
    lds_barrier();
    float col[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) col[i] = s_h[t.r0 + i][t.c];
    taps<4, 14>(winv, col, g[q]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xo[i] = maybe_clamp(xo[i], clamp1);
    grad[i] = g[0][i] + 2.0f * xo[i] * g[1][i] + yo[i] * g[2][i];
  }
}

template <int WGS>
__global__ void __launch_bounds__(256, WGS) ssim_fused_kernel(int H, int W, int n_planes, int clamp1, Window win,
                                                         const float *__restrict__ img1,
                                                         const float *__restrict__ img2, ImagePtrs img2_images,
                                                         const float *__restrict__ dL_dmean, float inv_numel,
                                                         float *__restrict__ ssim_sum, float *__restrict__ dL_dimg1) {
  // 40 KB of LDS per workgroup (three workgroups leave 39 KB of a CU's LDS to the other motion's blend workgroups in
  // the two-stream schedule).  The derivative planes live over the filtered maps, which are dead once the statistics
  // are in registers; x and y of the tile's own pixels are re-read from global before the second pass.
  __shared__ __attribute__((aligned(16))) float s_m[4 * MAP_WORDS + 8];  // four maps (+ the last row's over-read)
  __shared__ float s_h[HS][TS + 1];  // second pass: horizontally filtered derivative plane (42 x 32)
  __shared__ float s_red[4];
  static_assert(3 * HS * PS <= 4 * MAP_WORDS, "derivative planes must fit over the filtered maps");
  const WindowV winv = window_to_vgprs(win);
  const int tid = threadIdx.x;
  const SsimThread t = ssim_thread(tid);
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  const float scale = dL_dmean[0] * inv_numel;
  float m_acc = 0.0f;
  // The workgroups are persistent (three per CU).  Requesting the NEXT tile's inputs early (after the vertical pass,
  // or before the second window pass) was slower: 42 more live registers, and the requests' issue slots are the cost,
  // not their latency (55-65 us against 51).  Barriers order LDS traffic only (lds_barrier).
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int plane = tile / (tiles_x * tiles_y);
    const int x0 = (tile % tiles_x) * TS, y0 = ((tile / tiles_x) % tiles_y) * TS;
    const float *p1 = img1 + (size_t)plane * H * W;
    const float *p2 = img2_images.n ? img2_images.p[plane / img2_images.channels] +
                                          (size_t)(plane % img2_images.channels) * H * W
                                    : img2 + (size_t)plane * H * W;
    float grad[4], xo[4], yo[4];
    ssim_tile(t, winv, s_m, s_h, H, W, x0, y0, p1, p2, clamp1, m_acc, grad, xo, yo);
    const int gx = x0 + t.c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + t.r0 + i;
      if (gx < W && gy < H) dL_dimg1[(size_t)plane * H * W + (size_t)gy * W + gx] = grad[i] * scale;
    }
  }  // tiles
  float v = m_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// ---- SSIM + every other image term of a motion's batch in ONE tile pass -------------------------------------------
// (VERDICT rounds 2-4.)  A workgroup takes a 32 x 32 tile of ONE image through the SSIM tile body three times (one colour
// plane each), keeping per pixel and channel the gradient that does not depend on the neighbours -- SSIM + weighted MSE
// on the clamped image, main_train_dimo.py:331-343 -- in registers, then evaluates the mask term and the two smoothness
// terms (src/loss.py:64-106) of the tile from a 34 x 34 neighbourhood of the seven planes they read (clamped colour,
// depth, normal), staged in the LDS the SSIM maps have left, and writes the four gradient images and the blend backward's
// per-pixel S plane.  Against ssim_fused + image_loss: the SSIM gradient never leaves the registers (3 planes written and
// read back) and image / targets are read by one kernel instead of two (6 planes): 21 plane passes per image instead of
// 33, one launch instead of two.  A pair of neighbouring pixels that straddles a tile border is evaluated by both tiles
// (each keeps its own pixel's gradient; the pair's VALUE belongs to the tile of its right / lower pixel); inside a
// tile a thread owns a column of four pixels: its three inner vertical pairs are evaluated once, the others per pixel
// (3.25 evaluations per pixel against 2 in image_loss.hip's row walk).
constexpr int NB = TS + 2;        // neighbourhood edge
constexpr int NBS = NB + 1;       // its row stride in LDS
constexpr int NB_PLANE = NB * NBS;
static_assert(7 * NB_PLANE <= 4 * MAP_WORDS, "the neighbourhood must fit over the SSIM maps");

template <bool DEPTH, bool NORMAL, int WGS>
__global__ void __launch_bounds__(256, WGS) ssim_loss_tile_kernel(
    int H, int W, int n_images, Window win, LossParams prm, const float *__restrict__ image,
    const float *__restrict__ depth, const float *__restrict__ normal, const float *__restrict__ alpha,
    const float *__restrict__ gt, const float *__restrict__ mask, size_t mask_stride,
    const float *__restrict__ ssim_coef, float ssim_inv_numel, float *__restrict__ ssim_sum,
    float *__restrict__ loss_out, float *__restrict__ g_image, float *__restrict__ g_depth,
    float *__restrict__ g_normal, float *__restrict__ g_alpha, float *__restrict__ g_dot) {
  __shared__ __attribute__((aligned(16))) float s_m[4 * MAP_WORDS + 8];
  __shared__ float s_h[HS][TS + 1];
  __shared__ float s_red[8];
  const WindowV winv = window_to_vgprs(win);
  const int tid = threadIdx.x;
  const SsimThread t = ssim_thread(tid);
  const int HW = H * W;
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total = tiles_x * tiles_y * n_images;
  const float scale = ssim_coef[0] * ssim_inv_numel;
  float m_acc = 0.0f, loss = 0.0f;
  for (int unit = blockIdx.x; unit < total; unit += gridDim.x) {
    const int b = unit / (tiles_x * tiles_y);
    const int x0 = (unit % tiles_x) * TS, y0 = ((unit / tiles_x) % tiles_y) * TS;
    const float *img = image + (size_t)b * 3 * HW;
    const float *gtb = prm.gt_image[b] ? prm.gt_image[b] : gt + (size_t)b * 3 * HW;
    const float wm = prm.w_mse[b];
    const int gx = x0 + t.c;
    // ---- per channel: SSIM value + gradient, weighted MSE.  The per-pixel gradient of the clamped colour goes to its
    // final place in g_image and is picked up again below (a thread re-reads what it wrote; the barrier in between
    // drains its stores): kept in registers across the three unrolled tile bodies it spilled by the hundred
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
      float grad[4], xo[4], yo[4];
      ssim_tile(t, winv, s_m, s_h, H, W, x0, y0, img + ch * HW, gtb + ch * HW, 1, m_acc, grad, xo, yo);
      float *gout = g_image + ((size_t)b * 3 + ch) * HW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gy = y0 + t.r0 + i;
        const float e = xo[i] - yo[i];
        if (gx < W && gy < H) {
          loss += wm * e * e;
          gout[gy * W + gx] = 2.0f * wm * e + grad[i] * scale;
        }
      }
    }
    // ---- the neighbourhood of the pair terms: seven planes x 34 x 34 over the SSIM maps (dead now; the barrier also
    // covers s_h)
    float *const s_nb = s_m;
    const float *dep = DEPTH ? depth + (size_t)b * HW : nullptr;
    const float *nrm = NORMAL ? normal + (size_t)b * 3 * HW : nullptr;
    __syncthreads();  // (LDS free; this thread's g_image stores have landed)
    if (DEPTH || NORMAL) {
      for (int p = tid; p < NB * NB; p += 256) {
        const int hy = p / NB, hx = p - hy * NB;
        const int gyy = min(max(y0 + hy - 1, 0), H - 1), gxx = min(max(x0 + hx - 1, 0), W - 1);
        const int o = gyy * W + gxx;
        float *d = s_nb + hy * NBS + hx;
        d[0] = clamp01(img[o]), d[NB_PLANE] = clamp01(img[HW + o]), d[2 * NB_PLANE] = clamp01(img[2 * HW + o]);
        d[3 * NB_PLANE] = DEPTH ? dep[o] : 0.0f;
        d[4 * NB_PLANE] = NORMAL ? nrm[o] : 0.0f;
        d[5 * NB_PLANE] = NORMAL ? nrm[HW + o] : 0.0f;
        d[6 * NB_PLANE] = NORMAL ? nrm[2 * HW + o] : 0.0f;
      }
      lds_barrier();
    }
    // ---- this thread's column of four pixels
    const float *alp = alpha + (size_t)b * HW;
    const float *mkb = prm.mask_image[b] ? prm.mask_image[b] : mask + (size_t)b * mask_stride;
    auto px_at = [&](int hy, int hx) {
      Px P;
      const float *d = s_nb + hy * NBS + hx;
      P.c[0] = d[0], P.c[1] = d[NB_PLANE], P.c[2] = d[2 * NB_PLANE];
      P.d = d[3 * NB_PLANE];
      P.n[0] = d[4 * NB_PLANE], P.n[1] = d[5 * NB_PLANE], P.n[2] = d[6 * NB_PLANE];
      return P;
    };
    // Row by row down the column (two pixels and two gradient sets live, not four): row k's vertical pair with the
    // row above completes pixel k - 1, which is then written.
    auto finish = [&](int i, const Px &P, const Grad &gg) {  // pixel i of the column: mask term, clamp backward, stores
      const int gy = y0 + t.r0 + i;
      if (gx < W && gy < H) {
        const unsigned off = 4u * (unsigned)(gy * W + gx);
        const float a = ld(alp, off), em = a - ld(mkb, off);
        loss += prm.w_mask * em * em;
        const float ga = 2.0f * prm.w_mask * em;
        float dot = ga * a;
        st(g_alpha + (size_t)b * HW, off, ga);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float raw = ld(img + k * HW, off);
          const float gk = (raw >= 0.0f && raw <= 1.0f) ? gg.c[k] : 0.0f;  // clamp backward
          st(g_image + (size_t)b * 3 * HW + k * HW, off, gk);
          dot += gk * raw;
        }
        if (DEPTH) st(g_depth + (size_t)b * HW, off, gg.d), dot += gg.d * P.d;
        if (NORMAL) {
#pragma unroll
          for (int k = 0; k < 3; ++k) st(g_normal + (size_t)b * 3 * HW + k * HW, off, gg.n[k]), dot += gg.n[k] * P.n[k];
        }
        if (g_dot) st(g_dot + (size_t)b * HW, off, dot);
      }
    };
    auto start = [&](int i, Grad &gg) {  // the per-pixel part of pixel i's colour gradient, left in g_image above
      const int gy = y0 + t.r0 + i;
      const unsigned off = 4u * (unsigned)((gx < W && gy < H) ? gy * W + gx : 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) gg.c[k] = ld(g_image + ((size_t)b * 3 + k) * HW, off);
      gg.d = gg.n[0] = gg.n[1] = gg.n[2] = 0.0f;
    };
    if (DEPTH || NORMAL) {
      const bool col_in = gx < W;
      Px up = px_at(t.r0, t.c + 1);  // the row above the column
      Grad gprev;
      gprev.c[0] = gprev.c[1] = gprev.c[2] = gprev.d = gprev.n[0] = gprev.n[1] = gprev.n[2] = 0.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {  // k = 4: the row below the column (its pair completes pixel 3)
        const Px lo = px_at(t.r0 + 1 + k, t.c + 1);
        Grad gcur;
        if (k < 4) start(k, gcur);
        const int yl = y0 + t.r0 + k;  // image row of the pair's lower pixel
        const bool pair_y = col_in && yl >= 1 && yl < H;
        Grad tt;
        const float l = pair_term<DEPTH, NORMAL>(up, lo, pair_y ? prm.w_smooth_y : 0.0f, pair_y ? prm.w_bilat_y : 0.0f, tt);
        if (k < 4) loss += l, add(gcur, tt, -1.0f);  // (counted with its lower pixel, an own pixel)
        if (k > 0) {
          add(gprev, tt, 1.0f);
          finish(k - 1, up, gprev);
        }
        if (k < 4) {
          // horizontal pairs of own pixel k: (x - 1, x), counted, and (x, x + 1), gradient only
          const bool row_in = yl < H;
          const bool pair_l = row_in && col_in && gx >= 1, pair_r = row_in && gx + 1 < W;
          const Px L = px_at(t.r0 + 1 + k, t.c);
          loss += pair_term<DEPTH, NORMAL>(L, lo, pair_l ? prm.w_smooth_x : 0.0f, pair_l ? prm.w_bilat_x : 0.0f, tt);
          add(gcur, tt, -1.0f);
          const Px R = px_at(t.r0 + 1 + k, t.c + 2);
          (void)pair_term<DEPTH, NORMAL>(lo, R, pair_r ? prm.w_smooth_x : 0.0f, pair_r ? prm.w_bilat_x : 0.0f, tt);
          add(gcur, tt, 1.0f);
        }
        gprev = gcur, up = lo;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Grad gg;
        start(i, gg);
        Px P;
        P.c[0] = P.c[1] = P.c[2] = P.d = P.n[0] = P.n[1] = P.n[2] = 0.0f;
        finish(i, P, gg);
      }
    }
  }  // units
  float v = m_acc, w = loss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64), w += __shfl_down(w, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = v, s_red[4 + (tid >> 6)] = w;
  __syncthreads();
  if (tid == 0) {
    atomicAdd(ssim_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    atomicAdd(loss_out + (blockIdx.x & 15) * 32, s_red[4] + s_red[5] + s_red[6] + s_red[7]);
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
  const int wgs = SSIM_WGS_PER_CU;
  const dim3 grid((unsigned)(tiles < wgs * 256 ? tiles : wgs * 256)), block(256);
  ScopedTimer tm(T_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fused_kernel<SSIM_WGS_PER_CU>, grid, block, 0, stream, H, W, (int)planes, clamp_img1, win,
                     img1, img2, ptrs, dL_dmean, 1.0f / (float)((double)planes * H * W), ssim_sum, dL_dimg1);
  return check_launch();
}

extern "C" int dimo_ssim_image_loss(int B, int H, int W, const float *image, const float *depth, const float *normal,
                                    const float *alpha, const float *gt, const float *mask, int mask_per_image,
                                    const float *w_mse_host, float w_mask, float w_smooth_x, float w_smooth_y,
                                    float w_bilat_x, float w_bilat_y, const float *ssim_coef, float *ssim_sum,
                                    float *loss_accum, float *g_image, float *g_depth, float *g_normal, float *g_alpha,
                                    float *g_dot, const float *const *gt_images_host,
                                    const float *const *mask_images_host, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (B < 0 || B > LOSS_MAX_B || H <= 0 || W <= 0) return DIMO_E_ARG;
  if (B == 0) return DIMO_OK;
  if (!image || !alpha || (!gt && !gt_images_host) || (!mask && !mask_images_host) || !w_mse_host || !loss_accum ||
      !g_image || !g_alpha || !ssim_coef || !ssim_sum)
    return DIMO_E_ARG;
  if ((depth == nullptr) != (g_depth == nullptr) || (normal == nullptr) != (g_normal == nullptr)) return DIMO_E_ARG;
  if ((long)H * W > (1L << 28)) return DIMO_E_ARG;  // byte offsets inside an image are 32-bit
  LossParams prm;
  for (int b = 0; b < LOSS_MAX_B; ++b) {
    prm.w_mse[b] = b < B ? w_mse_host[b] : 0.0f;
    prm.gt_image[b] = (gt_images_host && b < B) ? gt_images_host[b] : nullptr;
    prm.mask_image[b] = (mask_images_host && b < B) ? mask_images_host[b] : nullptr;
    if (b < B && ((gt_images_host && !prm.gt_image[b]) || (mask_images_host && !prm.mask_image[b]))) return DIMO_E_ARG;
  }
  prm.w_mask = w_mask, prm.w_smooth_x = w_smooth_x, prm.w_smooth_y = w_smooth_y;
  prm.w_bilat_x = w_bilat_x, prm.w_bilat_y = w_bilat_y;
  static const Window win = make_window();
  const long units = (long)((W + TS - 1) / TS) * ((H + TS - 1) / TS) * B;
  const int wgs = SSIM_WGS_PER_CU;
  const dim3 grid((unsigned)(units < wgs * 256 ? units : wgs * 256)), block(256);
  const size_t mstride = mask_per_image ? (size_t)H * W : 0;
  const float inv_numel = 1.0f / (float)((double)B * 3 * H * W);
  ScopedTimer tm(T_SSIM_FWD, stream);
#define DIMO_LAUNCH_SL2(D, N, G)                                                                                       \
  hipLaunchKernelGGL((ssim_loss_tile_kernel<D, N, G>), grid, block, 0, stream, H, W, B, win, prm, image, depth, normal, \
                     alpha, gt, mask, mstride, ssim_coef, inv_numel, ssim_sum, loss_accum, g_image, g_depth, g_normal, \
                     g_alpha, g_dot)
#define DIMO_LAUNCH_SL(D, N) DIMO_LAUNCH_SL2(D, N, SSIM_WGS_PER_CU)
  if (depth && normal) { DIMO_LAUNCH_SL(true, true); }
  else if (depth) { DIMO_LAUNCH_SL(true, false); }
  else if (normal) { DIMO_LAUNCH_SL(false, true); }
  else { DIMO_LAUNCH_SL(false, false); }
#undef DIMO_LAUNCH_SL2
#undef DIMO_LAUNCH_SL
  return check_launch();
}
