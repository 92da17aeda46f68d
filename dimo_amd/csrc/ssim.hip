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
// 168 VGPRs: THREE workgroups per CU, persistent (grid = 3 x 256 CUs).  At 128 VGPRs (four per CU) the vertical pass
// spilled 61 registers: 83 us against 51; one tile per workgroup instead of the persistent loop: 59.
constexpr int SSIM_WGS_PER_CU = 3, SSIM_GRID = SSIM_WGS_PER_CU * 256;
// Optional per-image base pointers of img2 (the targets of a batch live in a resident pool, one tensor per image:
// stacking them cost two copy kernels per motion at the head of every step).  n == 0: img2 is one contiguous tensor.
constexpr int SSIM_MAX_IMAGES = 32;
struct ImagePtrs {
  int n, channels;
  const float *p[SSIM_MAX_IMAGES];
};
// Optional phase trace of the fused kernel (tools/ssim_bisect.hip, -DDIMO_SSIM_TRACE): wave 0 of a workgroup stamps
// the 100 MHz clock at the phase boundaries of its first tile.
#ifdef DIMO_SSIM_TRACE
__device__ unsigned long long *g_ssim_trace = nullptr;  // [workgroups][16]
#define SSIM_MARK(k)                                                                                      \
  do {                                                                                                    \
    if (g_ssim_trace && tid == 0 && tile == (int)blockIdx.x) g_ssim_trace[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define SSIM_MARK(k)
#endif
__global__ void __launch_bounds__(256, SSIM_WGS_PER_CU) ssim_fused_kernel(int H, int W, int n_planes, int clamp1, Window win,
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
  float *const s_p = s_m;  // [3][HS][PS]
  const WindowV winv = window_to_vgprs(win);
  const int tid = threadIdx.x;
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  const int total_tiles = tiles_x * tiles_y * n_planes;
  const float scale = dL_dmean[0] * inv_numel;
  // first pass, vertical: 52 columns x 4 row groups (rows 0-10, 11-21, 22-32, 31-41 of the 42: the last overlaps)
  const bool v1 = tid < IS * 4;
  const int v1_c = tid % IS, v1_r0 = min((tid / IS) * V_ROWS, HS - V_ROWS);
  // first pass, horizontal: 42 rows x 6 groups of 8 columns (the last group: columns 40, 41 and six unused ones)
  const bool h1 = tid < HS * 6;
  const int h1_row = tid % HS, h1_c0 = (tid / HS) * 8;
  // second pass, horizontal: 42 rows x 4 groups of 8; vertical: column c, rows r0 .. r0 + 3
  const bool h2 = tid < HS * (TS / 8);
  const int h2_row = tid % HS, h2_c0 = (tid / HS) * 8;
  const int c = tid & (TS - 1), r0 = (tid >> 5) * 4;
  float m_acc = 0.0f;
  struct Tile {
    int plane, x0, y0;
    const float *p1, *p2;
  };
  auto tile_at = [&](int tile) {
    Tile t;
    t.plane = tile / (tiles_x * tiles_y);
    t.x0 = (tile % tiles_x) * TS, t.y0 = ((tile / tiles_x) % tiles_y) * TS;
    t.p1 = img1 + (size_t)t.plane * H * W;
    t.p2 = img2_images.n ? img2_images.p[t.plane / img2_images.channels] +
                               (size_t)(t.plane % img2_images.channels) * H * W
                         : img2 + (size_t)t.plane * H * W;
    return t;
  };
  float xs[2 * V_ROWS - 1], ys[2 * V_ROWS - 1];  // 21 input rows of this thread's column
  // Requests the inputs of a tile: address = scalar row base + one 32-bit lane offset for all 21 rows.  Tiles whose
  // 52 x 52 halo lies inside the image (three in four at 512^2) load unconditionally; the others mask the lanes
  // outside off, and nothing inside the masked region may USE a loaded value (a use makes the loads wait for one
  // another: 5 us for the 42 -- the clamp of img1 is applied when the vertical pass reads the registers).
  auto request = [&](const Tile &t) {
    if (!v1) return;
    const int gx = t.x0 + v1_c - 2 * SR, gy0 = t.y0 + v1_r0 - 2 * SR;
    const unsigned off = 4u * (unsigned)(v1_r0 * W + gx);
    const bool interior = t.x0 >= 2 * SR && t.x0 + TS + 2 * SR <= W && t.y0 >= 2 * SR && t.y0 + TS + 2 * SR <= H;
    if (interior) {
#pragma unroll
      for (int i = 0; i < 2 * V_ROWS - 1; ++i) {
        const ptrdiff_t row = (ptrdiff_t)(t.y0 + i - 2 * SR) * W;
        xs[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t.p1 + row) + off);
        ys[i] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t.p2 + row) + off);
      }
    } else {
      const bool in_x = gx >= 0 && gx < W;
#pragma unroll
      for (int i = 0; i < 2 * V_ROWS - 1; ++i) {
        const ptrdiff_t row = (ptrdiff_t)(t.y0 + i - 2 * SR) * W;
        float a = 0.0f, b = 0.0f;
        if (in_x && gy0 + i >= 0 && gy0 + i < H) {
          a = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t.p1 + row) + off);
          b = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(t.p2 + row) + off);
        }
        xs[i] = a, ys[i] = b;
      }
    }
  };
  // The workgroups are persistent (three per CU).  Requesting the NEXT tile's inputs early (after the vertical pass,
  // or before the second window pass) was slower: 42 more live registers, and the requests' issue slots are the cost,
  // not their latency (55-65 us against 51).  Barriers order LDS traffic only (lds_barrier).
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const Tile cur = tile_at(tile);
    const int plane = cur.plane, x0 = cur.x0, y0 = cur.y0;
    const float *p1 = cur.p1, *p2 = cur.p2;
    SSIM_MARK(0);
    request(cur);
    SSIM_MARK(1);
    lds_barrier();  // the previous tile's LDS has been consumed
    SSIM_MARK(2);
    if (v1) {
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
        for (int j = 0; j < V_ROWS; ++j) s_m[q * MAP_WORDS + (v1_r0 + j) * MS + v1_c] = o[j];
      }
    }
    SSIM_MARK(3);
    lds_barrier();
    SSIM_MARK(4);
    float d[3][8];  // derivative planes of this thread's 8 halo pixels
    if (h1) {
      float st[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float in[20];
        const float4 *row = reinterpret_cast<const float4 *>(&s_m[q * MAP_WORDS + h1_row * MS + h1_c0]);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float4 t = row[i];
          in[4 * i] = t.x, in[4 * i + 1] = t.y, in[4 * i + 2] = t.z, in[4 * i + 3] = t.w;
        }
        taps<8, 20>(winv, in, st[q]);
      }
      const int gy = y0 + h1_row - SR;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int hx = h1_c0 + i, gx = x0 + hx - SR;
        const bool in = hx < HS && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const SsimPoint sp = ssim_point(st[0][i], st[1][i], st[2][i], st[3][i]);
        const bool own = h1_row >= SR && h1_row < SR + TS && hx >= SR && hx < SR + TS;  // this tile's 32 x 32 outputs
        if (in && own) m_acc += sp.m;
        d[0][i] = in ? sp.d_mu1 : 0.0f, d[1][i] = in ? sp.d_e11 : 0.0f, d[2][i] = in ? sp.d_e12 : 0.0f;
      }
    }
    SSIM_MARK(5);
    lds_barrier();  // every map has been read: the derivative planes may overwrite them
    SSIM_MARK(6);
    if (h1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float4 *row = reinterpret_cast<float4 *>(&s_p[(k * HS + h1_row) * PS + h1_c0]);
        row[0] = make_float4(d[k][0], d[k][1], d[k][2], d[k][3]);
        row[1] = make_float4(d[k][4], d[k][5], d[k][6], d[k][7]);
      }
    }
    // x and y of this thread's four output pixels (used after the second pass), then the next tile's inputs: the
    // statistics' registers are free now (requesting right after the vertical pass spilled: 48 -> 65 us)
    const int gx = x0 + c;
    float xo[4], yo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i;
      const int o = (gx < W && gy < H) ? gy * W + gx : 0;
      xo[i] = p1[o], yo[i] = p2[o];
    }
    float g[3][4];
    SSIM_MARK(7);
#pragma unroll
    for (int q = 0; q < 3; ++q) {  // (sharing the barriers between the planes, with a buffer each, changed nothing)
      lds_barrier();  // s_p complete (q = 0) / s_h free again
      if (h2) {
        float in[20], o[8];
        const float4 *row = reinterpret_cast<const float4 *>(&s_p[(q * HS + h2_row) * PS + h2_c0]);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float4 t = row[i];
          in[4 * i] = t.x, in[4 * i + 1] = t.y, in[4 * i + 2] = t.z, in[4 * i + 3] = t.w;
        }
        taps<8, 20>(winv, in, o);
#pragma unroll
        for (int i = 0; i < 8; ++i) s_h[h2_row][h2_c0 + i] = o[i];
      }
      lds_barrier();
      float col[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) col[i] = s_h[r0 + i][c];
      taps<4, 14>(winv, col, g[q]);
    }
    SSIM_MARK(8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = y0 + r0 + i;
      if (gx < W && gy < H) {
        const size_t o = (size_t)gy * W + gx;
        dL_dimg1[(size_t)plane * H * W + o] = (g[0][i] + 2.0f * maybe_clamp(xo[i], clamp1) * g[1][i] + yo[i] * g[2][i]) * scale;
      }
    }
    SSIM_MARK(9);
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
  const dim3 grid((unsigned)(tiles < SSIM_GRID ? tiles : SSIM_GRID)), block(256);
  ScopedTimer tm(T_SSIM_FWD, stream);
  hipLaunchKernelGGL(ssim_fused_kernel, grid, block, 0, stream, H, W, (int)planes, clamp_img1, win, img1, img2, ptrs,
                     dL_dmean, 1.0f / (float)((double)planes * H * W), ssim_sum, dL_dimg1);
  return check_launch();
}
