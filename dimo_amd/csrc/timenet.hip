// TimeNet (renderer/latent_gs_renderer.py:184-245) forward and backward for a whole training step's batch of
// (motion, frame) pairs as a short chain of fp32 MFMA GEMMs -- rows = pairs x control points (2048 at the
// benchmark configuration), widths 104 / 256 / 360.  PyTorch's eager version of the same MLP is ~190 launches of
// 2-20 us kernels per step (1.4 ms of a 4.7 ms step).  Two implementations:
//   * the reference's shape class (width 256, <= 1 skip, embedding <= 128 columns): the whole forward is ONE launch
//     and the whole dgrad chain ONE launch (timenet_fwd_fused_kernel / timenet_bwd_fused_kernel below: a workgroup
//     walks 16 rows through every layer, activations in LDS, packed weights streamed through a register ring),
//     followed by the grouped wgrad and the embedding backward;
//   * any other shape: one GEMM launch per layer (forward D + 3 launches, backward D + 5):
//
//   forward : embed (NeRF positional encoding of control points and time + latent, src/pos_enc.py:6-54)
//             -> D x [Y = relu(X W^T + b)]   (the skip layer writes beside the embedding: the concat is free)
//             -> both head hidden layers in one launch -> head outputs (3 + 4 columns, wave dot products)
//   backward: head outputs -> one two-segment dgrad for both heads -> D dgrads with the ReLU mask fused in the
//             epilogue -> ONE grouped split-K wgrad launch for every weight and bias (HW fp32 atomics straight into
//             the flat gradient bucket) -> embedding backward (control points, latent rows).
//
// GEMM core: v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain), 32x64 block tile, 4 waves of 16x32, K staged
// through LDS 64 deep in [k][m] order (operand reads are one ds_read_b32 per lane), register prefetch of the next K
// stage.  The f32 MFMA issues at the vector rate, so the floor of a 2048x256x256 layer is 1.7 us of MFMA issue
// with every SIMD of the chip busy; HBM and LDS traffic are far from binding, global-load latency is what remains.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "../../include/dimo_hip.h"

using namespace dimo;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 32, BN = 64, BK = 64, LDT = 65;
constexpr int MAX_LAYERS = DIMO_TIMENET_MAX_LAYERS;
constexpr int MAX_PAIRS = DIMO_TIMENET_MAX_PAIRS;

// ---- operand staging -------------------------------------------------------------------------------------------
// A K stage is a ROWS x 64 block of an operand (ROWS = 32 output rows for A, 64 output columns for B), kept in LDS
// as S[k][m] (LDT = 65: the transposing store of a k-contiguous source and the row store of an m-contiguous one
// spread over the banks; the MFMA operand read S[k + lane/16][m0 + lane%16] is one ds_read_b32).  KC (k
// contiguous): element (r, k) at src[r * ld + k]; otherwise (r contiguous) at src[k * ld + r].
// Deep stages on purpose: with one or two workgroups per CU nothing hides a global load except the MFMAs of the
// stage in flight -- with 16-deep stages the chain was load-latency bound.
// VEC: 16-byte loads along the contiguous dimension (pointer 16-B aligned, ld and the contiguous extent multiples
// of 4) -- a quarter of the load instructions.
// Loads are UNCONDITIONAL from clamped addresses and the out-of-range lanes are zeroed when the stage is stored
// to LDS (`ok` bit per load): a branch around each load makes the compiler drain vmcnt at every join, which
// serialises the prefetch with the MFMAs it is supposed to hide under.
template <int ROWS, bool KC, bool VEC>
__device__ __forceinline__ uint32_t load_stage(const float *__restrict__ src, int ld, int rows, int r0, int k0,
                                               int kend, float (&v)[ROWS / 4]) {
  constexpr int PER = ROWS / 4;  // elements per thread
  uint32_t ok = 0;
  if (VEC) {
    constexpr int NQ = KC ? 16 : ROWS / 4;  // quads along the contiguous dimension
    constexpr int STEP = 256 / NQ;          // threads stacked along the other one
    const int q = (threadIdx.x % NQ) * 4, h = threadIdx.x / NQ;
#pragma unroll
    for (int e = 0; e < PER / 4; ++e) {
      const int r = r0 + (KC ? h + STEP * e : q), k = k0 + (KC ? q : h + STEP * e);
      ok |= (uint32_t)(r < rows && k < kend) << e;
      const int rc = min(r, rows - (KC ? 1 : 4)), kc = min(k, kend - (KC ? 4 : 1));
      const float4 x = *reinterpret_cast<const float4 *>(src + (KC ? (size_t)rc * ld + kc : (size_t)kc * ld + rc));
      v[4 * e] = x.x, v[4 * e + 1] = x.y, v[4 * e + 2] = x.z, v[4 * e + 3] = x.w;
    }
    return ok;
  }
  constexpr int NL = KC ? 64 : ROWS;  // lanes along the contiguous dimension
  constexpr int STEP = 256 / NL;
  const int lo = threadIdx.x % NL, hi = threadIdx.x / NL;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int r = r0 + (KC ? hi + STEP * e : lo), k = k0 + (KC ? lo : hi + STEP * e);
    ok |= (uint32_t)(r < rows && k < kend) << e;
    const int rc = min(r, rows - 1), kc = min(k, kend - 1);
    v[e] = src[KC ? (size_t)rc * ld + kc : (size_t)kc * ld + rc];
  }
  return ok;
}
template <int ROWS, bool KC, bool VEC>
__device__ __forceinline__ void store_stage(float (*S)[LDT], uint32_t ok, const float (&v)[ROWS / 4]) {
  constexpr int PER = ROWS / 4;
  if (VEC) {
    constexpr int NQ = KC ? 16 : ROWS / 4;
    constexpr int STEP = 256 / NQ;
    const int q = (threadIdx.x % NQ) * 4, h = threadIdx.x / NQ;
#pragma unroll
    for (int e = 0; e < PER / 4; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = ((ok >> e) & 1u) ? v[4 * e + j] : 0.f;
        if (KC)
          S[q + j][h + STEP * e] = x;
        else
          S[h + STEP * e][q + j] = x;
      }
    return;
  }
  constexpr int NL = KC ? 64 : ROWS;
  constexpr int STEP = 256 / NL;
  const int lo = threadIdx.x % NL, hi = threadIdx.x / NL;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const float x = ((ok >> e) & 1u) ? v[e] : 0.f;
    if (KC)
      S[lo][hi + STEP * e] = x;
    else
      S[hi + STEP * e][lo] = x;
  }
}

// Workgroup tile 32 x 64, wave tile 16 x 32 = two v_mfma_f32_16x16x4_f32 accumulators: 2048 x 256 outputs are 256
// workgroups x 4 waves = one wave per SIMD of the whole chip (64 x 64 tiles of 32x32x2 MFMAs filled half of it and
// took twice as long per layer).
struct Acc {
  f32x4 c0, c1;  // columns wn .. wn+15 and wn+16 .. wn+31; lane l, reg r: row 4 (l / 16) + r, column l % 16
};

// acc += A[m0.., kbeg..kend) * B[kbeg..kend), n0..); bias_sum += this thread's share of the column sums of A
// (column threadIdx % 32, k rows (threadIdx / 32) * 8 .. + 8 of every stage)
template <bool A_KC, bool B_KC, bool BIAS_SUM, bool VEC>
__device__ __forceinline__ void gemm_segment(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                             int ldb, int M, int N, int m0, int n0, int kbeg, int kend,
                                             float (*As)[LDT], float (*Bs)[LDT], Acc &acc, float &bias_sum) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 16, wn = (wave & 1) * 32;
  const int nk = (kend - kbeg + BK - 1) / BK;
  float ra[BM / 4], rb[BN / 4];
  uint32_t oka = load_stage<BM, A_KC, VEC>(A, lda, M, m0, kbeg, kend, ra);
  uint32_t okb = load_stage<BN, B_KC, VEC>(B, ldb, N, n0, kbeg, kend, rb);
  for (int it = 0; it < nk; ++it) {
    __syncthreads();  // the previous stage (or segment) has been consumed
    store_stage<BM, A_KC, VEC>(As, oka, ra);
    store_stage<BN, B_KC, VEC>(Bs, okb, rb);
    __syncthreads();
    if (it + 1 < nk) {  // in flight under this stage's MFMAs
      oka = load_stage<BM, A_KC, VEC>(A, lda, M, m0, kbeg + (it + 1) * BK, kend, ra);
      okb = load_stage<BN, B_KC, VEC>(B, ldb, N, n0, kbeg + (it + 1) * BK, kend, rb);
    }
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 4) {
      const float a = As[kk + (lane >> 4)][wm + (lane & 15)];
      const float b0 = Bs[kk + (lane >> 4)][wn + (lane & 15)];
      const float b1 = Bs[kk + (lane >> 4)][wn + 16 + (lane & 15)];
      acc.c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc.c0, 0, 0, 0);
      acc.c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc.c1, 0, 0, 0);
    }
    if (BIAS_SUM) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) bias_sum += As[(t >> 5) * 8 + kk][t & 31];
    }
  }
}

// ---- dense layer / dgrad ---------------------------------------------------------------------------------------
struct GemmArgs {
  const float *A[2], *B[2];  // up to two K segments accumulated into the same tile
  int lda[2], ldb[2], K[2], nseg;
  float *C;
  int ldc, M, N;
  const float *bias;              // [N] or null
  const float *mask;              // C *= (mask > 0) for columns >= mask_from_col (ReLU backward), or null
  int ldmask, mask_from_col, relu, accumulate;
  // blockIdx.z == 1 (second problem sharing A): the two head hidden layers
  const float *B_alt, *bias_alt;
  float *C_alt;
};

template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ float As[BK][LDT], Bs[BK][LDT];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool alt = blockIdx.z == 1;
  Acc acc;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc.c0[i] = 0.f, acc.c1[i] = 0.f;
  float unused = 0.f;
  for (int s = 0; s < g.nseg; ++s)
    gemm_segment<A_KC, B_KC, false, VEC>(g.A[s], g.lda[s], (alt && s == 0) ? g.B_alt : g.B[s], g.ldb[s], g.M, g.N,
                                         m0, n0, 0, g.K[s], As, Bs, acc, unused);
  const float *bias = alt ? g.bias_alt : g.bias;
  float *C = alt ? g.C_alt : g.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mb = m0 + (wave >> 1) * 16 + 4 * (lane >> 4);
  // gather the ReLU mask / the old C values first (unconditional loads from clamped positions): a load inside the
  // per-element branch made the compiler wait for each one in turn -- serial L2 round trips that doubled the time
  // of the dgrad launches
  float mk[8], old[8], bv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + (wave & 1) * 32 + 16 * h + (lane & 15);
    const int nc = min(n, g.N - 1);
    bv[h] = bias ? bias[nc] : 0.f;
    const bool masked = g.mask && n >= g.mask_from_col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(mb + r, g.M - 1);
      mk[4 * h + r] = masked ? g.mask[(size_t)m * g.ldmask + nc] : 1.0f;
      old[4 * h + r] = g.accumulate ? C[(size_t)m * g.ldc + nc] : 0.0f;
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + (wave & 1) * 32 + 16 * h + (lane & 15);
    if (n >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mb + r;
      if (m >= g.M) continue;
      float v = (h ? acc.c1[r] : acc.c0[r]) + bv[h];
      if (g.relu) v = fmaxf(v, 0.f);
      if (!(mk[4 * h + r] > 0.f)) v = 0.f;
      C[(size_t)m * g.ldc + n] = old[4 * h + r] + v;
    }
  }
}

// ---- grouped split-K weight gradients --------------------------------------------------------------------------
// problem p: gW[M x N] += dZ^T X with dZ [rows x M], X [rows x N] (both row-major), gbias[M] += column sums of dZ
struct WgradProblem {
  const float *dZ, *X;
  float *gW, *gbias;
  int ld_dz, ld_x, ld_w, M, N, tiles_n, tile_begin, vec;
};
struct WgradArgs {
  WgradProblem p[MAX_LAYERS];
  int nprob, rows, kchunk;
  int tiles;          // wgrad64_kernel: tiles per K chunk (1-D grid)
};

// ---- embedding -------------------------------------------------------------------------------------------------
struct PairTable {
  float time[MAX_PAIRS];
  int latent_row[MAX_PAIRS];
};

// cat[row, 0:E] = [sin/cos(2^f x) f < pts_freqs | sin/cos(2^f t) f < time_freqs | latent]  (src/pos_enc.py:36-45:
// per frequency sin(all dims) then cos(all dims))
__global__ void embed_kernel(int P, int Mc, int E, int ld, int pts_freqs, int time_freqs, int latent_dim,
                             const float *__restrict__ c_xyz, const float *__restrict__ latent_table, PairTable pt,
                             float *__restrict__ cat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * Mc * E) return;
  const int col = idx % E, row = idx / E;
  const int p = row / Mc, m = row % Mc;
  const int npts = 6 * pts_freqs, ntime = 2 * time_freqs;
  float v;
  if (col < npts) {
    const int f = col / 6, w = col % 6;
    const float x = c_xyz[m * 3 + (w % 3)] * exp2f((float)f);
    v = w < 3 ? sinf(x) : cosf(x);
  } else if (col < npts + ntime) {
    const int c = col - npts, f = c >> 1;
    const float x = pt.time[p] * exp2f((float)f);
    v = (c & 1) ? cosf(x) : sinf(x);
  } else {
    v = latent_table[(size_t)pt.latent_row[p] * latent_dim + (col - npts - ntime)];
  }
  cat[(size_t)row * ld + col] = v;
}

// Backward of the embedding, 64 rows (pair p, control point m) per block:
//   g_c_xyz[m, d]              += sum_f 2^f (g_sin cos - g_cos sin)      (sin / cos re-read from the saved embedding)
//   g_latent_table[row(p), j]  += g_cat[(p, m), lat0 + j]
// Each thread folds its rows first; the adds across blocks / pairs are hardware fp32 atomics.
struct EmbedBwdArgs {
  int rows, Mc, ld, pts_freqs, lat0, latent_dim, nblocks;
  const float *cat, *g_cat;
  float *g_c_xyz, *g_latent_table;
  int latent_row[MAX_PAIRS];
};
__device__ __forceinline__ void embed_bwd_body(int block, int rows, int Mc, int ld, int pts_freqs, int lat0,
                                               int latent_dim, const float *__restrict__ cat,
                                               const float *__restrict__ g_cat, const int *latent_row,
                                               float *__restrict__ g_c_xyz, float *__restrict__ g_latent_table) {
  const int row0 = block * 64, t = threadIdx.x;
  if (g_c_xyz && t < 192) {
    const int row = row0 + t / 3, d = t % 3;
    if (row < rows) {
      const size_t base = (size_t)row * ld;
      float s = 0.f;
      for (int f = 0; f < pts_freqs; ++f) {
        const float sn = cat[base + 6 * f + d], cs = cat[base + 6 * f + 3 + d];
        s += exp2f((float)f) * (g_cat[base + 6 * f + d] * cs - g_cat[base + 6 * f + 3 + d] * sn);
      }
      unsafeAtomicAdd(g_c_xyz + (row % Mc) * 3 + d, s);
    }
  }
  if (!g_latent_table) return;
  for (int j0 = 0; j0 < latent_dim; j0 += 32) {
    const int j = j0 + (t & 31), first = row0 + (t >> 5) * 8;
    if (j >= latent_dim) continue;
    float s = 0.f;
    int cur = -1;
    for (int row = first; row < min(rows, first + 8); ++row) {
      const int p = row / Mc;
      if (p != cur) {
        if (cur >= 0) unsafeAtomicAdd(g_latent_table + (size_t)latent_row[cur] * latent_dim + j, s);
        cur = p, s = 0.f;
      }
      s += g_cat[(size_t)row * ld + lat0 + j];
    }
    if (cur >= 0) unsafeAtomicAdd(g_latent_table + (size_t)latent_row[cur] * latent_dim + j, s);
  }
}

// ---- the same grouped split-K problem on 64 x 64 tiles of v_mfma_f32_32x32x2_f32 ------------------------------------
// In wgrad_kernel above a 16 x 32 wave tile reads three operands per two 16x16x4 MFMAs, and with the [k][m] staging at
// leading dimension 65 the two 16-lane rows of a ds_read_b32 lane group overlap in 15 of 16 banks (bank = dword
// address mod 32 for that instruction): every operand read is a 2-way conflict (profiles/r04_sq_summary.txt: 46 % of
// the LDS cycles).  Here a wave owns a 32 x 32 tile: ONE read per operand and 64-cycle MFMA, 32 consecutive dwords per
// lane group (conflict-free at any leading dimension: 0 conflicts measured), the stage stored with ds_write_b128
// straight from the 16-byte global loads (rows of 64 floats, no padding).  Both operands are k-major in memory
// (dZ [rows x M], X [rows x N]), so the stage is a plain copy.  Two accumulators (even / odd pairs of k) keep
// consecutive MFMAs of a wave independent.  The split over K is chosen so that the launch is ONE round of workgroups
// (four per CU: 32 KB of LDS each), in chunks of whole half stages.
// What it is bound by (a bisection build of round 4, profiles/r04_timenet_wgrad64.txt): NOT the matrix pipe -- the launch's
// skeleton (first loads, LDS stores, barriers) is 14 us, the 4.1 M fp32 atomics alone add 10.8 (~1.4 elements per
// clock and L2 channel), the operand loads alone 14.1, the MFMAs alone 9.9, and the four barely overlap: one round
// of workgroups runs its phases in lock step.  44 us with the embedding backward inside, against 42 + 10.7.
constexpr int W_T = 64, W_K = 64;  // tile edge (M and N), K rows per stage
typedef float f32x16 __attribute__((ext_vector_type(16)));

// loads this thread's share of a stage: rows k0 + t / 16 + 16 e (e < 4), columns c0 + 4 (t % 16) .. + 3 of a k-major
// operand (element (k, c) at src[k * ld + c]).  Unconditional loads from clamped addresses; bit 4 e + j of `ok` says
// whether element j of quad e is in range, and the others are zeroed when the stage is stored (behind the barrier: a
// select next to the load becomes a branch around it, and every such load then waits for its own round trip).
__device__ __forceinline__ uint32_t wload(const float *__restrict__ src, int ld, int cols, int c0, int k0, int kend,
                                          bool vec, float4 (&v)[4]) {
  const int t = threadIdx.x, c = c0 + 4 * (t & 15), h = t >> 4;
  uint32_t ok = 0;
  if (vec) {  // (workgroup-uniform) cols % 4 == 0: a quad is inside or outside as a whole
    const int cc = min(c, cols - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + h + 16 * e;
      ok |= (c < cols && k < kend ? 0xfu : 0u) << (4 * e);
      v[e] = *reinterpret_cast<const float4 *>(src + (size_t)min(k, kend - 1) * ld + cc);
    }
  } else {  // element by element (the head outputs: 3 and 4 columns)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + h + 16 * e;
      const float *row = src + (size_t)min(k, kend - 1) * ld;
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = row[min(c + j, cols - 1)];
        ok |= (uint32_t)(c + j < cols && k < kend) << (4 * e + j);
      }
      v[e] = make_float4(x[0], x[1], x[2], x[3]);
    }
  }
  return ok;
}
__device__ __forceinline__ void wstore(float (*S)[W_T], uint32_t ok, const float4 (&v)[4]) {
  const int t = threadIdx.x, q = 4 * (t & 15), h = t >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t o = ok >> (4 * e);
    *reinterpret_cast<float4 *>(&S[h + 16 * e][q]) = make_float4((o & 1u) ? v[e].x : 0.f, (o & 2u) ? v[e].y : 0.f,
                                                                 (o & 4u) ? v[e].z : 0.f, (o & 8u) ? v[e].w : 0.f);
  }
}

// The embedding backward rides in the same launch (its 64-row blocks are the FIRST workgroups: 32 of them at the
// benchmark size, ten dependent rounds of loads each -- 10 us as a launch of its own behind this one).
__global__ __launch_bounds__(256, 4) void wgrad64_kernel(WgradArgs g, EmbedBwdArgs e) {
  __shared__ __attribute__((aligned(16))) float As[W_K][W_T], Bs[W_K][W_T];
  const int v = (int)blockIdx.x;
  if (v < e.nblocks) {
    embed_bwd_body(v, e.rows, e.Mc, e.ld, e.pts_freqs, e.lat0, e.latent_dim, e.cat, e.g_cat, e.latent_row, e.g_c_xyz,
                   e.g_latent_table);
    return;
  }
  const int bx = (v - e.nblocks) % g.tiles, by = (v - e.nblocks) / g.tiles;
  int pi = 0;
  while (pi + 1 < g.nprob && bx >= g.p[pi + 1].tile_begin) ++pi;
  const WgradProblem &p = g.p[pi];
  const int tile = bx - p.tile_begin;
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * W_T, n0 = tn * W_T;
  const int kbeg = by * g.kchunk, kend = min(g.rows, kbeg + g.kchunk);
  if (kbeg >= kend) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const bool vec_a = (p.vec & 1) != 0, vec_b = (p.vec & 2) != 0;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
  float bsum = 0.f;
  const bool bias = tn == 0 && p.gbias;
  const int nk = (kend - kbeg + W_K - 1) / W_K;
  float4 ra[4], rb[4];
  uint32_t oka = wload(p.dZ, p.ld_dz, p.M, m0, kbeg, kend, vec_a, ra);
  uint32_t okb = wload(p.X, p.ld_x, p.N, n0, kbeg, kend, vec_b, rb);
  for (int it = 0; it < nk; ++it) {
    __syncthreads();  // the previous stage has been consumed
    wstore(As, oka, ra);
    wstore(Bs, okb, rb);
    __syncthreads();
    if (it + 1 < nk) {  // in flight under this stage's MFMAs
      oka = wload(p.dZ, p.ld_dz, p.M, m0, kbeg + (it + 1) * W_K, kend, vec_a, ra);
      okb = wload(p.X, p.ld_x, p.N, n0, kbeg + (it + 1) * W_K, kend, vec_b, rb);
    }
    // operand mapping of 32x32x2: A[m = lane % 32][k = lane / 32], B[k = lane / 32][n = lane % 32]; the operands of
    // the next two MFMAs are read while the current two run (two register sets, each half unrolled in full).  The
    // second half of a stage is skipped when the chunk ends in the first (chunks are whole HALF stages).
    const float *pa = &As[lane >> 5][wm + (lane & 31)], *pb = &Bs[lane >> 5][wn + (lane & 31)];
    const int valid = kend - (kbeg + it * W_K);  // k rows of this stage (the others are zero)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && valid <= W_K / 2) break;
      const int k0 = half * (W_K / 2);
      float a[2][2], b[2][2];
      a[0][0] = pa[k0 * W_T], b[0][0] = pb[k0 * W_T], a[0][1] = pa[(k0 + 2) * W_T], b[0][1] = pb[(k0 + 2) * W_T];
#pragma unroll
      for (int kk = k0; kk < k0 + W_K / 2; kk += 4) {
        const int cur = (kk >> 2) & 1;
        if (kk + 4 < k0 + W_K / 2) {
          a[cur ^ 1][0] = pa[(kk + 4) * W_T], b[cur ^ 1][0] = pb[(kk + 4) * W_T];
          a[cur ^ 1][1] = pa[(kk + 6) * W_T], b[cur ^ 1][1] = pb[(kk + 6) * W_T];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise folds the two sets into one and waits per pair)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][0], b[cur][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][1], b[cur][1], acc1, 0, 0, 0);
      }
    }
    if (bias) {  // column t % 64, k rows 16 (t / 64) .. + 15 of the stage (rows past kend are zero)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) bsum += As[(t >> 6) * 16 + kk][t & 63];
    }
  }
  if (bias && m0 + (t & 63) < p.M) unsafeAtomicAdd(p.gbias + m0 + (t & 63), bsum);
  // accumulator layout of 32x32: lane l, register r: row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
  const int n = n0 + wn + (lane & 31);
  const int mb = m0 + wm + 4 * (lane >> 5);
  if (n >= p.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mb + (r & 3) + 8 * (r >> 2);
    if (m < p.M) unsafeAtomicAdd(p.gW + (size_t)m * p.ld_w + n, acc0[r] + acc1[r]);
  }
}

// ---- head output layers (W -> 3 and W -> 4): one wave per row ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void head_out_kernel(int rows, int Wd, const float *__restrict__ hp,
                                                       const float *__restrict__ hr, const float *__restrict__ Wp,
                                                       const float *__restrict__ bp, const float *__restrict__ Wr,
                                                       const float *__restrict__ br, float *__restrict__ d_xyz,
                                                       float *__restrict__ d_rot) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float a[3] = {0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane; c < Wd; c += 64) {
    const float x = hp[(size_t)row * Wd + c], y = hr[(size_t)row * Wd + c];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = fmaf(x, Wp[i * Wd + c], a[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fmaf(y, Wr[i * Wd + c], q[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) a[i] = wave_sum(a[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = wave_sum(q[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d_xyz[(size_t)row * 3 + i] = a[i] + bp[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) d_rot[(size_t)row * 4 + i] = q[i] + br[i];
  }
}

// dZp = (g_d_xyz Wp) * (hp > 0), dZr = (g_d_rot Wr) * (hr > 0)
__global__ void head_out_bwd_kernel(int rows, int Wd, const float *__restrict__ g_xyz, const float *__restrict__ g_rot,
                                    const float *__restrict__ hp, const float *__restrict__ hr,
                                    const float *__restrict__ Wp, const float *__restrict__ Wr,
                                    float *__restrict__ dzp, float *__restrict__ dzr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * Wd) return;
  const int row = idx / Wd, c = idx % Wd;
  float vp = 0.f, vr = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) vp = fmaf(g_xyz[row * 3 + i], Wp[i * Wd + c], vp);
#pragma unroll
  for (int i = 0; i < 4; ++i) vr = fmaf(g_rot[row * 4 + i], Wr[i * Wd + c], vr);
  dzp[idx] = hp[idx] > 0.f ? vp : 0.f;
  dzr[idx] = hr[idx] > 0.f ? vr : 0.f;
}

// ---- the whole forward in ONE launch (W = 256 nets) -------------------------------------------------------------
// Activation-stationary: a workgroup (8 waves) owns 16 rows and walks them through every layer; the activations stay
// in LDS (written to the workspace as well, for the backward), only the weights stream in, straight from L2 into the
// MFMA B operand.  128 workgroups x 8 waves = one wave per SIMD of the chip for the 2048-row step batch, and the
// chain of D + 3 dependent launches (each a global-load -> LDS -> barrier -> MFMA pipeline that starts cold) becomes
// one.
//
// Operand mapping of v_mfma_f32_16x16x4_f32 (A[m = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16]): for a
// block of 16 k values, lane (kq, m) fetches ONE float4 X[m][k0 + 4 kq .. + 3] from LDS and lane (kq, n) ONE float4
// W[n][k0 + 4 kq .. + 3] from global memory; component i of both feeds MFMA i of the block, which therefore contracts
// k0 + {i, 4 + i, 8 + i, 12 + i} -- a permutation of the block's k values, i.e. the same sum.
constexpr int FR = 16;           // rows per workgroup
constexpr int FW = 256;          // hidden width: 8 waves x 32 columns
constexpr int FH_LD = FW + 4;    // leading dimensions of the LDS tiles: 4 x odd floats (16-byte rows, spread banks)
constexpr int FE_MAX = 128;      // embedding columns kept in LDS (zero padded: FPF whole blocks, see embed_blocks)
constexpr int FC_LD = FE_MAX + 4;
constexpr int FPF = 8;           // weight blocks in flight per wave; every layer's block count is a multiple of it,
                                 // so the register ring lines up across layer boundaries
__host__ __device__ constexpr int embed_blocks(int) { return FE_MAX / 16; }  // one whole chunk (fused_chunk)
static_assert((FE_MAX / 16) % FPF == 0, "the embedding chunk is a whole number of rings");
static_assert((FW / 16) % FPF == 0, "hidden blocks per layer must be a multiple of the ring depth");

struct FusedArgs {
  int R, Mc, E, CAT, D, skip, pts_freqs, time_freqs, latent_dim;
  const float *c_xyz, *latent_table;
  const float *W[MAX_LAYERS], *b[MAX_LAYERS];
  const float *Wp[MAX_LAYERS];  // packed copies of the hidden layers' weights (pack_weights_kernel)
  float *cat, *act[MAX_LAYERS], *hp, *hr, *d_xyz, *d_rot;
  int act_ld[MAX_LAYERS];
  PairTable pt;
};

// Packed weights: the B operand of one (16 columns x 16 k) MFMA block is 64 lanes x float4; stored in that order
// (block (nt, kb) at ((nt * nkb + kb) * 64 + lane) float4s) a wave's load is ONE contiguous KiB.  Straight from the
// row-major matrix the same load touches 16 rows 1 KiB (or 1.4 KiB) apart -- a handful of L2 channels for the
// whole chip, which reads the same weights at the same time; the layers then ran at the speed of those channels
// (9 us per 256 x 256 layer, fused or not).  Segments ([embedding | hidden] of the skip layer) are padded to whole
// 16-k blocks with zeros.  Repacked every step (the weights have just been updated): 0.7 M floats.
struct PackJob {
  const float *W;   // [256][ldw]
  float *out;
  int ldw, E_seg, H_seg, nkb;  // columns of the embedding / hidden segment, k blocks per column tile
};
struct PackArgs {
  PackJob job[MAX_LAYERS];
  int njobs;
};
__global__ __launch_bounds__(256) void pack_weights_kernel(PackArgs a) {
  const PackJob &j = a.job[blockIdx.y];
  const int nE = j.E_seg ? embed_blocks(j.E_seg) : 0;
  const int total = (FW / 16) * j.nkb * 64;  // float4s
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, blk = i >> 6;
    const int kb = blk % j.nkb, nt = blk / j.nkb;
    const int kq = lane >> 4, mn = lane & 15;
    // column of this float4 inside its segment, and the segment's first column / length
    const bool emb = kb < nE;
    const int kseg = 16 * (emb ? kb : kb - nE) + 4 * kq;
    const int seg0 = emb ? 0 : j.E_seg, seglen = emb ? j.E_seg : j.H_seg;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kseg + 3 < seglen) v = *reinterpret_cast<const float4 *>(j.W + (size_t)(nt * 16 + mn) * j.ldw + seg0 + kseg);
    reinterpret_cast<float4 *>(j.out)[i] = v;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL address spaces:
// on gfx9 that is s_waitcnt vmcnt(0) as well, i.e. every layer would wait for its activation stores to reach L2 and
// for the NEXT layer's prefetched weights before any wave may go on.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The layers of the fused forward as one stream of weight blocks per wave: the FPF-deep register ring of B operands
// runs ACROSS layer boundaries (weights do not depend on activations), so a layer never starts with a cold memory
// pipeline.
struct WaveWeights {
  const float4 *w0, *w1;  // this lane's float4 of block 0 of the wave's two column tiles
  int nE, nkb;            // embedding blocks (A from the embedding tile), blocks in all
};
// NT = column tiles (16 columns each) per wave: 2 with 8 waves per workgroup, 1 with 16
template <int NT = 2>
__device__ __forceinline__ WaveWeights wave_weights(const float *packed, int E, bool use_embed, bool use_hidden,
                                                    int wave, int lane) {
  WaveWeights w;
  w.nE = use_embed ? embed_blocks(E) : 0;
  w.nkb = w.nE + (use_hidden ? FW / 16 : 0);
  w.w0 = reinterpret_cast<const float4 *>(packed) + (size_t)(NT * wave) * w.nkb * 64 + lane;
  w.w1 = NT == 2 ? w.w0 + (size_t)w.nkb * 64 : w.w0;
  return w;
}
template <int NT = 2>
__device__ __forceinline__ void prefetch_blocks(const WaveWeights &w, float4 (&q0)[FPF], float4 (&q1)[FPF]) {
#pragma unroll
  for (int u = 0; u < FPF; ++u) {
    const int blk = min(u, w.nkb - 1);
    q0[u] = w.w0[blk * 64];
    if (NT == 2) q1[u] = w.w1[blk * 64];
  }
}

// One CHUNK of LEN weight blocks (LEN a multiple of FPF, fully unrolled so that every register of the ring has a
// static index and the compiler counts the loads in flight exactly -- with a runtime block loop it re-synchronised
// the whole ring, s_waitcnt vmcnt(0), at every trip): acc += X[16 x 16 LEN] * W^T.  Ring slots 0 .. FPF-1 hold
// blocks 0 .. FPF-1 of this chunk on entry; consumed slots are refilled with the rest of the chunk and then with the
// first FPF blocks of the NEXT chunk of the stream (n0 / n1), so they hold those on exit.
// NT / NTN: column tiles per wave of this chunk / of the next one (2: the wave's 32 columns; 1: a 16-column tile of
// the 128 embedding columns in the backward -- q1 / c1 are then left alone).
template <int LEN, int NT = 2, int NTN = 2>
__device__ __forceinline__ void fused_chunk(const float *xa /* this lane's A pointer at the chunk's first column */,
                                            const float4 *__restrict__ w0, const float4 *__restrict__ w1,
                                            const float4 *__restrict__ n0, const float4 *__restrict__ n1,
                                            float4 (&q0)[FPF], float4 (&q1)[FPF], f32x4 &c0, f32x4 &c1) {
  static_assert(LEN % FPF == 0, "chunks are whole rings");
  float4 a_next = *reinterpret_cast<const float4 *>(xa);
#pragma unroll
  for (int blk = 0; blk < LEN; ++blk) {
    const int u = blk % FPF;
    const float4 a = a_next;
    if (blk + 1 < LEN) a_next = *reinterpret_cast<const float4 *>(xa + 16 * (blk + 1));
    // A wave that is ahead yields: the SIMD's arbiter otherwise serves its oldest wave first, wave 0 of a workgroup
    // is through a layer's 16 blocks after 2.9 us and wave 15 after 5.2 (profiles/r03_timenet_notes.txt), and the
    // layer ends with one wave per SIMD chaining dependent MFMAs behind a thin weight stream.  The priority falls in
    // four steps over the chunk and is re-stated at every block (stated only where it changes: half the gain):
    // forward 75.3 -> 69.5 us, backward group 121.9 -> 115.3 us on one box (tools/timenet_probe.py).
    {
      const int q4 = (4 * blk) / LEN;  // (constant after unrolling)
      if (q4 == 0) __builtin_amdgcn_s_setprio(3);
      else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
      else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    const float4 p0 = q0[u], p1 = q1[u];
    if (blk + FPF < LEN) {
      q0[u] = w0[(blk + FPF) * 64];
      if (NT == 2) q1[u] = w1[(blk + FPF) * 64];
    } else {
      q0[u] = n0[(blk + FPF - LEN) * 64];
      if (NTN == 2) q1[u] = n1[(blk + FPF - LEN) * 64];
    }
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, p0.x, c0, 0, 0, 0);
    if (NT == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, p1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, p0.y, c0, 0, 0, 0);
    if (NT == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, p1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, p0.z, c0, 0, 0, 0);
    if (NT == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, p1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, p0.w, c0, 0, 0, 0);
    if (NT == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, p1.w, c1, 0, 0, 0);
  }
}

// acc = [embed | hidden] (LDS tiles, zero padded to whole chunks) * W^T for the wave's 32 columns: an embedding chunk
// (layer 0 and the layer after the skip) and / or a hidden chunk; the ring runs on into the next layer's weights (wn)
template <int NT = 2>
__device__ __forceinline__ void fused_matmul(const float *s_c, const float *s_in, const WaveWeights &w,
                                             const WaveWeights &wn, float4 (&q0)[FPF], float4 (&q1)[FPF], f32x4 &c0,
                                             f32x4 &c1, int lane) {
  const int kq = lane >> 4, mn = lane & 15;
  constexpr int EB = FE_MAX / 16, HB = FW / 16;
  const bool hidden = w.nkb > w.nE;
  if (w.nE) {
    const float4 *n0 = hidden ? w.w0 + EB * 64 : wn.w0, *n1 = hidden ? w.w1 + EB * 64 : wn.w1;
    fused_chunk<EB, NT, NT>(s_c + mn * FC_LD + 4 * kq, w.w0, w.w1, n0, n1, q0, q1, c0, c1);
  }
  if (hidden)
    fused_chunk<HB, NT, NT>(s_in + mn * FH_LD + 4 * kq, w.w0 + w.nE * 64, w.w1 + w.nE * 64, wn.w0, wn.w1, q0, q1, c0, c1);
}

// relu(acc + bias) -> LDS tile + workspace
template <int NT = 2>
__device__ __forceinline__ void fused_epilogue(const f32x4 &c0, const f32x4 &c1, float b0, float b1, float *s_out,
                                               float *__restrict__ gout, int ld_out, int row0, int R, int lane,
                                               int wave) {
  const int kq = lane >> 4, mn = lane & 15, n0 = wave * 16 * NT;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * kq + r;
    const float v0 = fmaxf(c0[r] + b0, 0.f), v1 = fmaxf(c1[r] + b1, 0.f);
    s_out[m * FH_LD + n0 + mn] = v0;
    if (NT == 2) s_out[m * FH_LD + n0 + 16 + mn] = v1;
    if (row0 + m < R) {
      gout[(size_t)(row0 + m) * ld_out + n0 + mn] = v0;
      if (NT == 2) gout[(size_t)(row0 + m) * ld_out + n0 + 16 + mn] = v1;
    }
  }
}

// NW waves per workgroup: 8 (each wave 32 columns) or 16 (16 columns: twice the weight blocks in flight per CU -- the
// kernel lives on the latency of its weight stream, 128 workgroups on 256 CUs)
template <int NW>
__global__ __launch_bounds__(64 * NW) void timenet_fwd_fused_kernel(FusedArgs g) {
  constexpr int NT = 16 / NW, NTHR = 64 * NW;
  __shared__ __attribute__((aligned(16))) float s_c[FR * FC_LD];
  __shared__ __attribute__((aligned(16))) float s_h[2][FR * FH_LD];
  __shared__ float s_wo[7 * FW];  // the 3 + 4 rows of the head output layers (read 16 x per workgroup at the very end)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * FR;
  for (int e = t; e < 7 * FW; e += NTHR) s_wo[e] = e < 3 * FW ? g.W[g.D + 1][e] : g.W[g.D + 3][e - 3 * FW];
  float4 q0[FPF], q1[FPF];
  WaveWeights w = wave_weights<NT>(g.Wp[0], g.E, true, false, wave, lane);
  prefetch_blocks<NT>(w, q0, q1);  // in flight under the embedding
  // embedding of the 16 rows (same arithmetic as embed_kernel) -> LDS (zero padded) and the workspace
  const int npts = 6 * g.pts_freqs, ntime = 2 * g.time_freqs;
  for (int e = t; e < FR * FE_MAX; e += NTHR) {
    const int m = e / FE_MAX, col = e % FE_MAX;
    const int row = min(row0 + m, g.R - 1);
    const int p = row / g.Mc, cp = row % g.Mc;
    float v = 0.f;
    if (col < npts) {
      const int f = col / 6, wd = col % 6;
      const float x = g.c_xyz[cp * 3 + (wd % 3)] * exp2f((float)f);
      v = wd < 3 ? sinf(x) : cosf(x);
    } else if (col < npts + ntime) {
      const int c = col - npts, f = c >> 1;
      const float x = g.pt.time[p] * exp2f((float)f);
      v = (c & 1) ? cosf(x) : sinf(x);
    } else if (col < g.E) {
      v = g.latent_table[(size_t)g.pt.latent_row[p] * g.latent_dim + (col - npts - ntime)];
    }
    s_c[m * FC_LD + col] = v;
    if (col < g.E && row0 + m < g.R) g.cat[(size_t)(row0 + m) * g.CAT + col] = v;
  }
  __syncthreads();
  int cur = 0;  // s_h[cur] holds the input of the next hidden layer
  // layer sequence: deformnet 0 .. D-1, then the two head hidden layers (both read h[D-1])
  for (int step = 0; step < g.D + 2; ++step) {
    const bool head = step >= g.D;
    const int li = head ? g.D + 2 * (step - g.D) : step;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    // (requested before the matmul: after it they would be a full memory round trip on the critical path)
    const float bias0 = g.b[li][wave * 16 * NT + (lane & 15)];
    const float bias1 = NT == 2 ? g.b[li][wave * 32 + 16 + (lane & 15)] : 0.f;
    WaveWeights wn = w;  // (after the last layer the ring refills with clamped repeats of its own blocks)
    if (step + 1 < g.D + 2) {
      const int ns = step + 1, nli = ns >= g.D ? g.D + 2 * (ns - g.D) : ns;
      wn = wave_weights<NT>(g.Wp[nli], g.E, ns < g.D && ns - 1 == g.skip, true, wave, lane);
    }
    fused_matmul<NT>(s_c, s_h[cur], w, wn, q0, q1, c0, c1, lane);
    w = wn;
    if (!head) {
      fused_epilogue<NT>(c0, c1, bias0, bias1, s_h[cur ^ 1], g.act[li], g.act_ld[li], row0, g.R, lane, wave);
      cur ^= 1;
      lds_barrier();
      continue;
    }
    // head hidden layer -> s_h[cur ^ 1]; its 3 / 4 output columns are wave dot products (two rows per wave, the
    // arithmetic of head_out_kernel)
    const int hd = step - g.D;
    fused_epilogue<NT>(c0, c1, bias0, bias1, s_h[cur ^ 1], hd ? g.hr : g.hp, FW, row0, g.R, lane, wave);
    lds_barrier();
    const float *Wo = s_wo + (hd ? 3 * FW : 0), *bo = g.b[li + 1];
    const int nout = hd ? 4 : 3;
    for (int rr = 0; rr < FR / NW; ++rr) {
      const int m = (FR / NW) * wave + rr;
      const float *x = s_h[cur ^ 1] + m * FH_LD;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = lane; c < FW; c += 64) {
        const float xv = x[c];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nout) a[i] = fmaf(xv, Wo[i * FW + c], a[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nout) a[i] = wave_sum(a[i]);
      if (lane == 0 && row0 + m < g.R) {
        float *dst = hd ? g.d_rot + (size_t)(row0 + m) * 4 : g.d_xyz + (size_t)(row0 + m) * 3;
        for (int i = 0; i < nout; ++i) dst[i] = a[i] + bo[i];
      }
    }
    lds_barrier();  // s_h[cur ^ 1] is overwritten by the second head
  }
}

// ---- the forward on EVERY CU: 8 rows per workgroup, v_mfma_f32_4x4x1_16b_f32 --------------------------------------
// The 16-row kernel above puts the benchmark's 2048 rows on 128 of the 256 CUs: a layer costs a workgroup 3.4 us of
// MFMA issue (16 waves x 64 MFMAs of 32 cycles on 4 SIMDs) behind a 3.1 us weight stream, 5.8 us with its barrier.
// 16x16x4 cannot take fewer than 16 rows; the multi-block form 4x4x1 can: 16 blocks of (4 rows x 1 k) x (1 k x 4
// columns), i.e. with the SAME four rows in every block ONE instruction is a 4-row x 64-column x 1-k product --
// lane l supplies A = X[l % 4][k] and B = W[column l][k] and receives, in register r, Y[row r][column l].  Measured
// (tools/mfma4_rate.hip): 10.1-10.5 cycles per instruction with >= 8 independent accumulators per SIMD, 80 % of the
// 16x16x4 rate per MAC, 43 cycles dependent latency.  Eight rows per workgroup = 256 workgroups for 2048 rows, half
// the MFMA work per CU; the weight stream per CU is what it was (every workgroup reads every weight).
//
// Work split: 16 waves = 4 column groups of 64 x 4 K-slices.  A wave holds rows 0-3 and 4-7 of its 64 columns, two
// accumulators each (even / odd k: 16 independent chains per SIMD); the K-slices are summed through LDS in the
// epilogue, which also adds the bias, applies the ReLU and writes the next layer's input tile.
// Packed weights for this kernel ("quads"): quad (cg, kq) = 64 lanes x float4 { W[64 cg + lane][seg + 4 kq .. + 3] },
// at float4 index (cg * nkq + kq) * 64 + lane -- one contiguous KiB per wave load, as before.
constexpr int R8 = 8;             // rows per workgroup
constexpr int QF = 8;             // quads in flight per wave (ring depth; 16 waves: 128 KB in flight per CU -- with 4,
                                  // the stream ran dry under every layer's epilogue: 54 us against ...)
constexpr int SUB = 2 * QF;       // quads per sub-chunk: every ring slot is refilled TWICE per loop body, so the body
                                  // ends with the ring in the registers it started in (one refill: 16 register moves
                                  // at the back edge, each waiting for its load)
constexpr int KSE = 2;            // K-slices of the embedding segment: its 32 quads are two sub-chunks, taken by the
                                  // K-slices 0 and 1 of every column group (the four K-slices of a column group share
                                  // a SIMD, so the SIMDs stay balanced)
constexpr int EQ = FE_MAX / 4;    // k-quads of the (zero padded) embedding segment: 32
constexpr int HQ = FW / 4;        // ... of a hidden segment: 64
constexpr int KS = 4;             // K-slices (waves per column group)
static_assert(EQ / KSE == SUB && HQ / KS == SUB, "a wave's embedding / hidden chunk is one sub-chunk");

__global__ __launch_bounds__(256) void pack_weights_quads_kernel(PackArgs a) {
  const PackJob &j = a.job[blockIdx.y];
  const int nE = j.E_seg ? EQ : 0, nkq = nE + (j.H_seg ? HQ : 0);
  const int total = (FW / 64) * nkq * 64;  // float4s
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, q = i >> 6;
    const int kq = q % nkq, cg = q / nkq;
    const bool emb = kq < nE;
    const int kseg = 4 * (emb ? kq : kq - nE);
    const int seg0 = emb ? 0 : j.E_seg, seglen = emb ? j.E_seg : j.H_seg;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kseg + 3 < seglen) v = *reinterpret_cast<const float4 *>(j.W + (size_t)(cg * 64 + lane) * j.ldw + seg0 + kseg);
    reinterpret_cast<float4 *>(j.out)[i] = v;
  }
}

struct Wave4 {
  const float4 *e, *h;  // this lane's float4 of the first quad of the wave's embedding / hidden chunk
  bool has_e, has_h;    // (wave-uniform: the chunks are taken or skipped by scalar branches)
};
__device__ __forceinline__ Wave4 wave4(const float *packed, bool use_embed, bool use_hidden, int cg, int ks, int lane) {
  const int nE = use_embed ? EQ : 0, nkq = nE + (use_hidden ? HQ : 0);
  const float4 *base = reinterpret_cast<const float4 *>(packed) + (size_t)cg * nkq * 64 + lane;
  Wave4 w;
  w.has_e = use_embed && ks < KSE, w.has_h = use_hidden;
  w.e = base + (size_t)((ks & (KSE - 1)) * (EQ / KSE)) * 64;
  w.h = base + (size_t)(nE + ks * (HQ / KS)) * 64;
  return w;
}
__device__ __forceinline__ const float4 *first_chunk(const Wave4 &w) { return w.has_e ? w.e : w.h; }

// LEN quads of one wave: acc += X[8 rows x 4 LEN k] * W^T for the wave's 64 columns.  xa = this lane's A pointer
// (row lane % 4 of the tile, first k of the chunk), rows 4-7 are ld4 = 4 * leading dimension floats further.  The ring
// holds the chunk's first QF quads on entry and the first QF quads of the NEXT chunk of the wave's stream (nx) on exit.
// MODE (bisection builds, DIMO_TIMENET_BISECT): 0 = the kernel; 1 = weight stream without the MFMAs (every quad is
// folded into the accumulators with four adds); 2 = MFMAs without the weight stream (the ring is never refilled)
template <int LEN, int MODE = 0>
__device__ __forceinline__ void chunk4(const float *xa, int ld4, const float4 *__restrict__ w,
                                       const float4 *__restrict__ nx, float4 (&q)[QF], f32x4 &c0e, f32x4 &c0o,
                                       f32x4 &c1e, f32x4 &c1o) {
  static_assert(LEN % QF == 0, "chunks are whole rings");
  float4 lo_n = *reinterpret_cast<const float4 *>(xa), hi_n = *reinterpret_cast<const float4 *>(xa + ld4);
#pragma unroll
  for (int b = 0; b < LEN; ++b) {
    const int u = b % QF;
    const float4 lo = lo_n, hi = hi_n;
    if (b + 1 < LEN) {
      lo_n = *reinterpret_cast<const float4 *>(xa + 4 * (b + 1));
      hi_n = *reinterpret_cast<const float4 *>(xa + ld4 + 4 * (b + 1));
    }
    {  // a wave that is ahead yields (see fused_chunk)
      const int q4 = (4 * b) / LEN;
      if (q4 == 0) __builtin_amdgcn_s_setprio(3);
      else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
      else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    const float4 p = q[u];
    if (MODE != 2) q[u] = b + QF < LEN ? w[(b + QF) * 64] : nx[(b + QF - LEN) * 64];
    if (MODE == 1) {
      c0e[0] += p.x + lo.x, c0o[0] += p.y + hi.y, c1e[0] += p.z, c1o[0] += p.w;
      continue;
    }
    c0e = __builtin_amdgcn_mfma_f32_4x4x1f32(lo.x, p.x, c0e, 0, 0, 0);
    c1e = __builtin_amdgcn_mfma_f32_4x4x1f32(hi.x, p.x, c1e, 0, 0, 0);
    c0o = __builtin_amdgcn_mfma_f32_4x4x1f32(lo.y, p.y, c0o, 0, 0, 0);
    c1o = __builtin_amdgcn_mfma_f32_4x4x1f32(hi.y, p.y, c1o, 0, 0, 0);
    c0e = __builtin_amdgcn_mfma_f32_4x4x1f32(lo.z, p.z, c0e, 0, 0, 0);
    c1e = __builtin_amdgcn_mfma_f32_4x4x1f32(hi.z, p.z, c1e, 0, 0, 0);
    c0o = __builtin_amdgcn_mfma_f32_4x4x1f32(lo.w, p.w, c0o, 0, 0, 0);
    c1o = __builtin_amdgcn_mfma_f32_4x4x1f32(hi.w, p.w, c1o, 0, 0, 0);
  }
}

constexpr int G_LD = FW + 4, GC_LD = FE_MAX + 4;  // leading dimensions of the 8-row LDS tiles

// the wave's K-slice of a layer as SUB-CHUNKS of two rings (SUB = 2 QF quads): [embedding (K-slices 0, 1)] [hidden];
// `nx` = the first sub-chunk of the wave's next layer.  One loop body for all of them: with a branch per chunk the compiler
// joins the two paths with register moves of the ring -- and a move of a register whose load is in flight waits for
// it (s_waitcnt vmcnt(0) at the end of every embedding chunk).
template <int MODE = 0>
__device__ __forceinline__ void matmul4(const float *s_c, const float *s_in, const Wave4 &w, const float4 *nx,
                                        float4 (&q)[QF], f32x4 &c0e, f32x4 &c0o, f32x4 &c1e, f32x4 &c1o, int ks,
                                        int lane) {
  const int r = lane & 3;
  const int first = w.has_e ? 0 : 1, last = w.has_h ? 2 : 1;  // sub-chunk ids: 0 = embedding, 1 = hidden
  for (int sc = first; sc < last; ++sc) {
    const float *xa = sc == 0 ? s_c + r * GC_LD + ks * (FE_MAX / KSE) : s_in + r * G_LD + ks * (FW / KS);
    const int ld4 = sc == 0 ? 4 * GC_LD : 4 * G_LD;
    const float4 *wq = sc == 0 ? w.e : w.h;
    const float4 *nq = sc + 1 < last ? w.h : nx;
    chunk4<SUB, MODE>(xa, ld4, wq, nq, q, c0e, c0o, c1e, c1o);
  }
}

// partial sums of the wave -> LDS [ks][row][column]
__device__ __forceinline__ void store_partials(float *s_part, const f32x4 &c0e, const f32x4 &c0o, const f32x4 &c1e,
                                               const f32x4 &c1o, int cg, int ks, int lane) {
  float *sp = s_part + (size_t)ks * R8 * FW + cg * 64 + lane;
#pragma unroll
  for (int r = 0; r < 4; ++r) sp[r * FW] = c0e[r] + c0o[r], sp[(4 + r) * FW] = c1e[r] + c1o[r];
}

template <int MODE = 0>
__global__ __launch_bounds__(1024) void timenet_fwd_fused8_kernel(FusedArgs g) {
  __shared__ __attribute__((aligned(16))) float s_c[R8 * GC_LD];
  __shared__ __attribute__((aligned(16))) float s_h[2][R8 * G_LD];
  __shared__ __attribute__((aligned(16))) float s_part[KS * R8 * FW];
  __shared__ float s_wo[7 * FW];
  // every bias of the net: a global load that is waited for in a layer's epilogue drains the weight ring with it
  // (s_waitcnt vmcnt counts in order), an LDS read does not
  __shared__ float s_bias[(MAX_LAYERS - 2) * FW + 8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cg = wave & 3, ks = wave >> 2;  // (the four K-slices of a column group share a SIMD)
  const int row0 = blockIdx.x * R8;
  for (int e = t; e < 7 * FW; e += 1024) s_wo[e] = e < 3 * FW ? g.W[g.D + 1][e] : g.W[g.D + 3][e - 3 * FW];
  for (int e = t; e < (g.D + 2) * FW; e += 1024) {
    const int st = e / FW, li = st >= g.D ? g.D + 2 * (st - g.D) : st;
    s_bias[e] = g.b[li][e & (FW - 1)];
  }
  if (t < 7) s_bias[(MAX_LAYERS - 2) * FW + t] = t < 3 ? g.b[g.D + 1][t] : g.b[g.D + 3][t - 3];
  float4 q[QF];
  Wave4 w = wave4(g.Wp[0], true, false, cg, ks, lane);
  {
    // the ring starts with the first quads of the wave's FIRST chunk: layer 0 has only the embedding segment, which
    // the K-slices >= KSE do not take -- their first chunk is layer 1's
    const float4 *f = w.has_e ? w.e : first_chunk(wave4(g.Wp[1], 0 == g.skip, true, cg, ks, lane));
#pragma unroll
    for (int u = 0; u < QF; ++u) q[u] = f[u * 64];  // in flight under the embedding
  }
  const int npts = 6 * g.pts_freqs, ntime = 2 * g.time_freqs;
  {  // embedding of the 8 rows (the arithmetic of embed_kernel): one element per thread
    const int m = t / FE_MAX, col = t % FE_MAX;
    const int row = min(row0 + m, g.R - 1);
    const int p = row / g.Mc, cp = row % g.Mc;
    float v = 0.f;
    if (col < npts) {
      const int f = col / 6, wd = col % 6;
      const float x = g.c_xyz[cp * 3 + (wd % 3)] * exp2f((float)f);
      v = wd < 3 ? sinf(x) : cosf(x);
    } else if (col < npts + ntime) {
      const int c = col - npts, f = c >> 1;
      const float x = g.pt.time[p] * exp2f((float)f);
      v = (c & 1) ? cosf(x) : sinf(x);
    } else if (col < g.E) {
      v = g.latent_table[(size_t)g.pt.latent_row[p] * g.latent_dim + (col - npts - ntime)];
    }
    s_c[m * GC_LD + col] = v;
    if (col < g.E && row0 + m < g.R) g.cat[(size_t)(row0 + m) * g.CAT + col] = v;
  }
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < g.D + 2; ++step) {
    const bool head = step >= g.D;
    const int li = head ? g.D + 2 * (step - g.D) : step;
    f32x4 c0e = {0.f, 0.f, 0.f, 0.f}, c0o = c0e, c1e = c0e, c1o = c0e;
    Wave4 wn = w;  // (after the last layer the ring refills with repeats of its own quads)
    if (step + 1 < g.D + 2) {
      const int ns = step + 1, nli = ns >= g.D ? g.D + 2 * (ns - g.D) : ns;
      wn = wave4(g.Wp[nli], ns < g.D && ns - 1 == g.skip, true, cg, ks, lane);
    }
    matmul4<MODE>(s_c, s_h[cur], w, first_chunk(wn), q, c0e, c0o, c1e, c1o, ks, lane);
    w = wn;
    store_partials(s_part, c0e, c0o, c1e, c1o, cg, ks, lane);
    lds_barrier();
    float *s_out = s_h[cur ^ 1];
    float *gout = head ? (step == g.D ? g.hp : g.hr) : g.act[li];
    const int ld_out = head ? FW : g.act_ld[li];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      // (this thread's two outputs: row t / 256 and four rows further, column t % 256)
      const int e = t + 1024 * h2, row = e / FW, col = e & (FW - 1);
      float v = s_bias[step * FW + col];
#pragma unroll
      for (int k = 0; k < KS; ++k) v += s_part[k * R8 * FW + e];
      v = fmaxf(v, 0.f);
      s_out[row * G_LD + col] = v;
      if (row0 + row < g.R) gout[(size_t)(row0 + row) * ld_out + col] = v;
    }
    lds_barrier();
    if (!head) {
      cur ^= 1;
      continue;
    }
    // head output columns (3 / 4): a wave per row, the arithmetic of head_out_kernel
    const int hd = step - g.D;
    const float *Wo = s_wo + (hd ? 3 * FW : 0), *bo = s_bias + (MAX_LAYERS - 2) * FW + (hd ? 3 : 0);
    const int nout = hd ? 4 : 3;
    if (wave < R8) {
      const int m = wave;
      const float *x = s_out + m * G_LD;
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = lane; c < FW; c += 64) {
        const float xv = x[c];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nout) a[i] = fmaf(xv, Wo[i * FW + c], a[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nout) a[i] = wave_sum(a[i]);
      if (lane == 0 && row0 + m < g.R) {
        float *dst = hd ? g.d_rot + (size_t)(row0 + m) * 4 : g.d_xyz + (size_t)(row0 + m) * 3;
        for (int i = 0; i < nout; ++i) dst[i] = a[i] + bo[i];
      }
    }
    lds_barrier();  // s_h[cur ^ 1] is overwritten by the second head
  }
}

// ---- the dgrad chain in ONE launch (same shape class) ------------------------------------------------------------
// Mirrors the forward: a workgroup owns 16 rows; dZ of the layer above stays in LDS (and goes to the workspace for the
// weight gradients), the TRANSPOSED weights stream through the same register ring.  Sequence of matmuls per workgroup
// (D = 8, skip = 4):  [dZp Wp0 + dZr Wr0] -> dZ7 ; dZ7 W7 -> dZ6 ; dZ6 W6 -> dZ5 ; dZ5 W5[:, E:] -> dZ4 and
// dZ5 W5[:, :E] -> gE ; dZ4 W4 -> dZ3 ; ... ; dZ1 W1 -> dZ0 ; dZ0 W0 -> gE (accumulated in registers, written once).
// Every hidden product is masked by the ReLU of the layer it enters (saved activation > 0).
//
// Packed transposed block (jt, nb), lane (kq, mn): float4 { W[16 nb + 4 kq + i][col0 + 16 jt + mn] }, i = 0..3
// (zero for columns past `ncols`): the contraction now runs over the ROWS of W.
struct PackTJob {
  const float *W;
  float *out;
  int ldw, col0, ncols, ntiles;
};
struct PackTArgs {
  PackTJob job[MAX_LAYERS + 2];
  int njobs;
};

struct FusedBwdArgs {
  int R, E, CAT, D, skip;
  const float *g_d_xyz, *g_d_rot, *Wp1, *Wr1, *hp, *hr;
  float *dzp, *dzr, *g_cat;
  const float *Tp[MAX_LAYERS];                  // packed transposed hidden-column weights of layer l (heads: D, D + 2)
  const float *TpE[MAX_LAYERS];                 // packed transposed embedding-column weights (layer 0, layer skip + 1)
  const float *mask[MAX_LAYERS];                // saved activation of layer l (ReLU mask of dZ[l])
  float *dz[MAX_LAYERS];
  int ld[MAX_LAYERS];                           // leading dimension of mask[l] / dz[l]
};

// (acc) * (mask > 0) -> LDS tile + workspace
__device__ __forceinline__ void fused_bwd_epilogue(const f32x4 &c0, const f32x4 &c1, const float (&mk)[8], float *s_out,
                                                   float *__restrict__ gout, int ld_out, int row0, int R, int lane,
                                                   int wave) {
  const int kq = lane >> 4, mn = lane & 15, n0 = wave * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * kq + r;
    const float v0 = mk[r] > 0.f ? c0[r] : 0.f, v1 = mk[4 + r] > 0.f ? c1[r] : 0.f;
    s_out[m * FH_LD + n0 + mn] = v0;
    s_out[m * FH_LD + n0 + 16 + mn] = v1;
    if (row0 + m < R) {
      gout[(size_t)(row0 + m) * ld_out + n0 + mn] = v0;
      gout[(size_t)(row0 + m) * ld_out + n0 + 16 + mn] = v1;
    }
  }
}


// ---- the dgrad chain with 8 rows per workgroup (4x4x1 MFMA, see timenet_fwd_fused8_kernel) ------------------------
// Same matmul sequence as timenet_bwd_fused_kernel; what differs:
//   * the ReLU masks of EVERY layer of the workgroup's 8 rows are fetched once, at the start, and kept as bits in LDS
//     (2.5 KB): a mask fetched per layer is a global load that the epilogue has to wait for, and on gfx9 that wait
//     (s_waitcnt vmcnt counts in order) drains the weight ring with it;
//   * the two embedding-gradient products (dZ[skip+1] W_{skip+1}[:, :E] and dZ[0] W_0) are ONE product over the
//     concatenated contraction at the very end -- 2 column groups x 8 K-slices = all 16 waves with a whole sub-chunk
//     each -- for which dZ[skip+1] stays in a fourth LDS tile.
// Transposed quads: quad (cg, kq) = 64 lanes x float4 { W[4 kq + i][col0 + 64 cg + lane] }, i = 0..3 (zero past
// `ncols`), at float4 index (cg * 64 + kq) * 64 + lane.
__global__ __launch_bounds__(256) void pack_weights_t_quads_kernel(PackTArgs a) {
  const PackTJob &j = a.job[blockIdx.y];
  const int total = j.ntiles * HQ * 64;  // float4s; ntiles = column groups of 64 here
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, qd = i >> 6;
    const int kq = qd % HQ, cg = qd / HQ;
    const int col = 64 * cg + lane, row = 4 * kq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < j.ncols) {
      const float *p = j.W + (size_t)row * j.ldw + j.col0 + col;
      v = make_float4(p[0], p[j.ldw], p[2 * (size_t)j.ldw], p[3 * (size_t)j.ldw]);
    }
    reinterpret_cast<float4 *>(j.out)[i] = v;
  }
}

__global__ __launch_bounds__(1024) void timenet_bwd_fused8_kernel(FusedBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float s_d[4][R8 * G_LD];
  __shared__ __attribute__((aligned(16))) float s_part[KS * R8 * FW];
  __shared__ unsigned long long s_mask[MAX_LAYERS * R8 * (FW / 64)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cg = wave & 3, ks = wave >> 2;
  const int row0 = blockIdx.x * R8;
  // this lane's float4 of the first quad of the wave's sub-chunk in a packed transposed matrix (4 column groups)
  auto hq = [&](const float *packed) {
    return reinterpret_cast<const float4 *>(packed) + ((size_t)cg * HQ + ks * SUB) * 64 + lane;
  };
  // ... and for the final embedding-gradient product: 2 column groups x 8 K-slices (4 per contracted matrix)
  const int cg2 = wave & 1, ks8 = wave >> 1;
  auto eq = [&](const float *packed) {
    return reinterpret_cast<const float4 *>(packed) + ((size_t)cg2 * HQ + (ks8 & 3) * SUB) * 64 + lane;
  };
  float4 q[QF];
  {
    const float4 *f = hq(g.Tp[g.D]);
#pragma unroll
    for (int u = 0; u < QF; ++u) q[u] = f[u * 64];
  }
  // ReLU masks of the 8 rows, every layer: a wave covers 64 consecutive columns of one row per round
  for (int m = 0; m < g.D; ++m) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int e = t + 1024 * h2, row = e / FW, col = e & (FW - 1);
      const float v = g.mask[m][(size_t)min(row0 + row, g.R - 1) * g.ld[m] + col];
      const unsigned long long bal = __ballot(v > 0.f);
      if (lane == 0) s_mask[(m * R8 + row) * (FW / 64) + (col >> 6)] = bal;
    }
  }
  // dZp = (g_d_xyz Wp1) * (hp > 0), dZr = (g_d_rot Wr1) * (hr > 0)   (head_out_bwd_kernel's arithmetic)
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int e = t + 1024 * h2, m = e / FW, c = e & (FW - 1);
    const int row = min(row0 + m, g.R - 1);
    float vp = 0.f, vr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) vp = fmaf(g.g_d_xyz[row * 3 + i], g.Wp1[i * FW + c], vp);
#pragma unroll
    for (int i = 0; i < 4; ++i) vr = fmaf(g.g_d_rot[row * 4 + i], g.Wr1[i * FW + c], vr);
    vp = g.hp[(size_t)row * FW + c] > 0.f ? vp : 0.f;
    vr = g.hr[(size_t)row * FW + c] > 0.f ? vr : 0.f;
    s_d[0][m * G_LD + c] = vp;
    s_d[1][m * G_LD + c] = vr;
    if (row0 + m < g.R) g.dzp[(size_t)(row0 + m) * FW + c] = vp, g.dzr[(size_t)(row0 + m) * FW + c] = vr;
  }
  __syncthreads();

  // the wave's K-slice of up to two products into the same accumulators (sub-chunks of one loop body: matmul4)
  auto mm = [&](int nsub, const float *t0, const float4 *w0, const float *t1, const float4 *w1, const float4 *nx,
                f32x4 &c0e, f32x4 &c0o, f32x4 &c1e, f32x4 &c1o) {
    for (int sc = 0; sc < nsub; ++sc) {
      const float *xa = (sc == 0 ? t0 : t1) + (lane & 3) * G_LD + ks * (FW / KS);
      const float4 *wq = sc == 0 ? w0 : w1;
      const float4 *nq = sc + 1 < nsub ? w1 : nx;
      chunk4<SUB>(xa, 4 * G_LD, wq, nq, q, c0e, c0o, c1e, c1o);
    }
  };
  // K-slices summed through LDS, ReLU mask of layer `ml` applied, -> LDS tile + workspace; `keep`: this step's INPUT
  // tile is copied to tile 3 (dZ of the layer that read the embedding, for the product at the end)
  auto epilogue = [&](const f32x4 &c0e, const f32x4 &c0o, const f32x4 &c1e, const f32x4 &c1o, int ml, float *s_out,
                      const float *keep) {
    store_partials(s_part, c0e, c0o, c1e, c1o, cg, ks, lane);
    lds_barrier();
    float *gout = g.dz[ml];
    const int ld_out = g.ld[ml];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int e = t + 1024 * h2, row = e / FW, col = e & (FW - 1);
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) v += s_part[k * R8 * FW + e];
      const unsigned long long bits = s_mask[(ml * R8 + row) * (FW / 64) + (col >> 6)];
      v = ((bits >> (col & 63)) & 1ull) ? v : 0.f;
      s_out[row * G_LD + col] = v;
      if (row0 + row < g.R) gout[(size_t)(row0 + row) * ld_out + col] = v;
      if (keep) s_d[3][row * G_LD + col] = keep[row * G_LD + col];
    }
    lds_barrier();
  };

  int cur;
  {  // head: dZ[D-1] = (dZp Wp0 + dZr Wr0) * (h[D-1] > 0)
    f32x4 c0e = {0.f, 0.f, 0.f, 0.f}, c0o = c0e, c1e = c0e, c1o = c0e;
    mm(2, s_d[0], hq(g.Tp[g.D]), s_d[1], hq(g.Tp[g.D + 2]), hq(g.Tp[g.D - 1]), c0e, c0o, c1e, c1o);
    epilogue(c0e, c0o, c1e, c1o, g.D - 1, s_d[2], nullptr);
    cur = 2;
  }
  // what the ring runs on into after layer 1: the wave's chunk of the final product (any valid quads for a wave that
  // takes no part in it)
  const bool from_skip = ks8 < 4;  // K-slices 0-3 contract dZ[skip+1], 4-7 dZ[0]
  const bool takes_part = !(from_skip && g.skip < 0);
  const float4 *final_w = eq(from_skip && g.skip >= 0 ? g.TpE[g.skip + 1] : g.TpE[0]);
  for (int l = g.D - 1; l >= 1; --l) {  // dZ[l-1] = (dZ[l] W_l[:, hidden columns]) * (h[l-1] > 0)
    f32x4 c0e = {0.f, 0.f, 0.f, 0.f}, c0o = c0e, c1e = c0e, c1o = c0e;
    const float4 *nx = l > 1 ? hq(g.Tp[l - 1]) : final_w;
    mm(1, s_d[cur], hq(g.Tp[l]), s_d[cur], hq(g.Tp[l]), nx, c0e, c0o, c1e, c1o);
    const int out = (cur + 1) % 3;
    epilogue(c0e, c0o, c1e, c1o, l - 1, s_d[out], l - 1 == g.skip ? s_d[cur] : nullptr);
    cur = out;
  }
  {  // gE = [dZ[skip+1] | dZ[0]] [W_{skip+1}[:, :E] ; W_0]: 2 column groups x 8 K-slices; leaves the chip once
    f32x4 c0e = {0.f, 0.f, 0.f, 0.f}, c0o = c0e, c1e = c0e, c1o = c0e;
    if (takes_part) {
      const float *tile = from_skip ? s_d[3] : s_d[cur];
      chunk4<SUB>(tile + (lane & 3) * G_LD + (ks8 & 3) * (FW / KS), 4 * G_LD, final_w, final_w, q, c0e, c0o, c1e, c1o);
    }
    float *sp = s_part + (size_t)ks8 * R8 * FE_MAX + cg2 * 64 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) sp[r * FE_MAX] = c0e[r] + c0o[r], sp[(4 + r) * FE_MAX] = c1e[r] + c1o[r];
    lds_barrier();
    const int row = t / FE_MAX, col = t & (FE_MAX - 1);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += s_part[k * R8 * FE_MAX + t];
    if (col < g.E && row0 + row < g.R) g.g_cat[(size_t)(row0 + row) * g.CAT + col] = v;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
struct Plan {
  int D, Wd, skip, E, CAT, rows;
  // float offsets into the workspace
  size_t cat, g_cat, hp, hr, dzp, dzr, act[MAX_LAYERS], dz[MAX_LAYERS], total;
  size_t packed[MAX_LAYERS];  // packed forward weights of layer l (fused forward only; 0 floats otherwise)
  int nkb[MAX_LAYERS];
  size_t packed_t[MAX_LAYERS], packed_te[MAX_LAYERS];  // packed transposed weights (fused backward): hidden / embedding columns
  bool has_t[MAX_LAYERS], has_te[MAX_LAYERS];
};

bool make_plan(const dimo_timenet_desc *d, int rows, Plan &pl) {
  if (!d || d->D < 2 || d->D + 4 > MAX_LAYERS || d->W < 1 || d->skip >= d->D - 1 || d->pts_freqs < 0 ||
      d->time_freqs < 0 || d->latent_dim < 0 || rows < 0)
    return false;
  pl.D = d->D, pl.Wd = d->W, pl.skip = d->skip, pl.rows = rows;
  pl.E = 6 * d->pts_freqs + 2 * d->time_freqs + d->latent_dim;
  pl.CAT = pl.E + (d->skip >= 0 ? d->W : 0);
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t at = o;
    o += (n + 63) & ~size_t(63);
    return at;
  };
  const size_t R = (size_t)rows;
  pl.cat = take(R * pl.CAT), pl.g_cat = take(R * pl.CAT);
  pl.hp = take(R * pl.Wd), pl.hr = take(R * pl.Wd), pl.dzp = take(R * pl.Wd), pl.dzr = take(R * pl.Wd);
  for (int l = 0; l < d->D; ++l) {
    if (l == d->skip) {
      pl.act[l] = pl.cat + pl.E, pl.dz[l] = pl.g_cat + pl.E;
    } else {
      pl.act[l] = take(R * pl.Wd), pl.dz[l] = take(R * pl.Wd);
    }
  }
  for (int l = 0; l < d->D + 4; ++l) pl.packed[l] = 0, pl.nkb[l] = 0;
  if (pl.Wd == 256) {  // FW (the fused forward's shape class; see fused_forward_ok)
    const int nE = embed_blocks(pl.E), nH = 256 / 16;
    for (int l = 0; l < d->D + 4; ++l) {
      const bool hidden_layer = l < d->D || l == d->D || l == d->D + 2;
      if (!hidden_layer) continue;
      const bool use_embed = l == 0 || (l < d->D && l - 1 == d->skip), use_hidden = l > 0;
      pl.nkb[l] = (use_embed ? nE : 0) + (use_hidden ? nH : 0);
      pl.packed[l] = take((size_t)256 * pl.nkb[l] * 16);
    }
  }
  for (int l = 0; l < d->D + 4; ++l) pl.has_t[l] = pl.has_te[l] = false, pl.packed_t[l] = pl.packed_te[l] = 0;
  if (pl.Wd == 256) {
    for (int l = 0; l < d->D + 4; ++l) {
      const bool hidden_in = (l >= 1 && l < d->D) || l == d->D || l == d->D + 2;  // layers with a hidden-width input
      const bool embed_in = l == 0 || (l < d->D && l - 1 == d->skip);
      if (hidden_in) pl.has_t[l] = true, pl.packed_t[l] = take((size_t)256 * 256);
      if (embed_in) pl.has_te[l] = true, pl.packed_te[l] = take((size_t)128 * 256);
    }
  }
  pl.total = o;
  return true;
}

// input of deformnet layer l: (offset, leading dimension, K)
void layer_input(const Plan &pl, int l, size_t &off, int &ld, int &K) {
  if (l == 0) {
    off = pl.cat, ld = pl.CAT, K = pl.E;
  } else if (l - 1 == pl.skip) {
    off = pl.cat, ld = pl.CAT, K = pl.CAT;
  } else {
    off = pl.act[l - 1], ld = pl.Wd, K = pl.Wd;
  }
}
int act_ld(const Plan &pl, int l) { return l == pl.skip ? pl.CAT : pl.Wd; }

// 16-byte loads are legal for an operand whose base is 16-B aligned, ld % 4 == 0 and contiguous extent % 4 == 0
bool vec_ok(const float *p, int ld, int extent) {
  return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (extent & 3) == 0;
}

template <bool A_KC, bool B_KC>
void launch_gemm(const GemmArgs &g, int z, hipStream_t s) {
  bool vec = true;
  for (int i = 0; i < g.nseg; ++i)
    vec = vec && vec_ok(g.A[i], g.lda[i], A_KC ? g.K[i] : g.M) && vec_ok(g.B[i], g.ldb[i], B_KC ? g.K[i] : g.N);
  if (z > 1) vec = vec && vec_ok(g.B_alt, g.ldb[0], B_KC ? g.K[0] : g.N);
  dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, z);
  if (vec)
    gemm_kernel<A_KC, B_KC, true><<<grid, 256, 0, s>>>(g);
  else
    gemm_kernel<A_KC, B_KC, false><<<grid, 256, 0, s>>>(g);
}

// the one-launch forward handles the reference's shape class: width 256, at most one skip, embedding <= 112 columns,
// every operand 16-byte aligned with row lengths that are multiples of 4
bool fused_forward_ok(const dimo_timenet_desc *d, const Plan &pl) {
  if (pl.Wd != FW || pl.E > FE_MAX || (pl.E & 3) || pl.E < 4 || d->D + 4 > MAX_LAYERS) return false;
  for (int l = 0; l < d->D + 4; ++l)
    if ((reinterpret_cast<uintptr_t>(d->weight[l]) & 15) != 0) return false;
  return true;
}

// rows per workgroup of the fused FORWARD: 16 (default) or 8 (DIMO_TIMENET_ROWS_FWD=8)
int timenet_rows(const char *var, int dflt) {
  const char *e = getenv(var);
  const int v = e ? atoi(e) : dflt;
  return v == 8 || v == 16 ? v : dflt;
}

// workgroups per packed matrix: one float4 per thread (the loops are grid-stride; 16 until the end of round 4, four to
// six dependent rounds per thread on the step's critical path)
unsigned pack_wgs() { return 64u; }

bool fill_pairs(int P, const float *times, const int *rows, PairTable &pt) {
  if (P > MAX_PAIRS) return false;
  for (int p = 0; p < P; ++p) pt.time[p] = times[p], pt.latent_row[p] = rows ? rows[p] : p;
  return true;
}

// the dgrad chain's packed (transposed) weights of every layer; fills g.Tp / g.TpE when `g` is given
void pack_t_jobs(const dimo_timenet_desc *d, const Plan &pl, float *ws, PackTArgs &pa, FusedBwdArgs *g) {
  const int D = d->D, Wd = pl.Wd;
  for (int l = 0; l < D + 4; ++l) {
    const int K = l == 0 ? pl.E : (l < D && l - 1 == pl.skip ? pl.CAT : Wd);  // row length of weight[l]
    if (pl.has_t[l]) {
      PackTJob &j = pa.job[pa.njobs++];
      j.W = d->weight[l], j.out = ws + pl.packed_t[l], j.ldw = K, j.col0 = K - Wd, j.ncols = Wd, j.ntiles = 16;
      if (g) g->Tp[l] = j.out;
    }
    if (pl.has_te[l]) {
      PackTJob &j = pa.job[pa.njobs++];
      j.W = d->weight[l], j.out = ws + pl.packed_te[l], j.ldw = K, j.col0 = 0, j.ncols = pl.E, j.ntiles = 8;
      if (g) g->TpE[l] = j.out;
    }
  }
  for (int k = 0; k < pa.njobs; ++k) pa.job[k].ntiles /= 4;  // (column groups of 64, not tiles of 16)
}
void launch_pack_t(const PackTArgs &pa, hipStream_t s) {
  pack_weights_t_quads_kernel<<<dim3(pack_wgs(), pa.njobs), 256, 0, s>>>(pa);
}

}  // namespace

extern "C" size_t dimo_timenet_workspace_bytes(const dimo_timenet_desc *d, int P, int M) {
  Plan pl;
  if (P < 0 || M < 0 || !make_plan(d, P * M, pl)) return 0;
  return pl.total * sizeof(float) + 256;
}

extern "C" int dimo_timenet_forward(const dimo_timenet_desc *d, int P, int M, const float *c_xyz,
                                    const float *times_host, const float *latent_table, const int *latent_rows_host,
                                    float *d_xyz, float *d_rot, void *workspace, size_t workspace_bytes,
                                    void *stream) {
  Plan pl;
  if (P < 0 || M < 0 || !make_plan(d, P * M, pl)) return DIMO_E_ARG;
  if (P == 0 || M == 0) return DIMO_OK;
  if (!c_xyz || !times_host || !d_xyz || !d_rot || !workspace || (d->latent_dim > 0 && !latent_table))
    return DIMO_E_ARG;
  for (int l = 0; l < d->D + 4; ++l)
    if (!d->weight[l] || !d->bias[l]) return DIMO_E_ARG;
  if (workspace_bytes < pl.total * sizeof(float)) return DIMO_E_WORKSPACE;
  PairTable pt;
  if (!fill_pairs(P, times_host, latent_rows_host, pt)) return DIMO_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_errors();
  ScopedTimer timer(T_TIMENET_FWD, s);
  float *ws = static_cast<float *>(workspace);
  const int R = pl.rows;
  if (fused_forward_ok(d, pl)) {
    FusedArgs g = {};
    g.R = R, g.Mc = M, g.E = pl.E, g.CAT = pl.CAT, g.D = d->D, g.skip = d->skip, g.pts_freqs = d->pts_freqs;
    g.time_freqs = d->time_freqs, g.latent_dim = d->latent_dim, g.c_xyz = c_xyz, g.latent_table = latent_table;
    for (int l = 0; l < d->D + 4; ++l) g.W[l] = d->weight[l], g.b[l] = d->bias[l];
    g.cat = ws + pl.cat, g.hp = ws + pl.hp, g.hr = ws + pl.hr, g.d_xyz = d_xyz, g.d_rot = d_rot;
    for (int l = 0; l < d->D; ++l) g.act[l] = ws + pl.act[l], g.act_ld[l] = act_ld(pl, l);
    g.pt = pt;
    PackArgs pa = {};
    for (int l = 0; l < d->D + 4; ++l) {
      if (!pl.nkb[l]) continue;
      const bool use_embed = l == 0 || (l < d->D && l - 1 == d->skip), use_hidden = l > 0;
      PackJob &j = pa.job[pa.njobs++];
      j.W = d->weight[l], j.out = ws + pl.packed[l], j.E_seg = use_embed ? pl.E : 0, j.H_seg = use_hidden ? FW : 0;
      j.ldw = j.E_seg + j.H_seg, j.nkb = pl.nkb[l];
      g.Wp[l] = ws + pl.packed[l];
    }
    // 16 rows per workgroup (128 of the 256 CUs for the benchmark's 2048 rows) by default: alone on the device the
    // 8-row kernel is faster (55 against 69 us), but in the training step the forward runs NEXT TO the step's KNN, which
    // takes the other half of the chip: 77 us against 84 (both kernels in flight).  DIMO_TIMENET_ROWS_FWD=8 selects it.
    static const bool rows8 = timenet_rows("DIMO_TIMENET_ROWS_FWD", 16) == 8;
    if (rows8) {
      pack_weights_quads_kernel<<<dim3(pack_wgs(), pa.njobs), 256, 0, s>>>(pa);
      timenet_fwd_fused8_kernel<0><<<(R + R8 - 1) / R8, 1024, 0, s>>>(g);
      return check_launch();
    }
    pack_weights_kernel<<<dim3(pack_wgs(), pa.njobs), 256, 0, s>>>(pa);
    timenet_fwd_fused_kernel<16><<<(R + FR - 1) / FR, 1024, 0, s>>>(g);
    return check_launch();
  }
  {
    const int n = R * pl.E;
    embed_kernel<<<(n + 255) / 256, 256, 0, s>>>(P, M, pl.E, pl.CAT, d->pts_freqs, d->time_freqs, d->latent_dim,
                                                 c_xyz, latent_table, pt, ws + pl.cat);
  }
  for (int l = 0; l < d->D; ++l) {
    GemmArgs g = {};
    size_t off;
    layer_input(pl, l, off, g.lda[0], g.K[0]);
    g.A[0] = ws + off, g.B[0] = d->weight[l], g.ldb[0] = g.K[0], g.nseg = 1;
    g.C = ws + pl.act[l], g.ldc = act_ld(pl, l), g.M = R, g.N = pl.Wd, g.bias = d->bias[l], g.relu = 1;
    launch_gemm<true, true>(g, 1, s);
  }
  {  // pts_layers[0] and rot_layers[0] share the input
    GemmArgs g = {};
    g.A[0] = ws + pl.act[d->D - 1], g.lda[0] = act_ld(pl, d->D - 1), g.K[0] = pl.Wd, g.nseg = 1;
    g.B[0] = d->weight[d->D], g.ldb[0] = pl.Wd, g.bias = d->bias[d->D], g.C = ws + pl.hp;
    g.B_alt = d->weight[d->D + 2], g.bias_alt = d->bias[d->D + 2], g.C_alt = ws + pl.hr;
    g.ldc = pl.Wd, g.M = R, g.N = pl.Wd, g.relu = 1;
    launch_gemm<true, true>(g, 2, s);
  }
  head_out_kernel<<<(R + 3) / 4, 256, 0, s>>>(R, pl.Wd, ws + pl.hp, ws + pl.hr, d->weight[d->D + 1],
                                              d->bias[d->D + 1], d->weight[d->D + 3], d->bias[d->D + 3], d_xyz,
                                              d_rot);
  return check_launch();
}

extern "C" int dimo_timenet_backward(const dimo_timenet_desc *d, int P, int M, const float *g_d_xyz,
                                     const float *g_d_rot, const float *times_host, const int *latent_rows_host,
                                     float *g_c_xyz, float *g_latent_table, void *workspace, size_t workspace_bytes,
                                     void *stream) {
  Plan pl;
  if (P < 0 || M < 0 || !make_plan(d, P * M, pl)) return DIMO_E_ARG;
  if (P == 0 || M == 0) return DIMO_OK;
  if (!g_d_xyz || !g_d_rot || !times_host || !workspace) return DIMO_E_ARG;
  for (int l = 0; l < d->D + 4; ++l)
    if (!d->weight[l] || !d->g_weight[l] || !d->g_bias[l]) return DIMO_E_ARG;
  if (workspace_bytes < pl.total * sizeof(float)) return DIMO_E_WORKSPACE;
  PairTable pt;
  if (!fill_pairs(P, times_host, latent_rows_host, pt)) return DIMO_E_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  clear_errors();
  ScopedTimer timer(T_TIMENET_BWD, s);
  float *ws = static_cast<float *>(workspace);
  const int R = pl.rows, D = d->D, Wd = pl.Wd;
  const bool fused = fused_forward_ok(d, pl);
  if (fused) {
    PackTArgs pa = {};
    FusedBwdArgs g = {};
    g.R = R, g.E = pl.E, g.CAT = pl.CAT, g.D = D, g.skip = pl.skip;
    g.g_d_xyz = g_d_xyz, g.g_d_rot = g_d_rot, g.Wp1 = d->weight[D + 1], g.Wr1 = d->weight[D + 3];
    g.hp = ws + pl.hp, g.hr = ws + pl.hr, g.dzp = ws + pl.dzp, g.dzr = ws + pl.dzr, g.g_cat = ws + pl.g_cat;
    pack_t_jobs(d, pl, ws, pa, &g);
    for (int l = 0; l < D; ++l) g.mask[l] = ws + pl.act[l], g.dz[l] = ws + pl.dz[l], g.ld[l] = act_ld(pl, l);
    // The pack depends on the weights alone, but it belongs HERE: written right in front of the chain, the packed
    // weights are in the L2 when the chain streams them.  Packed early on this stream, in the shadow of the renders
    // (a `dimo_timenet_backward_prepare` call after the forks), the launch left the serial tail and the chain lost
    // more than that -- backward group 106 -> 119 us in the step, 8098 -> 8054 frames/s; packed NEXT TO the forward on
    // a private stream, the forward (bound by its own weight stream) went 69 -> 79 us, 8073 -> 7982 frames/s.
    launch_pack_t(pa, s);
    // the dgrad chain: 8 rows per workgroup on every CU (4x4x1 MFMA; 103 against 115 us for the backward group with
    // round 4's 16-row chain on 16x16x4, and the same in the training step; that chain is gone since round 5)
    timenet_bwd_fused8_kernel<<<(R + R8 - 1) / R8, 1024, 0, s>>>(g);
  }
  if (!fused) {
    const int n = R * Wd;
    head_out_bwd_kernel<<<(n + 255) / 256, 256, 0, s>>>(R, Wd, g_d_xyz, g_d_rot, ws + pl.hp, ws + pl.hr,
                                                        d->weight[D + 1], d->weight[D + 3], ws + pl.dzp,
                                                        ws + pl.dzr);
  }
  if (!fused) {  // dZ[D-1] = (dZp Wp0 + dZr Wr0) * (h[D-1] > 0)
    GemmArgs g = {};
    g.nseg = 2;
    g.A[0] = ws + pl.dzp, g.A[1] = ws + pl.dzr, g.lda[0] = g.lda[1] = Wd, g.K[0] = g.K[1] = Wd;
    g.B[0] = d->weight[D], g.B[1] = d->weight[D + 2], g.ldb[0] = g.ldb[1] = Wd;
    g.C = ws + pl.dz[D - 1], g.ldc = act_ld(pl, D - 1), g.M = R, g.N = Wd;
    g.mask = ws + pl.act[D - 1], g.ldmask = act_ld(pl, D - 1);
    launch_gemm<true, false>(g, 1, s);
  }
  for (int l = D - 1; l >= 0 && !fused; --l) {  // gradient of layer l's input
    GemmArgs g = {};
    size_t in_off;
    int in_ld, K;
    layer_input(pl, l, in_off, in_ld, K);
    g.nseg = 1, g.A[0] = ws + pl.dz[l], g.lda[0] = act_ld(pl, l), g.K[0] = Wd;
    g.B[0] = d->weight[l], g.ldb[0] = K, g.M = R, g.N = K;
    if (l == 0) {  // embedding columns: accumulate onto the skip layer's contribution (if any)
      g.C = ws + pl.g_cat, g.ldc = pl.CAT, g.accumulate = pl.skip >= 0;
    } else if (l - 1 == pl.skip) {  // [embedding | h_skip]: only the h columns pass a ReLU
      g.C = ws + pl.g_cat, g.ldc = pl.CAT, g.mask = ws + pl.cat, g.ldmask = pl.CAT, g.mask_from_col = pl.E;
    } else {
      g.C = ws + pl.dz[l - 1], g.ldc = act_ld(pl, l - 1);
      g.mask = ws + pl.act[l - 1], g.ldmask = act_ld(pl, l - 1);
    }
    launch_gemm<true, false>(g, 1, s);
  }
  EmbedBwdArgs eb = {};
  {
    eb.rows = R, eb.Mc = M, eb.ld = pl.CAT, eb.pts_freqs = d->pts_freqs;
    eb.lat0 = 6 * d->pts_freqs + 2 * d->time_freqs, eb.latent_dim = d->latent_dim;
    eb.cat = ws + pl.cat, eb.g_cat = ws + pl.g_cat;
    eb.g_c_xyz = d->pts_freqs > 0 ? g_c_xyz : nullptr, eb.g_latent_table = d->latent_dim > 0 ? g_latent_table : nullptr;
    for (int q = 0; q < P; ++q) eb.latent_row[q] = pt.latent_row[q];
    eb.nblocks = (eb.g_c_xyz || eb.g_latent_table) ? (R + 63) / 64 : 0;
  }
  {  // every weight / bias gradient in one grouped split-K launch
    // 64 x 64 tiles of 32x32x2 MFMAs (wgrad64_kernel), the embedding backward in the same launch.  (Rounds 2-4 ran
    // 32 x 64 tiles of 16x16x4 MFMAs with 2-way LDS bank conflicts on every operand and the embedding backward as a
    // launch of its own: profiles/r04_timenet_wgrad64.txt; removed in round 5.)
    const int tM = W_T, tN = W_T;
    WgradArgs w = {};
    int tiles = 0, np = 0;
    auto add = [&](const float *dz, int ld_dz, const float *x, int ld_x, int Mo, int No, int li) {
      WgradProblem &p = w.p[np++];
      p.dZ = dz, p.ld_dz = ld_dz, p.X = x, p.ld_x = ld_x, p.gW = d->g_weight[li], p.gbias = d->g_bias[li];
      p.ld_w = No, p.M = Mo, p.N = No, p.tiles_n = (No + tN - 1) / tN, p.tile_begin = tiles;
      p.vec = (vec_ok(dz, ld_dz, Mo) ? 1 : 0) | (vec_ok(x, ld_x, No) ? 2 : 0);
      tiles += ((Mo + tM - 1) / tM) * p.tiles_n;
    };
    for (int l = 0; l < D; ++l) {
      size_t off;
      int ld, K;
      layer_input(pl, l, off, ld, K);
      add(ws + pl.dz[l], act_ld(pl, l), ws + off, ld, Wd, K, l);
    }
    const float *hlast = ws + pl.act[D - 1];
    const int ldl = act_ld(pl, D - 1);
    add(ws + pl.dzp, Wd, hlast, ldl, Wd, Wd, D);
    add(g_d_xyz, 3, ws + pl.hp, Wd, 3, Wd, D + 1);
    add(ws + pl.dzr, Wd, hlast, ldl, Wd, Wd, D + 2);
    add(g_d_rot, 4, ws + pl.hr, Wd, 4, Wd, D + 3);
    w.nprob = np, w.rows = R;
    {
      // ONE round of workgroups, four per CU; K chunks of whole half stages.  (Measured and removed: XCD-contiguous
      // workgroup ids, 101.6 against 93.2 us for the backward group; 512 / 2048 / 4096 workgroups, 105 / 97 / 111.)
      const int target = 1024;
      const int halves = (R + W_K / 2 - 1) / (W_K / 2);
      int ksplit = (target + tiles / 2) / tiles;
      ksplit = ksplit < 1 ? 1 : (ksplit > halves ? halves : ksplit);
      w.kchunk = ((halves + ksplit - 1) / ksplit) * (W_K / 2);
      ksplit = (R + w.kchunk - 1) / w.kchunk;
      w.tiles = tiles;
      wgrad64_kernel<<<dim3(eb.nblocks + tiles * ksplit), 256, 0, s>>>(w, eb);
    }
  }
  return check_launch();
}
