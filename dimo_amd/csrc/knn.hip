// Nearest-neighbour kernels of the deform path.
//   dimo_knn   : knn_cuda.KNN(k, transpose_mode=True)   (main_train_dimo.py:502-509) -- every Gaussian's k
//                nearest control points, once per training step (N = 1e5 queries, M = 512 references).
//   dimo_dist2 : simple_knn._C.distCUDA2                (renderer/latent_gs_renderer.py:426) -- mean squared
//                distance to the 3 nearest other points, used once to initialise the scales.
// Both are brute force with the candidate set tiled through LDS (every lane reads the same candidate:
// conflict-free LDS broadcast).  Compiled with -ffp-contract=off: d2 = (dx*dx + dy*dy) + dz*dz exactly as
// the CPU oracle evaluates it, so distances AND indices are bit-exact.
#pragma clang fp contract(off)
#include "common.hpp"

namespace dimo {

constexpr int KNN_BLOCK = 64;     // one wave per workgroup: 1563 workgroups for 1e5 queries keep 256 CUs evenly loaded
constexpr int KNN_CHUNK = 512;    // reference points staged per LDS round (8 KiB as float4: with the 8 KiB of partial lists a
                                  // workgroup takes 16 KiB, so 8 of them fit a CU and the 1563 workgroups of 1e5 queries are
                                  // resident at once -- at 24 KiB only 6 were, and a seventh per CU doubled the duration)

// K = 4 (DIMO's setting): the four best (distance, index) pairs live as four 64-bit keys
//     key = (bits of d2) << 32 | index
// reinterpreted as doubles.  d2 >= 0, so the float's bit pattern orders like the number, the index in the low word
// breaks ties towards the lower index (the reference's strict '<' in ascending index order), and every such pattern
// is a finite non-negative double (a zero distance gives a subnormal, which v_min/max_f64 order correctly; f64
// denormals are never flushed on gfx9).  Inserting a candidate is then a branch-free 7-instruction min/max chain
// instead of a divergent compare-and-swap ladder.
// Four waves share 64 queries: wave w scans every fourth chunk-quarter of the references and keeps its own four
// best keys; the keys order totally (distance bits, then index), so the four partial lists merge into the same
// result in any order.  (One wave per 64 queries scanned all 512 references: 1563 waves for 1024 SIMDs, each a
// dependent chain of 512 x 17 instructions -- 47 us of mostly latency at the head of every training step.)
constexpr int KNN4_WAVES = 4;
__device__ __forceinline__ void knn4_insert(double &b0, double &b1, double &b2, double &b3, double key) {
  const double r0 = fmax(b0, key);
  b0 = fmin(b0, key);
  const double r1 = fmax(b1, r0);
  b1 = fmin(b1, r0);
  const double r2 = fmax(b2, r1);
  b2 = fmin(b2, r1);
  b3 = fmin(b3, r2);
}
// One block of KNN_BLOCK queries by the workgroup's KNN4_WAVES waves (each scans a share of the candidates).
__device__ __forceinline__ void knn4_block(int blk, int M, int N, int k, const float *__restrict__ ref,
                                           const float *__restrict__ query, float *__restrict__ dist,
                                           int64_t *__restrict__ idx, const int64_t *__restrict__ seed,
                                           float4 *s_ref, double (*s_best)[4][KNN_BLOCK]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blk * KNN_BLOCK + lane;
  float qx = 0, qy = 0, qz = 0;
  if (i < N) qx = query[3 * i], qy = query[3 * i + 1], qz = query[3 * i + 2];
  const double empty = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(INFINITY) << 32) | 0xffffffffull));
  double b0 = empty, b1 = empty, b2 = empty, b3 = empty;
  // Seeds (optional): four DISTINCT reference indices per query -- the previous step's neighbours.  The largest of
  // their distances bounds the query's true fourth-nearest distance from above, so only candidates within it can
  // enter the list: the result is the same as without seeds whatever the seeds are (stale ones only loosen the
  // bound; a repeated or out-of-range index switches the bound off for that query).
  float worst = INFINITY;
  if (seed && k == 4 && i < N) {
    const long long s0 = seed[4 * (size_t)i], s1 = seed[4 * (size_t)i + 1], s2 = seed[4 * (size_t)i + 2],
                    s3 = seed[4 * (size_t)i + 3];
    const bool ok = s0 >= 0 && s0 < M && s1 >= 0 && s1 < M && s2 >= 0 && s2 < M && s3 >= 0 && s3 < M && s0 != s1 &&
                    s0 != s2 && s0 != s3 && s1 != s2 && s1 != s3 && s2 != s3;
    if (ok) {
      float bound = 0.0f;
      const long long ss[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float *r = ref + 3 * (size_t)ss[j];
        const float dx = qx - r[0], dy = qy - r[1], dz = qz - r[2];
        bound = fmaxf(bound, dx * dx + dy * dy + dz * dz);  // (a NaN distance is dropped by fmaxf: the bound stays valid
      }                                                     //  for the finite ones, and NaNs never enter a list anyway)
      worst = bound;
    }
  }
  // Block bound (seeded searches).  The 64 queries of a block are Morton neighbours (densify.py: sort_spatially): with
  // c the centre of their bounding box, r = max |q - c| and D = the largest seed bound, a candidate p with
  // |p - c| > D + r is farther than D from EVERY query of the block (triangle inequality), so it fails every lane's
  // `d2 <= worst` test below and can be skipped without evaluating it 64 times.  Each wave filters its share of the
  // candidates with one distance per lane and walks the survivors (a few per cent of the 512 control points); the
  // survivors are evaluated exactly as before, so distances and indices stay bit-exact.  Any non-finite query or
  // missing bound in the block switches the filter off for the block.
  float cx = 0.f, cy = 0.f, cz = 0.f, T2 = INFINITY;
  bool prune = false;
  {
    const bool live = i < N;
    const bool ok = !live || (worst < INFINITY && fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY);
    if (seed && __builtin_amdgcn_ballot_w64(!ok) == 0ull) {
      float lo[3] = {live ? qx : INFINITY, live ? qy : INFINITY, live ? qz : INFINITY};
      float hi[3] = {live ? qx : -INFINITY, live ? qy : -INFINITY, live ? qz : -INFINITY};
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
          hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
      cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
      const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
      float r2 = live ? ex * ex + ey * ey + ez * ez : 0.f, dm = live ? worst : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        r2 = fmaxf(r2, __shfl_xor(r2, o, 64));
        dm = fmaxf(dm, __shfl_xor(dm, o, 64));
      }
      const float T = (sqrtf(dm) + sqrtf(r2)) * 1.0001f + 1e-6f;  // (generous against the rounding of the filter itself)
      T2 = T * T;
      prune = T2 < INFINITY;  // (an empty block has lo = +inf: NaN centre, T2 stays unusable -> false)
    }
  }
  for (int base = 0; base < M; base += KNN_CHUNK) {
    const int cnt = min(KNN_CHUNK, M - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += KNN_BLOCK * KNN4_WAVES) {
      const float *r = ref + (size_t)(base + t) * 3;
      s_ref[t] = make_float4(r[0], r[1], r[2], 0.0f);
    }
    __syncthreads();
    const int per = (cnt + KNN4_WAVES - 1) / KNN4_WAVES;
    const int m0 = wave * per, m1 = min(cnt, m0 + per);
    if (prune) {
      for (int off = m0; off < m1; off += 64) {
        bool near = false;
        if (off + lane < m1) {
          const float4 c = s_ref[off + lane];
          const float fx = c.x - cx, fy = c.y - cy, fz = c.z - cz;
          near = fx * fx + fy * fy + fz * fz <= T2;
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(near);
        while (mask) {  // wave-uniform: the survivors of this share, in index order
          const int m = off + (int)__builtin_ctzll(mask);
          mask &= mask - 1ull;
          const float4 c = s_ref[m];
          const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (__builtin_amdgcn_ballot_w64(d2 <= worst) != 0ull) {
            knn4_insert(b0, b1, b2, b3, __hiloint2double((int)__float_as_uint(d2), base + m));
            worst = fminf(worst, __uint_as_float((unsigned)__double2hiint(b3)));
          }
        }
      }
      continue;
    }
    // A candidate can enter a lane's list only if it is not farther than the lane's current bound (the seed bound,
    // then the fourth best: an equal distance with a higher index loses the key comparison inside the insert): the
    // 12-instruction f64 insert runs only when some lane of the wave needs it.  With the Gaussians in Morton order
    // (densify.py: sort_spatially) the 64 queries of a wave share their nearest control points, and with seeds nearly
    // all of the 512 candidates are rejected by every lane.
    int m = m0;
    for (; m + 4 <= m1; m += 4) {  // four candidates' LDS reads in flight, then one uniform test per candidate
      float4 c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = s_ref[m + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dx = qx - c[u].x, dy = qy - c[u].y, dz = qz - c[u].z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (__builtin_amdgcn_ballot_w64(d2 <= worst) != 0ull) {
          knn4_insert(b0, b1, b2, b3, __hiloint2double((int)__float_as_uint(d2), base + m + u));
          worst = fminf(worst, __uint_as_float((unsigned)__double2hiint(b3)));
        }
      }
    }
    for (; m < m1; ++m) {
      const float4 c = s_ref[m];
      const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (__builtin_amdgcn_ballot_w64(d2 <= worst) != 0ull) {
        knn4_insert(b0, b1, b2, b3, __hiloint2double((int)__float_as_uint(d2), base + m));
        worst = fminf(worst, __uint_as_float((unsigned)__double2hiint(b3)));
      }
    }
  }
  s_best[wave][0][lane] = b0, s_best[wave][1][lane] = b1, s_best[wave][2][lane] = b2, s_best[wave][3][lane] = b3;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < KNN4_WAVES; ++w)
#pragma unroll
    for (int j = 0; j < 4; ++j) knn4_insert(b0, b1, b2, b3, s_best[w][j][lane]);
  if (i < N) {
    const double b[4] = {b0, b1, b2, b3};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < k) {
        const int lo = __double2loint(b[j]);
        dist[(size_t)i * k + j] = sqrtf(__uint_as_float((unsigned)__double2hiint(b[j])));
        idx[(size_t)i * k + j] = (int64_t)(lo == -1 && __double2hiint(b[j]) == (int)__float_as_uint(INFINITY) ? -1 : lo);
      }
  }
}
// Persistent workgroups over the query blocks: the grid is capped (dimo_knn_seeded) so that the search leaves room on
// every CU -- in the training step it runs on a private stream NEXT TO the TimeNet forward, whose workgroups (16 waves,
// 60 KB of LDS, one per CU) cannot start on a CU whose wave slots a flood of short search workgroups has taken.
__global__ void __launch_bounds__(KNN_BLOCK *KNN4_WAVES) knn4_kernel(int M, int N, int k, const float *__restrict__ ref,
                                                                      const float *__restrict__ query,
                                                                      float *__restrict__ dist, int64_t *__restrict__ idx,
                                                                      const int64_t *__restrict__ seed) {
  __shared__ float4 s_ref[KNN_CHUNK];
  __shared__ double s_best[KNN4_WAVES][4][KNN_BLOCK];
  const int nblk = (N + KNN_BLOCK - 1) / KNN_BLOCK;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    knn4_block(blk, M, N, k, ref, query, dist, idx, seed, s_ref, s_best);
    __syncthreads();  // (s_best / s_ref are re-used by the next block)
  }
}

template <int K>
__global__ void __launch_bounds__(KNN_BLOCK) knn_kernel(int M, int N, int k, const float *__restrict__ ref,
                                                        const float *__restrict__ query, float *__restrict__ dist,
                                                        int64_t *__restrict__ idx) {
  __shared__ float4 s_ref[KNN_CHUNK];
  const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  float qx = 0, qy = 0, qz = 0;
  if (i < N) qx = query[3 * i], qy = query[3 * i + 1], qz = query[3 * i + 2];
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) bd[j] = INFINITY, bi[j] = -1;

  for (int base = 0; base < M; base += KNN_CHUNK) {
    const int cnt = min(KNN_CHUNK, M - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += KNN_BLOCK) {
      const float *r = ref + (size_t)(base + t) * 3;
      s_ref[t] = make_float4(r[0], r[1], r[2], 0.0f);
    }
    __syncthreads();
    for (int m = 0; m < cnt; ++m) {
      const float4 c = s_ref[m];
      const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < bd[K - 1]) {
        // insertion keeping ascending order; strict '<' => ties keep the lower index first
        bd[K - 1] = d2, bi[K - 1] = base + m;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
          if (bd[j] < bd[j - 1]) {
            const float td = bd[j];
            bd[j] = bd[j - 1], bd[j - 1] = td;
            const int ti = bi[j];
            bi[j] = bi[j - 1], bi[j - 1] = ti;
          }
        }
      }
    }
  }
  if (i < N) {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j < k) {
        dist[(size_t)i * k + j] = sqrtf(bd[j]);
        idx[(size_t)i * k + j] = (int64_t)bi[j];
      }
  }
}

constexpr int D2_BLOCK = 256;
constexpr int D2_CHUNK = 2048;

__global__ void __launch_bounds__(D2_BLOCK) dist2_kernel(int N, const float *__restrict__ pts,
                                                         float *__restrict__ out) {
  __shared__ float s_p[D2_CHUNK * 3];
  const int i = blockIdx.x * D2_BLOCK + threadIdx.x;
  float qx = 0, qy = 0, qz = 0;
  if (i < N) qx = pts[3 * i], qy = pts[3 * i + 1], qz = pts[3 * i + 2];
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  for (int base = 0; base < N; base += D2_CHUNK) {
    const int cnt = min(D2_CHUNK, N - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += D2_BLOCK) s_p[t] = pts[(size_t)base * 3 + t];
    __syncthreads();
    for (int m = 0; m < cnt; ++m) {
      const float dx = qx - s_p[3 * m], dy = qy - s_p[3 * m + 1], dz = qz - s_p[3 * m + 2];
      float d2 = dx * dx + dy * dy + dz * dz;
      if (base + m == i) d2 = INFINITY;  // exclude the point itself (by index, not by distance)
      // keep the three smallest: branch-free min/max network
      const float n0 = fminf(b0, d2), r0 = fmaxf(b0, d2);
      const float n1 = fminf(b1, r0), r1 = fmaxf(b1, r0);
      b0 = n0, b1 = n1, b2 = fminf(b2, r1);
    }
  }
  if (i < N) out[i] = (b0 + b1 + b2) / 3.0f;
}

// ---- dist2 on a uniform grid ------------------------------------------------------------------------------------
// The brute-force kernel above is O(N^2): 5.3 ms at 1e5 points, half a second at 1e6 (upstream simple-knn orders the
// points along a Morton curve and prunes boxes).  Here: a uniform grid over the bounding box with ~4 points per cell;
// the points are counting-sorted by cell; a point scans the 3 x 3 x 3 block of cells around its own, then ring after
// ring of the surrounding cells until its third-best squared distance cannot be beaten by anything outside the
// block scanned so far (distance from the point to the block's nearest face, per axis, minus a rounding margin) --
// or the block covers the whole grid.  The three smallest d2 = (dx*dx + dy*dy) + dz*dz are the same numbers whatever
// the order they are met in, so the result equals the brute-force kernel's bit for bit.
struct D2Grid {
  int gx, gy, gz;       // cells per axis (1 on an axis without extent)
  uint32_t *bbox;       // [6] ordered-uint min xyz, max xyz
  uint32_t *count;      // [cells + 1] points per cell, then the exclusive scan
  uint32_t *cursor;     // [cells]
  float4 *sorted;       // [N] (x, y, z, bits of the original index)
};
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ void __launch_bounds__(256) d2_bbox_kernel(int N, const float *__restrict__ pts, uint32_t *__restrict__ bbox) {
  __shared__ uint32_t s[6];
  if (threadIdx.x < 3) s[threadIdx.x] = 0xffffffffu, s[3 + threadIdx.x] = 0u;
  __syncthreads();
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t u = f2ord(pts[3 * (size_t)i + a]);
      mn[a] = min(mn[a], u), mx[a] = max(mx[a], u);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) atomicMin(&s[a], mn[a]), atomicMax(&s[3 + a], mx[a]);
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], s[threadIdx.x]), atomicMax(&bbox[3 + threadIdx.x], s[3 + threadIdx.x]);
}
struct D2Map {  // origin, cells per unit length, cell size per axis; cells (1 where the box has no extent)
  float o[3], inv[3], cs[3];
  int g[3];
};
__device__ __forceinline__ D2Map d2_map(const D2Grid &G) {
  D2Map m;
  const int g[3] = {G.gx, G.gy, G.gz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = ord2f(G.bbox[a]), hi = ord2f(G.bbox[3 + a]);
    const float ext = hi - lo;
    m.o[a] = lo;
    m.g[a] = (ext > 0.0f && ext < INFINITY) ? g[a] : 1;
    m.cs[a] = m.g[a] > 1 ? ext / (float)m.g[a] : 0.0f;
    m.inv[a] = m.cs[a] > 0.0f ? 1.0f / m.cs[a] : 0.0f;
  }
  return m;
}
__device__ __forceinline__ int d2_cell_axis(const D2Map &m, int a, float p) {
  return m.g[a] > 1 ? min(m.g[a] - 1, max(0, (int)((p - m.o[a]) * m.inv[a]))) : 0;
}
__global__ void __launch_bounds__(256) d2_count_kernel(int N, const float *__restrict__ pts, D2Grid G) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const D2Map m = d2_map(G);
  const int cx = d2_cell_axis(m, 0, pts[3 * (size_t)i]), cy = d2_cell_axis(m, 1, pts[3 * (size_t)i + 1]),
            cz = d2_cell_axis(m, 2, pts[3 * (size_t)i + 2]);
  atomicAdd(&G.count[((size_t)cz * G.gy + cy) * G.gx + cx], 1u);
}
// one workgroup: exclusive scan of the cell counts (in place, count[cells] = N), cursors = starts
__global__ void __launch_bounds__(1024) d2_scan_kernel(int cells, D2Grid G) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < cells; base += 1024) {
    const int t = base + threadIdx.x;
    const uint32_t v = t < cells ? G.count[t] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (t < cells) G.count[t] = off + inc - v, G.cursor[t] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) G.count[cells] = carry_s;
}
__global__ void __launch_bounds__(256) d2_scatter_kernel(int N, const float *__restrict__ pts, D2Grid G) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const D2Map m = d2_map(G);
  const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  const int cx = d2_cell_axis(m, 0, x), cy = d2_cell_axis(m, 1, y), cz = d2_cell_axis(m, 2, z);
  const uint32_t pos = atomicAdd(&G.cursor[((size_t)cz * G.gy + cy) * G.gx + cx], 1u);
  G.sorted[pos] = make_float4(x, y, z, __uint_as_float((uint32_t)i));
}
__global__ void __launch_bounds__(256) d2_search_kernel(int N, D2Grid G, float *__restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const D2Map m = d2_map(G);
  const float4 q = G.sorted[j];
  const float qp[3] = {q.x, q.y, q.z};
  int c[3];
  float face[3];  // distance from the point to the nearer face of its own cell, minus a rounding margin
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    c[a] = d2_cell_axis(m, a, qp[a]);
    // (origin subtracted FIRST, as the cell assignment does: o + c * cs rounds at ulp(|o|), which for a cloud far from the
    // origin relative to its extent would eat the margin below; the margin also covers ulp(|q|) of the subtraction itself)
    const float f = (qp[a] - m.o[a]) - (float)c[a] * m.cs[a];
    const float margin = 1e-5f * m.cs[a] * (float)m.g[a] + 4.0f * 1.1920929e-7f * fmaxf(fabsf(qp[a]), fabsf(m.o[a]));
    face[a] = fmaxf(0.0f, fminf(f, m.cs[a] - f) - margin);
  }
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  for (int r = 1;; ++r) {
    const int lo[3] = {max(c[0] - r, 0), max(c[1] - r, 0), max(c[2] - r, 0)};
    const int hi[3] = {min(c[0] + r, m.g[0] - 1), min(c[1] + r, m.g[1] - 1), min(c[2] + r, m.g[2] - 1)};
    for (int z = lo[2]; z <= hi[2]; ++z)
      for (int y = lo[1]; y <= hi[1]; ++y) {
        const bool shell_zy = r == 1 || abs(z - c[2]) == r || abs(y - c[1]) == r;
        for (int x = lo[0]; x <= hi[0]; ++x) {
          if (!(shell_zy || abs(x - c[0]) == r)) continue;  // (interior cells were scanned by the smaller rings)
          const size_t cell = ((size_t)z * G.gy + y) * G.gx + x;
          const uint32_t e0 = G.count[cell], e1 = G.count[cell + 1];
          for (uint32_t e = e0; e < e1; ++e) {
            const float4 p = G.sorted[e];
            const float dx = qp[0] - p.x, dy = qp[1] - p.y, dz = qp[2] - p.z;
            float d2 = dx * dx + dy * dy + dz * dz;
            if (__float_as_uint(p.w) == __float_as_uint(q.w)) d2 = INFINITY;  // the point itself (by index)
            const float n0 = fminf(b0, d2), r0 = fmaxf(b0, d2);
            const float n1 = fminf(b1, r0), r1 = fmaxf(b1, r0);
            b0 = n0, b1 = n1, b2 = fminf(b2, r1);
          }
        }
      }
    // anything not yet seen lies beyond a face of the scanned block on some axis the block does not cover entirely
    bool all = true;
    float bound = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (c[a] - r < 0 && c[a] + r > m.g[a] - 1) continue;  // this axis is covered
      all = false;
      bound = fminf(bound, (float)(r - 1) * m.cs[a] + face[a] + (c[a] - r >= 0 && c[a] + r <= m.g[a] - 1 ? m.cs[a] : 0.0f));
    }
    // (an axis covered on one side only: the open side's face is at least (r - 1) cells + the own-cell face away; the
    // bound above uses that weaker figure there, and r cells + the own-cell face where both sides are open)
    if (all || b2 <= bound * bound * 0.99999f) break;
  }
  out[__float_as_uint(q.w)] = (b0 + b1 + b2) / 3.0f;
}

static int d2_cells_per_axis(int N) {
  int g = 1;
  while ((long)(g + 1) * (g + 1) * (g + 1) * 4 <= (long)N && g < 160) ++g;
  return g;
}
struct D2Layout {
  size_t bbox, count, cursor, sorted, bytes;
  int g;
  long cells;
  explicit D2Layout(int N) {
    g = d2_cells_per_axis(N > 0 ? N : 1);
    cells = (long)g * g * g;
    size_t o = 0;
    bbox = o, o = align_up(o + 8 * sizeof(uint32_t));
    count = o, o = align_up(o + (size_t)(cells + 1) * sizeof(uint32_t));
    cursor = o, o = align_up(o + (size_t)cells * sizeof(uint32_t));
    sorted = o, o = align_up(o + (size_t)(N > 0 ? N : 1) * sizeof(float4));
    bytes = o;
  }
};

}  // namespace dimo

using namespace dimo;

extern "C" size_t dimo_dist2_workspace_bytes(int N) { return D2Layout(N).bytes; }

extern "C" int dimo_dist2_grid(int N, const float *points, float *out, void *workspace, size_t workspace_bytes,
                               void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0) return DIMO_E_ARG;
  if (N == 0) return DIMO_OK;
  if (!points || !out || !workspace) return DIMO_E_ARG;
  const D2Layout L(N);
  if (workspace_bytes < L.bytes) return DIMO_E_WORKSPACE;
  D2Grid G;
  G.gx = G.gy = G.gz = L.g;
  G.bbox = at<uint32_t>(workspace, L.bbox), G.count = at<uint32_t>(workspace, L.count);
  G.cursor = at<uint32_t>(workspace, L.cursor), G.sorted = at<float4>(workspace, L.sorted);
  ScopedTimer tm(T_DIST2, stream);
  // bbox: min words all ones, max words zero; cell counts zero
  if (hipMemsetAsync(G.bbox, 0xff, 3 * sizeof(uint32_t), stream) != hipSuccess ||
      hipMemsetAsync(G.bbox + 3, 0, 3 * sizeof(uint32_t), stream) != hipSuccess ||
      hipMemsetAsync(G.count, 0, (size_t)(L.cells + 1) * sizeof(uint32_t), stream) != hipSuccess)
    return DIMO_E_LAUNCH;
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(d2_bbox_kernel, dim3(nb < 1024u ? nb : 1024u), dim3(256), 0, stream, N, points, G.bbox);
  hipLaunchKernelGGL(d2_count_kernel, dim3(nb), dim3(256), 0, stream, N, points, G);
  hipLaunchKernelGGL(d2_scan_kernel, dim3(1), dim3(1024), 0, stream, (int)L.cells, G);
  hipLaunchKernelGGL(d2_scatter_kernel, dim3(nb), dim3(256), 0, stream, N, points, G);
  hipLaunchKernelGGL(d2_search_kernel, dim3(nb), dim3(256), 0, stream, N, G, out);
  return check_launch();
}

extern "C" int dimo_knn(int M, int N, int k, const float *ref, const float *query, float *dist, int64_t *idx,
                        void *stream_) {
  return dimo_knn_seeded(M, N, k, ref, query, dist, idx, nullptr, stream_);
}

extern "C" int dimo_knn_seeded(int M, int N, int k, const float *ref, const float *query, float *dist, int64_t *idx,
                               const int64_t *seed_idx, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (M < 0 || N < 0 || k < 1 || k > 16) return DIMO_E_ARG;
  if (N == 0) return DIMO_OK;
  if (!query || !dist || !idx || (M > 0 && !ref)) return DIMO_E_ARG;
  const dim3 grid((N + KNN_BLOCK - 1) / KNN_BLOCK), block(KNN_BLOCK);
  ScopedTimer tm(T_KNN, stream);
  if (k <= 4) {
    // at most three workgroups (12 waves) per CU: see knn4_kernel
    const int cap = 768;
    const dim3 grid4((unsigned)((int)grid.x < cap || cap <= 0 ? (int)grid.x : cap));
    hipLaunchKernelGGL(knn4_kernel, grid4, dim3(KNN_BLOCK * KNN4_WAVES), 0, stream, M, N, k, ref, query, dist, idx,
                       seed_idx);
  }
  else if (k <= 8)
    hipLaunchKernelGGL(knn_kernel<8>, grid, block, 0, stream, M, N, k, ref, query, dist, idx);
  else
    hipLaunchKernelGGL(knn_kernel<16>, grid, block, 0, stream, M, N, k, ref, query, dist, idx);
  return check_launch();
}

extern "C" int dimo_dist2(int N, const float *points, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0) return DIMO_E_ARG;
  if (N == 0) return DIMO_OK;
  if (!points || !out) return DIMO_E_ARG;
  ScopedTimer tm(T_DIST2, stream);
  hipLaunchKernelGGL(dist2_kernel, dim3((N + D2_BLOCK - 1) / D2_BLOCK), dim3(D2_BLOCK), 0, stream, N, points, out);
  return check_launch();
}
