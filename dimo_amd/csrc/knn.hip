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
__global__ void __launch_bounds__(KNN_BLOCK *KNN4_WAVES) knn4_kernel(int M, int N, int k, const float *__restrict__ ref,
                                                                      const float *__restrict__ query,
                                                                      float *__restrict__ dist, int64_t *__restrict__ idx,
                                                                      const int64_t *__restrict__ seed) {
  __shared__ float4 s_ref[KNN_CHUNK];
  __shared__ double s_best[KNN4_WAVES][4][KNN_BLOCK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * KNN_BLOCK + lane;
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

}  // namespace dimo

using namespace dimo;

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
  if (k <= 4)
    hipLaunchKernelGGL(knn4_kernel, grid, dim3(KNN_BLOCK * KNN4_WAVES), 0, stream, M, N, k, ref, query, dist, idx,
                       seed_idx);
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
