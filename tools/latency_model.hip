// Cost model of SHORT kernels on MI355X (the binning / scan / KNN class: a few thousand workgroups, a few dependent
// memory round trips each).  hipcc --offload-arch=gfx950 -O3 tools/latency_model.hip -o /tmp/latency_model && /tmp/latency_model
// Prints the mean duration (HIP events around 20 back-to-back launches, divided by 20) of:
//   empty kernels at several grids; chains of 1 / 2 / 4 / 6 dependent global loads; the same with a 2.3 KB by-value
//   argument indexed by blockIdx.y; returning global atomics on a small table; 64-way same-address LDS atomics;
//   __syncthreads vs an LDS-only barrier with a load in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { void *p[8][32]; };  // 2 KB by value, like RenderBatch

__global__ void __launch_bounds__(256) k_empty(uint32_t *out) {
  if (threadIdx.x == 1000) out[0] = 1;
}
// chain of DEP dependent loads: idx = tab[idx]; tab is a random permutation over `n` words (n * 4 bytes >> L2)
template <int DEP>
__global__ void __launch_bounds__(256) k_chain(const uint32_t *__restrict__ tab, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t idx = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
  idx = idx % n;
#pragma unroll
  for (int d = 0; d < DEP; ++d) idx = tab[idx];
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = idx;
}
// the same, coalesced and L2-resident: idx = small[(idx + 1) & 1023] ... (dependent, but always a hit)
template <int DEP>
__global__ void __launch_bounds__(256) k_chain_hit(const uint32_t *__restrict__ small, uint32_t *__restrict__ out) {
  uint32_t idx = threadIdx.x;
#pragma unroll
  for (int d = 0; d < DEP; ++d) idx = small[(idx + d) & 1023u] & 1023u;
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = idx;
}
template <int DEP>
__global__ void __launch_bounds__(256) k_chain_big(Big b, uint32_t n, uint32_t *__restrict__ out) {
  const uint32_t *tab = reinterpret_cast<const uint32_t *>(b.p[blockIdx.y][0]);
  uint32_t idx = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
  idx = idx % n;
#pragma unroll
  for (int d = 0; d < DEP; ++d) idx = tab[idx];
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = idx;
}
// A returning atomics per workgroup thread subset on a table of `words` counters per render
__global__ void __launch_bounds__(256) k_atomics(uint32_t *__restrict__ tab, int words, int per_wg, int returning,
                                                 uint32_t *__restrict__ out) {
  uint32_t *t = tab + (size_t)blockIdx.y * words;
  uint32_t v = 0;
  if ((int)threadIdx.x < per_wg) {
    const uint32_t a = (blockIdx.x * 37u + threadIdx.x * 11u) % (uint32_t)words;
    if (returning) v = atomicAdd(&t[a], 1u); else atomicAdd(&t[a], 1u);
  }
  if (v == 0xffffffffu) out[0] = v;
}
__global__ void __launch_bounds__(256) k_lds_atomic(int same, int iters, uint32_t *__restrict__ out) {
  __shared__ uint32_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) s[i] = 0;
  __syncthreads();
  for (int i = 0; i < iters; ++i) atomicAdd(&s[same ? (i & 7) : ((threadIdx.x * 17 + i) & 1023)], 1u);
  __syncthreads();
  if (s[threadIdx.x] == 0xffffffffu) out[0] = 1;
}
// a load in flight across N barriers: full __syncthreads (drains vmcnt) vs LDS-only barrier
template <bool LIGHT>
__global__ void __launch_bounds__(256) k_barrier(const uint32_t *__restrict__ tab, uint32_t n, int nbar,
                                                 uint32_t *__restrict__ out) {
  __shared__ uint32_t s[256];
  uint32_t idx = ((blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x) % n;
  uint32_t acc = 0;
  for (int i = 0; i < nbar; ++i) {
    const uint32_t pre = tab[(idx + i * 7919u) % n];  // prefetch for the next round
    s[threadIdx.x] = acc;
    if (LIGHT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    acc += s[(threadIdx.x + 1) & 255];
    if (LIGHT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    acc += pre;
  }
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

template <class F>
static float timeit(F f, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3f * ms / reps;
}

int main() {
  const uint32_t n = 64u << 20;  // 256 MB of table: beyond L2, inside the Infinity Cache; 1 << 28 words = 1 GB: HBM
  uint32_t *tab, *out, *small, *atab;
  CK(hipMalloc(&tab, (size_t)n * 4));
  CK(hipMalloc(&out, (size_t)4096 * 8 * 256 * 4));
  CK(hipMalloc(&small, 4096));
  CK(hipMalloc(&atab, 8 * 4096 * 4));
  std::vector<uint32_t> h(n);
  uint64_t x = 88172645463325252ull;
  for (uint32_t i = 0; i < n; ++i) {
    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
    h[i] = (uint32_t)(x % n);
  }
  CK(hipMemcpy(tab, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(small, 0, 4096));
  CK(hipMemset(atab, 0, 8 * 4096 * 4));
  Big big;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 32; ++j) big.p[i][j] = tab;
  const dim3 grids[] = {dim3(391, 1), dim3(391, 4), dim3(391, 8), dim3(256, 8), dim3(2048, 8)};
  for (const dim3 &g : grids) {
    printf("grid (%u,%u) x 256 threads = %u workgroups\n", g.x, g.y, g.x * g.y);
    printf("  empty                         %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, g, dim3(256), 0, 0, out); }));
    printf("  1 dependent miss load         %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<1>, g, dim3(256), 0, 0, tab, n, out); }));
    printf("  2 dependent miss loads        %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<2>, g, dim3(256), 0, 0, tab, n, out); }));
    printf("  4 dependent miss loads        %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<4>, g, dim3(256), 0, 0, tab, n, out); }));
    printf("  6 dependent miss loads        %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain<6>, g, dim3(256), 0, 0, tab, n, out); }));
    printf("  1 dependent L2-hit load       %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain_hit<1>, g, dim3(256), 0, 0, small, out); }));
    printf("  4 dependent L2-hit loads      %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain_hit<4>, g, dim3(256), 0, 0, small, out); }));
    printf("  8 dependent L2-hit loads      %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain_hit<8>, g, dim3(256), 0, 0, small, out); }));
    printf("  2 miss loads, 2 KB by-value   %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_chain_big<2>, g, dim3(256), 0, 0, big, n, out); }));
    for (int ret = 0; ret < 2; ++ret)
      for (int per : {8, 32, 256})
        printf("  %3d %s atomics/wg on 1024 words  %7.2f us\n", per, ret ? "returning" : "plain    ",
               timeit([&] { hipLaunchKernelGGL(k_atomics, g, dim3(256), 0, 0, atab, 1024, per, ret, out); }));
    printf("  LDS atomics x16 spread        %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds_atomic, g, dim3(256), 0, 0, 0, 16, out); }));
    printf("  LDS atomics x16 same address  %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds_atomic, g, dim3(256), 0, 0, 1, 16, out); }));
    printf("  4 rounds prefetch+__syncthreads %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_barrier<false>, g, dim3(256), 0, 0, tab, n, 4, out); }));
    printf("  4 rounds prefetch+lds barrier   %7.2f us\n", timeit([&] { hipLaunchKernelGGL(k_barrier<true>, g, dim3(256), 0, 0, tab, n, 4, out); }));
  }
  // back-to-back dependent launches of a short kernel: the boundary cost
  printf("10 x (391,8) 1-load kernels back to back: %7.2f us each\n",
         timeit([&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_chain<1>, dim3(391, 8), dim3(256), 0, 0, tab, n, out); }) / 10);
  return 0;
}
