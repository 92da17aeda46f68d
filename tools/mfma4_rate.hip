// Issue rate and dependent latency of v_mfma_f32_4x4x1_16b_f32 against v_mfma_f32_16x16x4_f32 on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma4_rate.hip -o /tmp/mfma4_rate && /tmp/mfma4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC, bool SMALL>
__global__ void __launch_bounds__(256) rate_kernel(float *out, int iters, float a0, float b0) {
  f32x4 c[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) {
      if (SMALL) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
      else c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACC, bool SMALL>
void run(const char *name, int waves_per_simd) {
  float *out;
  const int blocks = 256 * waves_per_simd;  // 4 waves per block = one per SIMD
  hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<ACC, SMALL>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<ACC, SMALL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * ACC * waves_per_simd;
  const double flop = (SMALL ? 512.0 : 2048.0) * iters * ACC * (double)blocks * 4;
  printf("%-28s acc=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cyc @2.4GHz), %.1f TFLOP/s\n", name, ACC,
         waves_per_simd, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<1, true>("4x4x1_16b dependent", 1);
  run<2, true>("4x4x1_16b", 1);
  run<4, true>("4x4x1_16b", 1);
  run<8, true>("4x4x1_16b", 1);
  run<4, true>("4x4x1_16b", 4);
  run<8, true>("4x4x1_16b", 4);
  run<1, false>("16x16x4 dependent", 1);
  run<2, false>("16x16x4", 1);
  run<2, false>("16x16x4", 4);
  return 0;
}
