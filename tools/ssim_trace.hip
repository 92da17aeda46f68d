// The fused SSIM kernel alone, with the phase trace of ssim.hip (DIMO_SSIM_TRACE) at several occupancies:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I include tools/ssim_trace.hip -o /tmp/ssim_trace && /tmp/ssim_trace
#define DIMO_SSIM_TRACE
#include "../dimo_amd/csrc/ssim.hip"
#include <cstdio>
#include <vector>
namespace dimo {  // what api.hip provides in the library
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int, hipStream_t) {}
ScopedTimer::~ScopedTimer() {}
}
static void run(const char *what, int B, float *a, float *b, float *coef, float *sum, float *g, int dyn_lds = 0, int grid_cap = 4096) {
  const int H = 512, W = 512, planes = 3 * B;
  const Window win = make_window();
  ImagePtrs ptrs{};
  ptrs.channels = 3;
  const long tiles = (long)(W / TS) * (H / TS) * planes;
  const dim3 grid((unsigned)(tiles < grid_cap ? tiles : grid_cap)), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9f, tot = 0;
  for (int it = 0; it < 25; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(ssim_fused_kernel, grid, block, dyn_lds, 0, H, W, planes, 1, win, a, b, ptrs, coef, 1e-6f, sum, g);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 5) tot += ms, best = ms < best ? ms : best;
  }
  printf("B=%d %-44s avg %.1f us  best %.1f us\n", B, what, 1e3f * tot / 20, 1e3f * best);
  // phase trace of one more launch
  const int nwg = grid.x;
  unsigned long long *buf;
  hipMalloc(&buf, (size_t)nwg * 16 * 8);
  hipMemset(buf, 0, (size_t)nwg * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(dimo::g_ssim_trace), &buf, sizeof buf);
  hipLaunchKernelGGL(ssim_fused_kernel, grid, block, dyn_lds, 0, H, W, planes, 1, win, a, b, ptrs, coef, 1e-6f, sum, g);
  hipDeviceSynchronize();
  std::vector<unsigned long long> t((size_t)nwg * 16);
  hipMemcpy(t.data(), buf, t.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long *null = nullptr;
  hipMemcpyToSymbol(HIP_SYMBOL(dimo::g_ssim_trace), &null, sizeof null);
  hipFree(buf);
  double ph[9] = {0};
  int cnt = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int w = 0; w < nwg; ++w) {
    if (!t[w * 16] || !t[w * 16 + 9]) continue;
    ++cnt;
    for (int k = 0; k < 9; ++k) ph[k] += (double)(t[w * 16 + k + 1] - t[w * 16 + k]) * 0.01;
    tmin = t[w * 16] < tmin ? t[w * 16] : tmin, tmax = t[w * 16 + 9] > tmax ? t[w * 16 + 9] : tmax;
  }
  static const char *names[9] = {"loads", "barrier", "vertical", "barrier", "horizontal+stats", "barrier", "planes", "second pass", "final"};
  printf("     first tile of %d workgroups (us):", cnt);
  for (int k = 0; k < 9; ++k) printf(" %s %.2f |", names[k], ph[k] / cnt);
  printf(" span %.1f us\n", (double)(tmax - tmin) * 0.01);
}
int main() {
  for (int B : {4, 8}) {
    const size_t n = (size_t)B * 3 * 512 * 512;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    float *a, *b, *g, *coef, *sum;
    hipMalloc(&a, n * 4), hipMalloc(&b, n * 4), hipMalloc(&g, n * 4), hipMalloc(&coef, 4), hipMalloc(&sum, 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < n; ++i) h[i] = 1.0f - h[i] * 0.5f;
    hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice);
    const float c = -0.2f;
    hipMemcpy(coef, &c, 4, hipMemcpyHostToDevice);
    hipMemset(sum, 0, 4);
    run("persistent, three workgroups per CU", B, a, b, coef, sum, g, 0, SSIM_GRID);
    run("one tile per workgroup", B, a, b, coef, sum, g, 0, 1 << 20);
    run("persistent, two workgroups per CU", B, a, b, coef, sum, g, 30 << 10, 512);
    run("persistent, one workgroup per CU", B, a, b, coef, sum, g, 90 << 10, 256);
    hipFree(a), hipFree(b), hipFree(g), hipFree(coef), hipFree(sum);
  }
  return 0;
}
