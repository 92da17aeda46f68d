// Latency of a cross-stream dependency: kernel K1 on stream A -> (signal) -> (wait) -> kernel K2 on stream B, measured
// on the device (s_memrealtime at K1's end and K2's start), for events and for stream memory operations.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/xstream_latency.hip -o /tmp/xstream_latency && /tmp/xstream_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void stamp_end(unsigned long long *t, int spin) {
  unsigned long long x = 0;
  for (int i = 0; i < spin; ++i) x += __builtin_amdgcn_s_memrealtime() & 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime() + (x & 0);
}
__global__ void stamp_start(unsigned long long *t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned long long *t;
  CK(hipHostMalloc((void **)&t, 64, hipHostMallocMapped));
  unsigned int *flag;
  CK(hipMalloc((void **)&flag, 64));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<double> lat;
    for (int it = 0; it < 40; ++it) {
      t[0] = t[1] = 0;
      if (mode == 0) {  // same stream: the plain kernel boundary
        hipLaunchKernelGGL(stamp_end, dim3(256), dim3(256), 0, a, t, 2000);
        hipLaunchKernelGGL(stamp_start, dim3(256), dim3(256), 0, a, t);
      } else if (mode == 1) {  // event record on A, wait on B
        hipLaunchKernelGGL(stamp_end, dim3(256), dim3(256), 0, a, t, 2000);
        CK(hipEventRecord(ev, a));
        CK(hipStreamWaitEvent(b, ev, 0));
        hipLaunchKernelGGL(stamp_start, dim3(256), dim3(256), 0, b, t);
      } else {  // stream memory operations
        CK(hipMemsetAsync(flag, 0, 4, a));
        CK(hipStreamSynchronize(a));
        hipLaunchKernelGGL(stamp_end, dim3(256), dim3(256), 0, a, t, 2000);
        CK(hipStreamWriteValue32(a, flag, (unsigned)(it + 1), 0));
        CK(hipStreamWaitValue32(b, flag, (unsigned)(it + 1), hipStreamWaitValueEq, 0xffffffffu));
        hipLaunchKernelGGL(stamp_start, dim3(256), dim3(256), 0, b, t);
      }
      CK(hipStreamSynchronize(a));
      CK(hipStreamSynchronize(b));
      if (it >= 5) lat.push_back((double)(long long)(t[1] - t[0]) * 0.01);
    }
    std::sort(lat.begin(), lat.end());
    printf("%-34s median %.1f us  min %.1f  max %.1f\n",
           mode == 0 ? "same stream" : mode == 1 ? "event record / stream wait event" : "stream write value / wait value",
           lat[lat.size() / 2], lat.front(), lat.back());
  }
  return 0;
}
