// Per-CU streaming rate of a SHARED, L2-resident buffer (the fused TimeNet's weight stream: every workgroup reads the
// same 0.25 MB per layer): one workgroup per CU, NW waves, each wave a private 1 KiB-per-load stream with DEPTH loads
// in flight.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH>
__global__ void __launch_bounds__(1024) stream_kernel(const float4 *__restrict__ w, int blocks_per_wave, int total_blocks,
                                                      int rot_shift, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // wave `wave` of every workgroup reads blocks wave * blocks_per_wave ... (the same addresses in every workgroup)
  const int rot = rot_shift >= 0 ? (blockIdx.x >> rot_shift) : 0;
  float4 q[DEPTH];
  float acc = 0.f;
  auto addr = [&](int b) { return w + ((size_t)((wave * blocks_per_wave + ((b + rot) % blocks_per_wave)) % total_blocks) * 64 + lane); };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) q[u] = *addr(u);
  for (int b = 0; b < blocks_per_wave; b += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const float4 v = q[u];
      q[u] = *addr(b + u + DEPTH);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 123.456f) out[0] = acc + nw;
}
template <int DEPTH>
static void run(const float4 *w, int nwg, int nw, int blocks_per_wave, int total_blocks, int rot_shift, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(nwg), dim3(64 * nw), 0, 0, w, blocks_per_wave, total_blocks, rot_shift, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it) best = ms < best ? ms : best;
  }
  const double bytes_per_wg = (double)nw * blocks_per_wave * 1024.0;
  printf("%3d workgroups x %2d waves, depth %2d, %s: %7.1f us  %6.1f GB/s per workgroup = %5.1f B/clk (2.4 GHz), %5.2f TB/s in all\n",
         nwg, nw, DEPTH, rot_shift >= 0 ? "rotated " : "in step", 1e3 * best, bytes_per_wg / (best * 1e-3) / 1e9,
         bytes_per_wg / (best * 1e-3) / 2.4e9, bytes_per_wg * nwg / (best * 1e-3) / 1e12);
}
int main() {
  const int total_blocks = 2816;  // 2.75 MiB of weights
  float4 *w;
  float *out;
  hipMalloc(&w, (size_t)total_blocks * 1024), hipMalloc(&out, 64);
  hipMemset(w, 0, (size_t)total_blocks * 1024);
  for (int nwg : {128, 256})
    for (int nw : {8, 16}) {
      const int bpw = 2816 / nw;  // every workgroup reads the whole buffer once
      run<4>(w, nwg, nw, bpw / 4 * 4, total_blocks, -1, out);
      run<8>(w, nwg, nw, bpw / 8 * 8, total_blocks, -1, out);
      run<16>(w, nwg, nw, bpw / 16 * 16, total_blocks, -1, out);
      run<8>(w, nwg, nw, bpw / 8 * 8, total_blocks, 3, out);
      run<16>(w, nwg, nw, bpw / 16 * 16, total_blocks, 3, out);
    }
  return 0;
}
