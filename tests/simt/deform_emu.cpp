// Host build of dimo_amd/csrc/deform.hip (linear-blend skinning forward / backward) and adam.hip (the flat Adam step) on
// the SIMT emulation shim.  TEST INFRASTRUCTURE ONLY: their C-ABI entry points are exported as they are and take host
// pointers here.
#include "deform_src.inc"
#include "adam_src.inc"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo

// wave_ops.hpp's 16-value wave reduction (the shim's spelling of its instruction sequence) on caller-supplied data:
// in = 64 lanes x 16 floats, out = 3 x 16 totals (all 16 values in use, 13, 10) -- the layout of the product's
// dimo_selftest_wave_reduce16 (blend.hip)
static void reduce16_probe_kernel(const float *in, float *out) {
  float v[16], w[16], u[16];
  for (int k = 0; k < 16; ++k) v[k] = w[k] = u[k] = in[threadIdx.x * 16 + k];
  const float t16 = dimo::wave_reduce16<16>(v), t13 = dimo::wave_reduce16<13>(w), t10 = dimo::wave_reduce16<10>(u);
  if ((threadIdx.x & 3) == 0) {
    const int s = dimo::reduce16_slot(threadIdx.x);
    out[s] = t16, out[16 + s] = t13, out[32 + s] = t10;
  }
}
// wave_scatter_add_match8 / wave_scatter_add<8> on one wave: table[idx[lane]][c] += v[lane][c] for the valid lanes
static void scatter_probe_kernel(float *table, int stride, const int *idx, const float *vals, const int *valid, int matched) {
  float v[8];
  const int lane = threadIdx.x;
  for (int k = 0; k < 8; ++k) v[k] = vals[lane * 8 + k];
  if (matched) dimo::wave_scatter_add_match8(table, stride, idx[lane], v, valid[lane] != 0, lane);
  else dimo::wave_scatter_add<8>(table, stride, idx[lane], v, valid[lane] != 0, lane);
}
extern "C" void simt_wave_reduce16(const float *in, float *out) {
  hipLaunchKernelGGL(reduce16_probe_kernel, dim3(1), dim3(64), 0, nullptr, in, out);
}
extern "C" void simt_wave_scatter(float *table, int stride, const int *idx, const float *vals, const int *valid, int matched) {
  hipLaunchKernelGGL(scatter_probe_kernel, dim3(1), dim3(64), 0, nullptr, table, stride, idx, vals, valid, matched);
}
