// Host build of the native step executor (dimo_amd/csrc/executor.hip) over the batched kernels of deform.hip,
// preprocess.hip, binning.hip and blend.hip -- one translation unit each as on the GPU -- on the SIMT emulation shim.
// TEST INFRASTRUCTURE ONLY: dimo_executor_* are exported as they are and take host pointers; this file holds what
// api.hip holds for them in the product.
#include "common.hpp"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo

extern "C" {
// out: geom, bin, img, rasterizer backward scratch, skinning backward scratch (per render) bytes; offsets of vals,
// ranges in bin and of total in geom
size_t dimo_deform_backward_scratch_bytes(int N, int M);
void simt_step_layout(int N, int M, int H, int W, int64_t R_cap, size_t out[8]) {
  dimo::GeomLayout G(N);
  dimo::BinLayout B(R_cap, H, W, N);
  dimo::ImgLayout I(H, W);
  out[0] = G.bytes, out[1] = B.bytes, out[2] = I.bytes;
  out[3] = dimo::align_up(B.cap * sizeof(dimo::SplatGrad)) + dimo::align_up(B.cap);
  out[4] = dimo_deform_backward_scratch_bytes(N, M);
  out[5] = B.vals_b, out[6] = B.ranges, out[7] = G.total;
}
}

// what a loss kernel is to the executor: something on `stream` that reads a render's images and leaves its gradient
// images -- here two asynchronous copies (tests/test_executor_emulated.py)
extern "C" void simt_enqueue_copy(void *dst, const void *src, size_t n, void *stream) {
  (void)hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}
