// Host build of the rasterizer -- dimo_amd/csrc/preprocess.hip, binning.hip and blend.hip, one translation unit each as
// on the GPU (tests/simt/build.py writes them) -- on the SIMT emulation shim.  TEST INFRASTRUCTURE ONLY: the sources'
// own C-ABI entry points (dimo_raster_preprocess_forward, dimo_raster_render_forward, dimo_raster_backward, ...) are
// exported as they are and take host pointers here; this file holds what api.hip holds for them in the product.
#include "common.hpp"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo

extern "C" {
// out: geom bytes, bin bytes, img bytes, backward scratch bytes; then the offsets the tests read back --
// geom: splat, rect, tiles, offsets, total; bin: vals, ranges; img: final_T, n_contrib, final_acc; bin capacity
void simt_raster_layout(int N, int H, int W, int64_t R_cap, size_t out[16]) {
  dimo::GeomLayout G(N);
  dimo::BinLayout B(R_cap, H, W, N);
  dimo::ImgLayout I(H, W);
  out[0] = G.bytes, out[1] = B.bytes, out[2] = I.bytes;
  out[3] = dimo::align_up(B.cap * sizeof(dimo::SplatGrad)) + dimo::align_up(B.cap);
  out[4] = G.splat, out[5] = G.rect, out[6] = G.tiles, out[7] = G.offsets, out[8] = G.total;
  out[9] = B.vals_b, out[10] = B.ranges;
  out[11] = I.final_T, out[12] = I.n_contrib, out[13] = I.final_acc;
  out[14] = B.cap, out[15] = sizeof(dimo::Splat);
}
}
