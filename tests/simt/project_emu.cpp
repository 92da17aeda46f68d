// Host build of dimo_amd/csrc/preprocess.hip (the projection kernel: forward and backward) on the SIMT emulation shim.
// TEST INFRASTRUCTURE ONLY: dimo_raster_preprocess_forward is exported as it is and takes host pointers here.
#include "preprocess_src.inc"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo

extern "C" {
// out: splat, rect, tiles, flags, total, key32, block_sums, bytes
void simt_project_layout(int N, size_t out[8]) {
  dimo::GeomLayout G(N);
  out[0] = G.splat, out[1] = G.rect, out[2] = G.tiles, out[3] = G.flags, out[4] = G.total, out[5] = G.key32;
  out[6] = G.block_sums, out[7] = G.bytes;
}
}
