// Host build of dimo_amd/csrc/knn.hip (KNN, distCUDA2 brute force and grid form) and fps.hip (farthest point sampling)
// on the SIMT emulation shim.  TEST INFRASTRUCTURE ONLY: their C-ABI entry points (dimo_knn, dimo_knn_seeded,
// dimo_dist2, dimo_dist2_grid, dimo_farthest_point_sample) are exported as they are and take host pointers here.
#include "knn_src.inc"
#include "fps_src.inc"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo
