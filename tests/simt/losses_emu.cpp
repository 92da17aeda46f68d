// Host build of the image-loss kernels -- dimo_amd/csrc/ssim.hip and image_loss.hip, one translation unit each as on the
// GPU -- on the SIMT emulation shim.  TEST INFRASTRUCTURE ONLY: the sources' own C-ABI entry points (dimo_ssim_forward,
// dimo_ssim_backward, dimo_ssim_forward_backward, dimo_image_loss, dimo_ssim_image_loss) are exported as they are and
// take host pointers here.
#include "common.hpp"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo
