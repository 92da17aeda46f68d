// Host build of dimo_amd/csrc/timenet.hip (TimeNet forward, dgrad chain, weight gradients) on the SIMT emulation shim.
// TEST INFRASTRUCTURE ONLY: dimo_timenet_workspace_bytes / _forward / _backward are exported as they are and take host
// pointers here.
#include "common.hpp"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo
