// Farthest point sampling (pytorch3d.ops.sample_farthest_points with random_start_point = False, as
// main_train_dimo.py:511-515 calls it): start at point 0, then K - 1 times take the point whose squared distance to
// the selected set is largest (lowest index among equals).  Inherently sequential in K; one workgroup of 1024
// threads keeps the running minimum distances in a caller-provided array (L2 resident) and finds each argmax with
// a wave butterfly + one LDS step.  Used once per FPS_iter (1000) steps of stage s1: ~3 us per selected point.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace dimo {

__global__ void __launch_bounds__(1024) fps_kernel(int N, int K, const float *__restrict__ xyz,
                                                   float *__restrict__ min_d, int64_t *__restrict__ out_idx) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_sel;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < N; i += 1024) min_d[i] = INFINITY;
  int sel = 0;
  if (t == 0) out_idx[0] = 0;
  for (int k = 1; k < K; ++k) {
    const float sx = xyz[3 * sel], sy = xyz[3 * sel + 1], sz = xyz[3 * sel + 2];
    float best = -1.0f;
    int best_i = 0x7fffffff;
    for (int i = t; i < N; i += 1024) {
      const float dx = xyz[3 * i] - sx, dy = xyz[3 * i + 1] - sy, dz = xyz[3 * i + 2] - sz;
      const float d = fminf(min_d[i], dx * dx + dy * dy + dz * dz);
      min_d[i] = d;
      if (d > best) best = d, best_i = i;  // ascending i per thread: the first maximum stays
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(best_i, o, 64);
      if (ov > best || (ov == best && oi < best_i)) best = ov, best_i = oi;
    }
    if (lane == 0) s_val[wave] = best, s_idx[wave] = best_i;
    __syncthreads();
    if (t == 0) {
      float b = s_val[0];
      int bi = s_idx[0];
      for (int w = 1; w < 16; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_idx[w] < bi)) b = s_val[w], bi = s_idx[w];
      s_sel = bi;
      out_idx[k] = bi;
    }
    __syncthreads();
    sel = s_sel;
  }
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_farthest_point_sample(int N, int K, const float *xyz, float *min_dist_scratch, int64_t *out_idx,
                                          void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0 || K < 0 || K > N) return DIMO_E_ARG;
  if (K == 0) return DIMO_OK;
  if (!xyz || !min_dist_scratch || !out_idx) return DIMO_E_ARG;
  hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), 0, stream, N, K, xyz, min_dist_scratch, out_idx);
  return check_launch();
}
