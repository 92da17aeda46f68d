// Adam over the flat parameter / gradient buckets in ONE launch (main_train_dimo.py:416-417,
// renderer/latent_gs_renderer.py:460-476: torch.optim.Adam, 12 parameter groups, eps = 1e-15).
//
// PyTorch's fused Adam issues one multi-tensor launch per parameter group (10 launches of ~36 us for 2 M
// parameters: launch-bound).  Every trainable tensor here is a view into one flat fp32 buffer, so the whole
// update is a single streaming pass: read p, g, m, v / write p, m, v (+ g = 0): 32 bytes per parameter,
// HBM-bound.  Per-group learning rates come from a small by-value segment table.  An optional device flag
// (the rasterizer's "instance capacity overflow" word) turns the update into a no-op, so a step whose
// gradients are invalid is skipped without the host ever waiting for the device.  The bias corrections use the
// number of updates actually APPLIED: the count of skipped launches lives on the device (two words, written
// alternately so that no block reads the word another block is writing) and is subtracted from the host's launch
// counter in the kernel -- every data-parallel replica sees the same all-reduced flag, so the replicas' effective
// step stays equal without the host ever reading the flag.
#include "common.hpp"

namespace dimo {

constexpr int ADAM_MAX_SEG = 32;
struct AdamSegs {
  int n;
  long long end[ADAM_MAX_SEG];  // exclusive end offset of segment k (segments are contiguous from 0)
  float lr[ADAM_MAX_SEG];
};

__global__ void __launch_bounds__(256) flat_adam_kernel(long long n, float *__restrict__ p, float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, AdamSegs segs,
                                                        float beta1, float beta2, float eps, long long launch,
                                                        const int *__restrict__ skip_flags, int n_flags,
                                                        int flag_stride, int zero_grad, int *__restrict__ skipped,
                                                        const uint32_t *__restrict__ report_src, int report_words,
                                                        uint32_t *report_dst, uint32_t report_seq,
                                                        float *__restrict__ zero_extra, long long zero_n,
                                                        long long begin, int final_part) {
  // Optional report for the host: `report_words` device words (the step's (R, overflow) instance counts) copied into
  // host-visible pinned memory behind a sequence number -- the host looks at them a step later without an event or a
  // copy engine (a 4 us device-to-host copy and the ~13 us gap behind it sat on every step's critical path).
  if (report_dst && final_part && blockIdx.x == 0) {
    for (int k = threadIdx.x; k < report_words; k += 256) report_dst[1 + k] = report_src[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(report_dst, report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // Optional scratch region zeroed for the NEXT step (its accumulators: a fill launch at the head of every step)
  if (final_part)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long long)gridDim.x * 256)
      zero_extra[i] = 0.f;
  bool skip = false;
  for (int k = 0; k < n_flags; ++k) skip |= skip_flags[(size_t)k * flag_stride] != 0;
  // effective step = launches so far - launches skipped before this one; this launch reads word (launch & 1) and
  // writes the other one
  __shared__ float s_bc[2];
  const int before = skipped ? skipped[launch & 1] : 0;
  if (threadIdx.x == 0) {
    const double t = (double)(launch - before);
    s_bc[0] = (float)(1.0 / (1.0 - pow((double)beta1, t)));
    s_bc[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
    // (a step taken as two launches -- [0, split) early, [split, n) later -- counts once: the final part writes)
    if (skipped && final_part && blockIdx.x == 0) skipped[(launch + 1) & 1] = before + (skip ? 1 : 0);
  }
  __syncthreads();
  const float inv_bc1 = s_bc[0], inv_sqrt_bc2 = s_bc[1];
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i0 = begin + ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
    if (i0 + 3 < n) {
      float4 gp = *reinterpret_cast<float4 *>(g + i0);
      if (!skip) {
        float4 pp = *reinterpret_cast<float4 *>(p + i0), mm = *reinterpret_cast<float4 *>(m + i0);
        float4 vv = *reinterpret_cast<float4 *>(v + i0);
        float *P = &pp.x, *G = &gp.x, *Mm = &mm.x, *V = &vv.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int s = 0;
          while (s + 1 < segs.n && i0 + c >= segs.end[s]) ++s;
          const float lr = segs.lr[s];
          Mm[c] = Mm[c] + (G[c] - Mm[c]) * (1.0f - beta1);
          V[c] = V[c] * beta2 + (1.0f - beta2) * G[c] * G[c];
          const float denom = sqrtf(V[c]) * inv_sqrt_bc2 + eps;
          P[c] = P[c] - (lr * inv_bc1) * (Mm[c] / denom);
        }
        *reinterpret_cast<float4 *>(p + i0) = pp;
        *reinterpret_cast<float4 *>(m + i0) = mm;
        *reinterpret_cast<float4 *>(v + i0) = vv;
      }
      if (zero_grad) *reinterpret_cast<float4 *>(g + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long i = i0; i < n; ++i) {
        if (!skip) {
          int s = 0;
          while (s + 1 < segs.n && i >= segs.end[s]) ++s;
          const float gi = g[i];
          const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
          const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
          m[i] = mi, v[i] = vi;
          p[i] = p[i] - (segs.lr[s] * inv_bc1) * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
        }
        if (zero_grad) g[i] = 0.f;
      }
    }
  }
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_flat_adam_step(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                                   int n_segments, const int64_t *segment_end_host, const float *segment_lr_host,
                                   float beta1, float beta2, float eps, int64_t step, const int *skip_flags,
                                   int n_flags, int flag_stride, int zero_grad, int *skipped_launches,
                                   const uint32_t *report_src, int report_words, uint32_t *report_dst_host,
                                   uint32_t report_seq, float *zero_extra, int64_t zero_n, int64_t range_begin,
                                   int64_t range_end, int final_part, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (n < 0 || n_segments < 1 || n_segments > ADAM_MAX_SEG || step < 1 || n_flags < 0) return DIMO_E_ARG;
  if (n == 0) return DIMO_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !segment_end_host || !segment_lr_host) return DIMO_E_ARG;
  if (n_flags > 0 && !skip_flags) return DIMO_E_ARG;
  if (report_words < 0 || zero_n < 0 || (report_dst_host && report_words > 0 && !report_src) || (zero_n > 0 && !zero_extra))
    return DIMO_E_ARG;
  if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
       reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
    return DIMO_E_ARG;  // float4 path needs 16-byte aligned buckets
  AdamSegs segs;
  segs.n = n_segments;
  for (int k = 0; k < ADAM_MAX_SEG; ++k) {
    segs.end[k] = k < n_segments ? (long long)segment_end_host[k] : (long long)n;
    segs.lr[k] = k < n_segments ? segment_lr_host[k] : 0.0f;
  }
  if (segs.end[n_segments - 1] != n) return DIMO_E_ARG;
  // one step as two launches: [range_begin, range_end) of the bucket (multiples of 4; 0, 0 = everything)
  if (range_begin == 0 && range_end == 0) range_end = n;
  if (range_begin < 0 || range_end > n || range_begin > range_end || (range_begin & 3) ||
      (range_end != n && (range_end & 3)))
    return DIMO_E_ARG;
  long long blocks = ((range_end - range_begin) / 4 + 255) / 256;
  // (the early part of a two-launch step runs on a private stream next to the TimeNet backward: it leaves room)
  const long long cap = final_part || (range_begin == 0 && range_end == n) ? 2048 : 768;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  ScopedTimer tm(T_ADAM, stream);
  hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (long long)range_end, params, grads,
                     exp_avg, exp_avg_sq, segs, beta1, beta2, eps, (long long)step, skip_flags, n_flags, flag_stride,
                     zero_grad, skipped_launches, report_src, report_words, report_dst_host, report_seq, zero_extra,
                     (long long)zero_n, (long long)range_begin, final_part);
  return check_launch();
}
