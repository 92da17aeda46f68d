// Fused backward tail of a deformation group: projection backward of every view of the group -> skinning backward,
// ONE kernel (the autograd of renderer/latent_gs_renderer.py:1255-1266 followed by that of
// :1191-1219).
//
// Before: `preprocess_bwd_batched` wrote 17 gradient floats per (render, Gaussian), `lbs_bwd_batched` read them back
// -- the leader's and every other member's of its group --, summed them and ran the skinning backward: 68 B written and
// 44 B re-read per (render, Gaussian) for values that live a few microseconds, and two launches.  Here a thread owns a
// Gaussian of one (motion, frame) group: it sums the instance records of each of the group's views, runs the projection
// backward on them, keeps the sum of the 14 per-Gaussian gradients (position 3, rotation 4, scale 3, opacity 1, colour 3)
// in registers across the views -- the skinned position / rotation / scale are the group's, loaded once --, runs the
// skinning backward ONCE on the sums, and writes the canonical gradients to the group leader's buffers (the fold,
// `accumulate_batched_kernel`, adds the groups in a fixed order: the per-Gaussian head of the bucket stays
// bit-reproducible) and the control-point contributions to its workgroup's LDS table as the stand-alone skinning backward
// does.  Same order of additions as the two-kernel path (views ascending), same maths (proj_math.hpp, deform_body.hpp).
#include "deform_body.hpp"
#include "proj_math.hpp"

namespace dimo {

// many-instance Gaussians a workgroup sums with whole waves per (iteration, view); beyond: by their own threads
constexpr int TAIL_BIG_MAX = 64;
constexpr int TAIL_MAX_DYNAMIC_LDS = 160 * 1024 - 4096;  // the rest is the kernel's static LDS (the wave-sum table)

// scalar base + 32-bit byte offset per lane (global_load / global_store with an SGPR base): a 64-bit address per lane
// and array cost a register PAIR each, held across the views' loops -- the first build spilled 40 registers of them
template <class T>
__device__ __forceinline__ T ldg(const void *base, unsigned off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + off);
}
template <class T>
__device__ __forceinline__ void stg(void *base, unsigned off, T v) {
  *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off) = v;
}

template <bool LOCAL_FRAME>
__global__ void __launch_bounds__(DEF_BLOCK, 3) tail_bwd_batched_kernel(int N, int M, int H, int W, uint32_t R_cap,
                                                                     GaussIO g, float scale_mod, const float *c_xyz,
                                                                     const float *c_lr, GeomLayout L, size_t flag_offset,
                                                                     RenderBatch b, float *__restrict__ partials_all) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ uint32_t s_big_n;
  __shared__ uint32_t s_big_lo[TAIL_BIG_MAX], s_big_hi[TAIL_BIG_MAX];
  __shared__ float s_big_sum[TAIL_BIG_MAX][13];
  float *s_cp = smem;                    // control-point table
  float *s_acc = smem + M * CP_STRIDE;   // control-point gradient accumulators
  const int group = blockIdx.y;
  const int lead = b.leader[group];
  const unsigned members = b.members[group];
  const dimo_render_desc &rl = b.r[lead];
  load_ctrl_to_lds(CtrlTable{c_xyz, c_lr, rl.d_xyz, rl.d_rot}, M, s_cp);
  for (int j = threadIdx.x; j < M * CP_STRIDE; j += blockDim.x) s_acc[j] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // whole workgroups stay in the loop (barriers inside; the control-point scatter combines the lanes of a wave); lanes
  // past N compute on the last Gaussian and contribute / store nothing
  for (int base = blockIdx.x * DEF_BLOCK; base < N; base += gridDim.x * DEF_BLOCK) {
    const bool valid = base + (int)threadIdx.x < N;
    const int i = valid ? base + (int)threadIdx.x : N - 1;
    // ---- the skinned Gaussian (projection backward: the group's views share it)
    const unsigned o1 = 4u * (unsigned)i, o3 = 12u * (unsigned)i, o4 = 16u * (unsigned)i;
    const float p[3] = {ldg<float>(rl.pts, o3), ldg<float>(rl.pts, o3 + 4u), ldg<float>(rl.pts, o3 + 8u)};
    const float4 qs4 = ldg<float4>(rl.rot, o4);
    const float qs[4] = {qs4.x, qs4.y, qs4.z, qs4.w};
    const float ss[3] = {ldg<float>(rl.scales, o3), ldg<float>(rl.scales, o3 + 4u), ldg<float>(rl.scales, o3 + 8u)};

    float gp[3] = {0.f, 0.f, 0.f}, gsc[3] = {0.f, 0.f, 0.f}, gsh[3] = {0.f, 0.f, 0.f}, gopac = 0.f;
    float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned mm = members; mm != 0u; mm &= mm - 1u) {  // the group's views, ascending (workgroup-uniform)
      const dimo_render_desc &r = b.r[__ffs(mm) - 1];
      const Splat *__restrict__ splat = at<Splat>(r.geom, L.splat);
      const uint32_t *__restrict__ offsets = at<uint32_t>(r.geom, L.offsets);
      const uint8_t *__restrict__ flags = at<uint8_t>(r.geom, L.flags);
      const SplatGrad *__restrict__ inst_grad = reinterpret_cast<const SplatGrad *>(r.bwd_scratch);
      const uint8_t *__restrict__ inst_flag = at<uint8_t>(r.bwd_scratch, flag_offset);
      // camera: read through the CONSTANT address space -- the matrices are not written during a step, and only that
      // promise lets the compiler fetch them with scalar loads into SGPRs (as plain global pointers, next to this
      // kernel's stores, they came through the vector memory path into 35 VGPRs)
      typedef const float __attribute__((address_space(4))) *ConstF;
      const ConstF Vg = (ConstF)(uintptr_t)r.view, Pg = (ConstF)(uintptr_t)r.proj, camg = (ConstF)(uintptr_t)r.campos;
      float V[16], P[16], cam[3];
#pragma unroll
      for (int k = 0; k < 16; ++k) V[k] = Vg[k], P[k] = Pg[k];
      cam[0] = camg[0], cam[1] = camg[1], cam[2] = camg[2];
      const bool visible = valid && ldg<int32_t>(r.radii, o1) > 0;
      uint32_t lo = ldg<uint32_t>(offsets, i == 0 ? 0u : o1 - 4u), hi = ldg<uint32_t>(offsets, o1);
      lo = i == 0 ? 0u : lo;
      lo = min(lo, R_cap), hi = min(hi, R_cap);
      Splat sp;
      {
        const float4 s0 = ldg<float4>(splat, 64u * (unsigned)i), s1 = ldg<float4>(splat, 64u * (unsigned)i + 16u);
        sp.x = s0.x, sp.y = s0.y, sp.A = s0.z, sp.B = s0.w, sp.C = s1.x, sp.opacity = s1.y;  // (all the backward reads)
      }
      const uint8_t clamp_fl = ldg<uint8_t>(flags, (unsigned)i);
      // a Gaussian with MANY instances is summed by a whole wave first (preprocess.hip: preprocess_bwd_body)
      __syncthreads();  // (the previous view's s_big_sum has been read)
      if (threadIdx.x == 0) s_big_n = 0u;
      __syncthreads();
      bool big = visible && hi > lo && hi - lo > 64u;
      uint32_t big_slot = 0;
      if (big) {
        big_slot = atomicAdd(&s_big_n, 1u);
        if (big_slot < (uint32_t)TAIL_BIG_MAX) s_big_lo[big_slot] = lo, s_big_hi[big_slot] = hi;
        else big = false;  // (more than the table holds: by its own thread, below)
      }
      __syncthreads();
      const uint32_t n_big = min(s_big_n, (uint32_t)TAIL_BIG_MAX);  // (workgroup-uniform)
      for (uint32_t bb = (uint32_t)wave; bb < n_big; bb += DEF_BLOCK / 64) {
        float a[13];
        wave_sum_instance_records(inst_grad, inst_flag, s_big_lo[bb], s_big_hi[bb], lane, a);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 13; ++k) s_big_sum[bb][k] = a[k];
        }
      }
      __syncthreads();
      ProjGrad pg = {};
      float dfeat[NFEAT] = {0, 0, 0, 0, 0, 0, 0};
      if (visible) {
        float m0 = 0, mx = 0, my = 0, mxx = 0, mxy = 0, myy = 0;
        if (big) {
          const float *a = s_big_sum[big_slot];
          m0 = a[0], mx = a[1], my = a[2], mxx = a[3], mxy = a[4], myy = a[5];
#pragma unroll
          for (int k = 0; k < NFEAT; ++k) dfeat[k] = a[6 + k];
          lo = hi;  // (nothing left for the loop below)
        }
        sum_instance_records(inst_grad, inst_flag, lo, hi, m0, mx, my, mxx, mxy, myy, dfeat);
        proj_backward_math(i, W, H, r.tanfovx, r.tanfovy, scale_mod, sp, p, qs, ss, nullptr, V, P, cam, m0, mx, my, mxx,
                           mxy, myy, dfeat, pg);
      }
      if (valid && r.g_means2D) {  // screen-space gradient of THIS view (densification statistics / the drop-in surface)
        stg<float>(r.g_means2D, o3, pg.dm2d[0]), stg<float>(r.g_means2D, o3 + 4u, pg.dm2d[1]);
        stg<float>(r.g_means2D, o3 + 8u, 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        gp[k] += pg.dmean[k], gsc[k] += pg.dsc[k];
        // degree-0 colour: rgb = max(SH_C0 f_dc + 0.5, 0) -- the forward left the clamp flags
        gsh[k] += SH_C0 * (((clamp_fl >> k) & 1) ? 0.0f : dfeat[k]);
      }
      go.x += pg.dq[0], go.y += pg.dq[1], go.z += pg.dq[2], go.w += pg.dq[3];
      gopac += pg.dop;
    }

    // ---- skinning backward on the sums (deform_body.hpp).  Its canonical inputs are requested only now: held across
    // the views' gather loops they cost 19 registers of a kernel that is short of them (the other waves of the SIMD
    // cover the round trip)
    const float x0 = ldg<float>(g.xyz, o3), x1 = ldg<float>(g.xyz, o3 + 4u), x2 = ldg<float>(g.xyz, o3 + 8u);
    const float4 q0 = ldg<float4>(g.rot, o4);
    const float4 dd = ldg<float4>(g.nn_dist, o4);
    int idx[DEF_K];
#pragma unroll
    for (int k = 0; k < DEF_K; ++k) idx[k] = (int)ldg<uint32_t>(g.nn_idx, 32u * (unsigned)i + 8u * k);  // (low words)
    const float op_raw = ldg<float>(g.opacity, o1);
    const float sc_raw[3] = {ldg<float>(g.scaling, o3), ldg<float>(g.scaling, o3 + 4u), ldg<float>(g.scaling, o3 + 8u)};
    float4 d_rot;
    float dxs[3];
    lbs_bwd_math<LOCAL_FRAME>(s_cp, s_acc, x0, x1, x2, q0, dd, idx, go, gp, valid, lane, d_rot, dxs);
    if (valid) {
      stg<float4>(rl.g_rot, o4, d_rot);
      const float o = 1.0f / (1.0f + __expf(-op_raw));
      stg<float>(rl.g_opac, o1, gopac * o * (1.0f - o));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        stg<float>(rl.g_means3D, o3 + 4u * c, dxs[c]);
        stg<float>(rl.g_scales, o3 + 4u * c, gsc[c] * __expf(sc_raw[c]));
        stg<float>(rl.g_shs, o3 + 4u * c, gsh[c]);
      }
    }
  }
  __syncthreads();
  if (LOCAL_FRAME) {  // columns 0..2 from the summed columns 4..6
    lbs_ctrl_position_grad(M, s_cp, s_acc);
    __syncthreads();
  }
  float *dst = partials_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * M * CP_STRIDE;
  for (int j = threadIdx.x; j < M * CP_STRIDE; j += blockDim.x) dst[j] = s_acc[j];
}

static void allow_big_lds_tail() {
  static const bool once = [] {
    // (dynamic + static LDS together must fit the 160 KB of a CU: asking for all of it on top of the kernel's own
    // 3.8 KB is refused with "invalid argument", which the next launch check then reports)
    const int lim = TAIL_MAX_DYNAMIC_LDS;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(tail_bwd_batched_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(tail_bwd_batched_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    clear_errors();
    return true;
  }();
  (void)once;
}

// The fused projection + skinning backward of the deformation groups of one launch (stage s2).  Writes one partial
// control-point table per workgroup at `partials` ([n_groups][grid][M][CP_STRIDE]); returns the grid through `grid_out`
// for the reduction that follows (deform.hip: lbs_backward_batched).
int tail_backward_batched(const dimo_step_common &c, const RenderBatch &b, int grid, float *partials,
                          hipStream_t stream) {
  GeomLayout L(c.N);
  const uint32_t cap = (uint32_t)(c.R_cap > 0xffffffffLL ? 0xffffffffu : (uint32_t)c.R_cap);
  const size_t flag_offset = align_up((size_t)(c.R_cap > 0 ? c.R_cap : 1) * sizeof(SplatGrad));
  GaussIO g{c.xyz, c.rotation, c.scaling, c.opacity, c.nn_dist, c.nn_idx};
  const size_t lds = 2 * (size_t)c.M * CP_STRIDE * sizeof(float);
  if (lds > (size_t)TAIL_MAX_DYNAMIC_LDS) return DIMO_E_ARG;  // (fused_tail() keeps such a model on the two-kernel path)
  allow_big_lds_tail();
  if (c.local_frame)
    hipLaunchKernelGGL(tail_bwd_batched_kernel<true>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M, c.H,
                       c.W, cap, g, c.scale_modifier, c.c_xyz, c.c_log_radius, L, flag_offset, b, partials);
  else
    hipLaunchKernelGGL(tail_bwd_batched_kernel<false>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M, c.H,
                       c.W, cap, g, c.scale_modifier, c.c_xyz, c.c_log_radius, L, flag_offset, b, partials);
  return check_launch();
}

}  // namespace dimo
