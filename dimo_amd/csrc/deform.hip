// Fused skinning stage of Renderer.render (stage s2): LBS weights, per-neighbour rotation, blend of the
// k = 4 control-point transforms, quaternion product, normalisation and the exp / sigmoid activations --
// renderer/latent_gs_renderer.py:1187-1219 -- as ONE streaming kernel forward and one backward.
//
// The reference (and a literal PyTorch restatement) runs this as ~25 eager kernels per direction, including
// a batched 3x3 matmul over N*4 tiny matrices and index_put scatter-adds into the 512 control points
// (measured on MI355X: 15 ms of the 19 ms a render took).  Here every Gaussian is one thread; the control
// point table (M rows x 11 floats) lives in LDS for the gathers, and the backward accumulates the
// control-point gradients in per-workgroup LDS (ds_add_f32), writes one partial table per workgroup and a
// second tiny kernel sums the partials in a fixed order: no global atomics, deterministic.
#include "deform_body.hpp"

namespace dimo {

template <bool LOCAL_FRAME>
__device__ __forceinline__ void lbs_fwd_body(int N, int M, GaussIO g, CtrlTable t, float *__restrict__ out_xyz,
                                             float *__restrict__ out_rot, float *__restrict__ out_scales,
                                             float *__restrict__ out_opacity) {
  extern __shared__ __attribute__((aligned(16))) float s_cp[];
  load_ctrl_to_lds(t, M, s_cp);
  __syncthreads();
  for (int i = blockIdx.x * DEF_BLOCK + threadIdx.x; i < N; i += gridDim.x * DEF_BLOCK) {
    const float x0 = g.xyz[3 * i], x1 = g.xyz[3 * i + 1], x2 = g.xyz[3 * i + 2];
    const float4 q0 = *reinterpret_cast<const float4 *>(g.rot + 4 * (size_t)i);
    const float4 dd = *reinterpret_cast<const float4 *>(g.nn_dist + 4 * (size_t)i);
    const float dist[DEF_K] = {dd.x, dd.y, dd.z, dd.w};
    float wt[DEF_K], W = 0.f;
    int idx[DEF_K];
#pragma unroll
    for (int k = 0; k < DEF_K; ++k) {
      idx[k] = (int)g.nn_idx[4 * (size_t)i + k];
      const float r = s_cp[idx[k] * CP_STRIDE + 3];
      wt[k] = __expf(-1.0f * dist[k] * dist[k] / (2.0f * (r * r))) + LBS_EPS;
      W += wt[k];
    }
    const float invW = 1.0f / fmaxf(W, NORM_EPS);
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, sw = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int k = 0; k < DEF_K; ++k) {
      const float *cp = s_cp + idx[k] * CP_STRIDE;
      const float w = wt[k] * invW;
      const float qw = cp[7], qx = cp[8], qy = cp[9], qz = cp[10];
      if (LOCAL_FRAME) {
        const float inv = 1.0f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        float R[9];
        quat_R(qw * inv, qx * inv, qy * inv, qz * inv, R);
        const float l0 = x0 - cp[0], l1 = x1 - cp[1], l2 = x2 - cp[2];
        p0 += w * (R[0] * l0 + R[1] * l1 + R[2] * l2 + cp[0] + cp[4]);
        p1 += w * (R[3] * l0 + R[4] * l1 + R[5] * l2 + cp[1] + cp[5]);
        p2 += w * (R[6] * l0 + R[7] * l1 + R[8] * l2 + cp[2] + cp[6]);
      } else {
        p0 += w * cp[4], p1 += w * cp[5], p2 += w * cp[6];
      }
      sw += w * qw, sx += w * qx, sy += w * qy, sz += w * qz;
    }
    if (!LOCAL_FRAME) p0 += x0, p1 += x1, p2 += x2;
    // q1 = qs (x) q0, then unit-normalise
    const float aw = sw, ax = sx, ay = sy, az = sz, bw = q0.x, bx = q0.y, by = q0.z, bz = q0.w;
    const float ow = aw * bw - ax * bx - ay * by - az * bz;
    const float ox = aw * bx + ax * bw + ay * bz - az * by;
    const float oy = aw * by - ax * bz + ay * bw + az * bx;
    const float oz = aw * bz + ax * by - ay * bx + az * bw;
    const float inv_n = 1.0f / fmaxf(sqrtf(ow * ow + ox * ox + oy * oy + oz * oz), NORM_EPS);
    out_xyz[3 * i] = p0, out_xyz[3 * i + 1] = p1, out_xyz[3 * i + 2] = p2;
    *reinterpret_cast<float4 *>(out_rot + 4 * (size_t)i) = make_float4(ow * inv_n, ox * inv_n, oy * inv_n, oz * inv_n);
    out_scales[3 * i] = __expf(g.scaling[3 * i]);
    out_scales[3 * i + 1] = __expf(g.scaling[3 * i + 1]);
    out_scales[3 * i + 2] = __expf(g.scaling[3 * i + 2]);
    out_opacity[i] = 1.0f / (1.0f + __expf(-g.opacity[i]));
  }
}

template <bool LOCAL_FRAME>
__global__ void __launch_bounds__(DEF_BLOCK) lbs_fwd_kernel(int N, int M, GaussIO g, CtrlTable t,
                                                            float *__restrict__ out_xyz, float *__restrict__ out_rot,
                                                            float *__restrict__ out_scales,
                                                            float *__restrict__ out_opacity) {
  lbs_fwd_body<LOCAL_FRAME>(N, M, g, t, out_xyz, out_rot, out_scales, out_opacity);
}
// blockIdx.y = deformation group of the batch (every group skins the same canonical Gaussians with its own TimeNet
// rows; the renders of a group read the leader's buffers)
template <bool LOCAL_FRAME>
__global__ void __launch_bounds__(DEF_BLOCK) lbs_fwd_batched_kernel(int N, int M, GaussIO g, const float *c_xyz,
                                                                    const float *c_lr, RenderBatch b) {
  const dimo_render_desc &r = b.r[b.leader[blockIdx.y]];
  lbs_fwd_body<LOCAL_FRAME>(N, M, g, CtrlTable{c_xyz, c_lr, r.d_xyz, r.d_rot}, r.pts, r.rot, r.scales, r.opac);
}

// Where a workgroup's control-point sums (LDS table [M][CP_STRIDE]) go: its own partial table in global memory, summed
// by lbs_reduce_kernel in a fixed order (the single-render C ABI, deterministic) ...
struct PartialTable {
  float *partials;  // [gridDim.x][M][CP_STRIDE]
  __device__ __forceinline__ void store(int M, const float *s_acc) const {
    float *dst = partials + (size_t)blockIdx.x * M * CP_STRIDE;
    for (int j = threadIdx.x; j < M * CP_STRIDE; j += blockDim.x) dst[j] = s_acc[j];
  }
};
// ... or straight to their destinations as fp32 atomics on the rows the workgroup TOUCHED (the step executor, round 6:
// a workgroup of Morton-neighbours names 10-30 of the 512 control points; rounds 2-5 wrote the whole 22 KB table per
// workgroup -- 69 MB per step -- and summed 384 tables per group in a second kernel on every motion's critical path).
// Columns 0..2 -> d c_xyz, 3 -> d c_log_radius (shared by all groups and streams: atomics, like the TimeNet backward's
// adds to the same words), 4..6 / 7..10 -> the gradients of the group's TimeNet rows.
struct AtomicRows {
  float *g_c_xyz, *g_c_lr, *g_d_xyz, *g_d_rot;
  __device__ __forceinline__ void store(int M, const float *s_acc) const {
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
      const float *ac = s_acc + m * CP_STRIDE;
      float v[CP_STRIDE];
#pragma unroll
      for (int c = 0; c < CP_STRIDE; ++c) v[c] = ac[c];
      bool any = false;
#pragma unroll
      for (int c = 3; c < CP_STRIDE; ++c) any |= v[c] != 0.f;  // (columns 0..2 derive from 4..6)
      if (!any) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c) unsafeAtomicAdd(g_c_xyz + 3 * m + c, v[c]);
      unsafeAtomicAdd(g_c_lr + m, v[3]);
#pragma unroll
      for (int c = 0; c < 3; ++c) unsafeAtomicAdd(g_d_xyz + 3 * m + c, v[4 + c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) unsafeAtomicAdd(g_d_rot + 4 * m + c, v[7 + c]);
    }
  }
};

template <bool LOCAL_FRAME, bool ACC, class EXTRA, class SINK>
__device__ __forceinline__ void lbs_bwd_body(int N, int M, GaussIO g, CtrlTable t, const float *g_xyz,
                                             const float *g_rot, const float *g_scales, const float *g_opacity,
                                             float *d_xyz_out, float *d_rot_out, float *d_scaling_out,
                                             float *d_opacity_out, const SINK sink, const EXTRA extra) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_cp = smem;                    // control-point table
  float *s_acc = smem + M * CP_STRIDE;   // control-point gradient accumulators
  load_ctrl_to_lds(t, M, s_cp);
  for (int j = threadIdx.x; j < M * CP_STRIDE; j += blockDim.x) s_acc[j] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // whole waves stay in the loop (the control-point scatter below combines lanes); lanes past N compute on the
  // last Gaussian and contribute / store nothing
  for (int base = blockIdx.x * DEF_BLOCK; base < N; base += gridDim.x * DEF_BLOCK) {
    const bool valid = base + (int)threadIdx.x < N;
    const int i = valid ? base + (int)threadIdx.x : N - 1;
    const float x0 = g.xyz[3 * i], x1 = g.xyz[3 * i + 1], x2 = g.xyz[3 * i + 2];
    const float4 q0 = *reinterpret_cast<const float4 *>(g.rot + 4 * (size_t)i);
    const float4 dd = *reinterpret_cast<const float4 *>(g.nn_dist + 4 * (size_t)i);
    // every global input of this Gaussian is requested here, before the first use: spread over the body they were five
    // dependent memory round trips per iteration of a kernel that runs ~3 waves per SIMD
    float4 go = *reinterpret_cast<const float4 *>(g_rot + 4 * (size_t)i);
    float gp[3] = {g_xyz[3 * i], g_xyz[3 * i + 1], g_xyz[3 * i + 2]};
    float gsc[3] = {g_scales[3 * i], g_scales[3 * i + 1], g_scales[3 * i + 2]};
    float gopac = g_opacity[i];
    const float op_raw = g.opacity[i];
    const float sc_raw[3] = {g.scaling[3 * i], g.scaling[3 * i + 1], g.scaling[3 * i + 2]};
    extra.add_all((size_t)i, go, gp, gsc, gopac);
    int idx[DEF_K];
#pragma unroll
    for (int k = 0; k < DEF_K; ++k) idx[k] = (int)g.nn_idx[4 * (size_t)i + k];
    float4 d_rot;
    float dxs[3];
    lbs_bwd_math<LOCAL_FRAME>(s_cp, s_acc, x0, x1, x2, q0, dd, idx, go, gp, valid, lane, d_rot, dxs);
    if (valid) {
      float4 *dst = reinterpret_cast<float4 *>(d_rot_out + 4 * (size_t)i);
      float4 v = d_rot;
      if (ACC) {
        const float4 old = *dst;
        v.x += old.x, v.y += old.y, v.z += old.z, v.w += old.w;
      }
      *dst = v;
    }
    if (valid) {
      const float o = 1.0f / (1.0f + __expf(-op_raw));
      const float gop = gopac * o * (1.0f - o);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gs = gsc[c] * __expf(sc_raw[c]);
        d_xyz_out[3 * i + c] = ACC ? d_xyz_out[3 * i + c] + dxs[c] : dxs[c];
        d_scaling_out[3 * i + c] = ACC ? d_scaling_out[3 * i + c] + gs : gs;
      }
      d_opacity_out[i] = ACC ? d_opacity_out[i] + gop : gop;
    }
  }
  __syncthreads();
  if (LOCAL_FRAME) {  // columns 0..2 from the summed columns 4..6 (deform_body.hpp)
    lbs_ctrl_position_grad(M, s_cp, s_acc);
    __syncthreads();
  }
  sink.store(M, s_acc);
}

template <bool LOCAL_FRAME, bool ACC>
__global__ void __launch_bounds__(DEF_BLOCK) lbs_bwd_kernel(
    int N, int M, GaussIO g, CtrlTable t, const float *__restrict__ g_xyz, const float *__restrict__ g_rot,
    const float *__restrict__ g_scales, const float *__restrict__ g_opacity, float *__restrict__ d_xyz_out,
    float *__restrict__ d_rot_out, float *__restrict__ d_scaling_out, float *__restrict__ d_opacity_out,
    float *__restrict__ partials) {
  lbs_bwd_body<LOCAL_FRAME, ACC>(N, M, g, t, g_xyz, g_rot, g_scales, g_opacity, d_xyz_out, d_rot_out, d_scaling_out,
                                 d_opacity_out, PartialTable{partials}, NoExtra{});
}
// The step executor's skinning backward.  blockIdx.y = deformation group; per-Gaussian gradients are written IN PLACE
// over the LEADER's rasterizer gradients (g_means3D -> d xyz, g_rot -> d rotation, g_scales -> d scaling, g_opac ->
// d opacity: same shapes) and accumulate_batched_kernel folds the leaders into the shared gradient views in a fixed
// order (the per-Gaussian head of the bucket stays bit-reproducible); the control-point sums leave the workgroup as
// atomics on the rows it touched (AtomicRows): no partial tables, no reduction kernel on the motion's critical path.
// (Round 6 also built the other end of that idea -- ONE thread per Gaussian walking the launch's groups one after the
// other, all 14 per-Gaussian sums kept in registers and added to the bucket with one atomic each: no fold either.  Parity
// green, and SLOWER: 106 us per launch of four groups against 38 + 15 + 24, the step -4 % -- the kernel is a latency
// chain per wave, and groups walked in sequence by a third of the waves take the sum of their latencies:
// profiles/r06_skinning_backward.txt.)
template <bool LOCAL_FRAME>
__global__ void __launch_bounds__(DEF_BLOCK) lbs_bwd_batched_kernel(int N, int M, GaussIO g, const float *c_xyz,
                                                                    const float *c_lr, RenderBatch b, float *g_c_xyz,
                                                                    float *g_c_lr) {
  const int lead = b.leader[blockIdx.y];
  const dimo_render_desc &r = b.r[lead];
  lbs_bwd_body<LOCAL_FRAME, false>(N, M, g, CtrlTable{c_xyz, c_lr, r.d_xyz, r.d_rot}, r.g_means3D, r.g_rot,
                                   r.g_scales, r.g_opac, r.g_means3D, r.g_rot, r.g_scales, r.g_opac,
                                   AtomicRows{g_c_xyz, g_c_lr, r.g_d_xyz, r.g_d_rot},
                                   GroupExtra{b, b.members[blockIdx.y] & ~(1u << lead)});
}

// dst[i] += sum_r src_r[i] for the five per-Gaussian gradient arrays of a batch (fixed order): the skinning backward
// left the first four in the group leaders' buffers, the colour gradient is per render.
__global__ void __launch_bounds__(256) accumulate_batched_kernel(int N, int n_renders, RenderBatch b, float *g_xyz,
                                                                 float *g_rotation, float *g_scaling,
                                                                 float *g_opacity, float *g_f_dc) {
  // segments of the flat index space, in units of FOUR floats (every array starts on a 16-byte boundary: the gradient
  // views by construction of the flat bucket, the slots' buffers by allocation): [0, 3N) xyz | rotation 4N | scaling 3N
  // | opacity N | f_dc 3N, each rounded up to whole float4s -- 16-byte loads and stores (round 6: the scalar form read
  // 38 MB in 23 us), the last float4 of a segment whose length is not a multiple of four element by element.
  const size_t n = (size_t)N;
  const size_t len[5] = {3 * n, 4 * n, 3 * n, n, 3 * n};
  size_t q4[6];
  q4[0] = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) q4[k + 1] = q4[k] + (len[k] + 3) / 4;
  unsigned leaders = 0;
  for (int q = 0; q < b.n_groups; ++q) leaders |= 1u << b.leader[q];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < q4[5]; i += (size_t)gridDim.x * 256) {
    const int which = i < q4[1] ? 0 : i < q4[2] ? 1 : i < q4[3] ? 2 : i < q4[4] ? 3 : 4;
    const size_t k = 4 * (i - q4[which]);
    float *dst = which == 0 ? g_xyz : which == 1 ? g_rotation : which == 2 ? g_scaling : which == 3 ? g_opacity : g_f_dc;
    if (k + 4 <= len[which]) {
      float4 s = *reinterpret_cast<const float4 *>(dst + k);
      float4 v[MAX_BATCH];
#pragma unroll
      for (int r = 0; r < MAX_BATCH; ++r) {  // (every load of the sum requested before the first add)
        const bool on = r < n_renders && (which == 4 || ((leaders >> r) & 1u));
        const dimo_render_desc &d = b.r[r];
        const float *src = which == 0 ? d.g_means3D : which == 1 ? d.g_rot : which == 2 ? d.g_scales
                           : which == 3 ? d.g_opac : d.g_shs;
        v[r] = on ? *reinterpret_cast<const float4 *>(src + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int r = 0; r < MAX_BATCH; ++r) {
        const bool on = r < n_renders && (which == 4 || ((leaders >> r) & 1u));
        if (on) s.x += v[r].x, s.y += v[r].y, s.z += v[r].z, s.w += v[r].w;  // (fixed order: render 0, 1, ...)
      }
      *reinterpret_cast<float4 *>(dst + k) = s;
    } else {
      for (size_t e = k; e < len[which]; ++e) {
        float s = dst[e];
        for (int r = 0; r < n_renders; ++r) {
          if (which != 4 && !((leaders >> r) & 1u)) continue;
          const dimo_render_desc &d = b.r[r];
          const float *src = which == 0 ? d.g_means3D : which == 1 ? d.g_rot : which == 2 ? d.g_scales
                             : which == 3 ? d.g_opac : d.g_shs;
          s += src[e];
        }
        dst[e] = s;
      }
    }
  }
}

// at most four workgroups of four waves per CU (see accumulate_batched_kernel)
static unsigned acc_grid(size_t total) {
  const size_t want = (total + 255) / 256;
  return (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

// sums the per-workgroup partial tables in a fixed order and scatters into the four gradient tensors
__global__ void __launch_bounds__(256) lbs_reduce_kernel(int M, int nblocks, int accumulate,
                                                         const float *__restrict__ partials,
                                                         float *__restrict__ d_c_xyz, float *__restrict__ d_c_lr,
                                                         float *__restrict__ d_d_xyz, float *__restrict__ d_d_rot) {
  // A workgroup owns 16 consecutive outputs; thread (jj = tid & 15, chunk = tid >> 4) sums tables chunk,
  // chunk+16, ... for output j0 + jj (16 lanes read 64 contiguous bytes), then the 16 chunk sums are added
  // in a fixed order through LDS -> deterministic, and M*11/16 workgroups keep every CU busy.
  __shared__ float s_part[16][17];
  const int jj = threadIdx.x & 15, chunk = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + jj;
  float s = 0.f;
  if (j < M * CP_STRIDE)
    for (int b = chunk; b < nblocks; b += 16) s += partials[(size_t)b * M * CP_STRIDE + j];
  s_part[chunk][jj] = s;
  __syncthreads();
  if (chunk != 0 || j >= M * CP_STRIDE) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += s_part[k][jj];
  const int m = j / CP_STRIDE, c = j % CP_STRIDE;
  float *dst;
  if (c < 3) dst = d_c_xyz + 3 * m + c;
  else if (c == 3) dst = d_c_lr + m;
  else if (c < 7) dst = d_d_xyz + 3 * m + (c - 4);
  else dst = d_d_rot + 4 * m + (c - 7);
  *dst = accumulate ? *dst + s : s;
}

// one workgroup per CU at most: the backward writes one partial control-point table per workgroup
inline int deform_grid(int N) {
  const int want = (N + DEF_BLOCK - 1) / DEF_BLOCK;
  return want < 1 ? 1 : (want > 256 ? 256 : want);
}

// workgroups per deformation group of a batched launch: `total` over the groups, at most one per DEF_BLOCK Gaussians
inline int batched_grid(int N, int n_groups, int total) {
  const int per = (N + DEF_BLOCK - 1) / DEF_BLOCK;
  const int g = total / (n_groups > 0 ? n_groups : 1);
  return per < 1 ? 1 : (g < 1 ? 1 : (g < per ? g : per));
}

// control-point tables larger than the default 64 KiB dynamic-LDS window need the opt-in attribute
inline void allow_big_lds() {
  static const bool once = [] {
    const int lim = 160 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_fwd_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_fwd_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_kernel<true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_kernel<false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_kernel<true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_kernel<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_fwd_batched_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_fwd_batched_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_batched_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lbs_bwd_batched_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    return true;
  }();
  (void)once;
}

// ---- batched host entry points (native step executor) ------------------------------------------------------------
void group_deformations(RenderBatch &b, int n) {
  b.n_groups = 0;
  for (int i = 0; i < MAX_BATCH; ++i) b.leader[i] = 0, b.members[i] = 0;
  for (int i = 0; i < n; ++i) {
    int q = 0;
    for (; q < b.n_groups; ++q) {
      const dimo_render_desc &l = b.r[b.leader[q]];
      if (l.d_xyz == b.r[i].d_xyz && l.d_rot == b.r[i].d_rot && l.g_d_xyz == b.r[i].g_d_xyz &&
          l.g_d_rot == b.r[i].g_d_rot)
        break;
    }
    if (q == b.n_groups) b.leader[b.n_groups++] = (unsigned char)i;
    b.members[q] |= (unsigned char)(1u << i);
    const dimo_render_desc &l = b.r[b.leader[q]];
    b.r[i].pts = l.pts, b.r[i].rot = l.rot, b.r[i].scales = l.scales, b.r[i].opac = l.opac;
  }
  for (int i = n; i < MAX_BATCH; ++i) b.r[i] = b.r[n - 1];  // padding entries follow the redirected last render
}

// ---- stage s1: direct deformation (renderer/latent_gs_renderer.py:1176-1177, 1211-1212, get_scaling :341-351) ------
// The TimeNet moves every Gaussian itself: pts = xyz + d_xyz[i], rotation = normalize(rotation), scale = exp(_r) on
// all three axes (the shared (1, 1) log-radius), opacity = sigmoid.  One thread per Gaussian, blockIdx.y = group.
__global__ void __launch_bounds__(256) s1_fwd_batched_kernel(int N, const float *__restrict__ xyz,
                                                             const float *__restrict__ rotation,
                                                             const float *__restrict__ opacity,
                                                             const float *__restrict__ log_r, RenderBatch b) {
  const dimo_render_desc &r = b.r[b.leader[blockIdx.y]];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float s = __expf(log_r[0]);
  r.pts[3 * i] = xyz[3 * i] + r.d_xyz[3 * i];
  r.pts[3 * i + 1] = xyz[3 * i + 1] + r.d_xyz[3 * i + 1];
  r.pts[3 * i + 2] = xyz[3 * i + 2] + r.d_xyz[3 * i + 2];
  const float4 q = *reinterpret_cast<const float4 *>(rotation + 4 * (size_t)i);
  const float inv_n = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  *reinterpret_cast<float4 *>(r.rot + 4 * (size_t)i) = make_float4(q.x * inv_n, q.y * inv_n, q.z * inv_n, q.w * inv_n);
  r.scales[3 * i] = s, r.scales[3 * i + 1] = s, r.scales[3 * i + 2] = s;
  r.opac[i] = 1.0f / (1.0f + __expf(-opacity[i]));
}
// Backward, in place over the LEADER's rasterizer gradients like lbs_bwd_batched_kernel (accumulate_batched_kernel
// then folds the leaders into the shared gradient views): d xyz = sum of the group's g_means3D, which is also the
// gradient of this pair's TimeNet output row; normalize / sigmoid backward; the radius gradient is reduced per
// workgroup and added atomically (a single float, N <= a few thousand in this stage).
__global__ void __launch_bounds__(256) s1_bwd_batched_kernel(int N, const float *__restrict__ rotation,
                                                             const float *__restrict__ opacity,
                                                             const float *__restrict__ log_r, float *g_log_r,
                                                             RenderBatch b) {
  __shared__ float s_red[4];
  const int lead = b.leader[blockIdx.y];
  const dimo_render_desc &r = b.r[lead];
  const unsigned others = b.members[blockIdx.y] & ~(1u << lead);
  const int i = blockIdx.x * 256 + threadIdx.x;
  float gr = 0.0f;
  if (i < N) {
    float gx = r.g_means3D[3 * i], gy = r.g_means3D[3 * i + 1], gz = r.g_means3D[3 * i + 2];
    float4 gq = *reinterpret_cast<const float4 *>(r.g_rot + 4 * (size_t)i);
    float gs = r.g_scales[3 * i] + r.g_scales[3 * i + 1] + r.g_scales[3 * i + 2];
    float go = r.g_opac[i];
    for (unsigned m = others; m; m &= m - 1) {
      const dimo_render_desc &o = b.r[__ffs(m) - 1];
      gx += o.g_means3D[3 * i], gy += o.g_means3D[3 * i + 1], gz += o.g_means3D[3 * i + 2];
      const float4 t = *reinterpret_cast<const float4 *>(o.g_rot + 4 * (size_t)i);
      gq.x += t.x, gq.y += t.y, gq.z += t.z, gq.w += t.w;
      gs += o.g_scales[3 * i] + o.g_scales[3 * i + 1] + o.g_scales[3 * i + 2];
      go += o.g_opac[i];
    }
    r.g_d_xyz[3 * i] += gx, r.g_d_xyz[3 * i + 1] += gy, r.g_d_xyz[3 * i + 2] += gz;
    r.g_means3D[3 * i] = gx, r.g_means3D[3 * i + 1] = gy, r.g_means3D[3 * i + 2] = gz;
    const float4 q = *reinterpret_cast<const float4 *>(rotation + 4 * (size_t)i);
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f), inv_n = 1.0f / nrm;
    const float ux = q.x * inv_n, uy = q.y * inv_n, uz = q.z * inv_n, uw = q.w * inv_n;
    const float dot = ux * gq.x + uy * gq.y + uz * gq.z + uw * gq.w;
    *reinterpret_cast<float4 *>(r.g_rot + 4 * (size_t)i) =
        make_float4((gq.x - ux * dot) * inv_n, (gq.y - uy * dot) * inv_n, (gq.z - uz * dot) * inv_n, (gq.w - uw * dot) * inv_n);
    r.g_scales[3 * i] = 0.0f, r.g_scales[3 * i + 1] = 0.0f, r.g_scales[3 * i + 2] = 0.0f;  // `_scaling` is unused in s1
    const float sg = 1.0f / (1.0f + __expf(-opacity[i]));
    r.g_opac[i] = go * sg * (1.0f - sg);
    gr = gs * __expf(log_r[0]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gr += __shfl_down(gr, o, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = gr;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(g_log_r, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

int lbs_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (c.stage1) {
    if (c.N <= 0 || n <= 0) return DIMO_OK;
    if (!c.log_r) return DIMO_E_ARG;
    ScopedTimer tm(T_DEFORM_FWD, stream);
    hipLaunchKernelGGL(s1_fwd_batched_kernel, dim3((c.N + 255) / 256, b.n_groups), dim3(256), 0, stream, c.N, c.xyz,
                       c.rotation, c.opacity, c.log_r, b);
    return check_launch();
  }
  if (c.N <= 0 || n <= 0) return DIMO_OK;
  GaussIO g{c.xyz, c.rotation, c.scaling, c.opacity, c.nn_dist, c.nn_idx};
  const size_t lds = (size_t)c.M * CP_STRIDE * sizeof(float);
  allow_big_lds();
  ScopedTimer tm(T_DEFORM_FWD, stream);
  // the control-point table is re-built per workgroup: fewer, fatter workgroups per render when batching
  const int fwd_total = 512;
  const int grid = batched_grid(c.N, b.n_groups, fwd_total);
  if (c.local_frame)
    hipLaunchKernelGGL(lbs_fwd_batched_kernel<true>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M, g, c.c_xyz,
                       c.c_log_radius, b);
  else
    hipLaunchKernelGGL(lbs_fwd_batched_kernel<false>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M, g,
                       c.c_xyz, c.c_log_radius, b);
  return check_launch();
}

// (Rounds 2-5 kept per-workgroup partial control-point tables and per-leader staging tables in `lbs_scratch`; since
// round 6 the control-point sums leave the kernel as atomics.  The size query stays for callers that still allocate it.)
size_t lbs_backward_batched_scratch_bytes(int N, int M, int n) {
  (void)N, (void)M, (void)n;
  return ALIGN;
}

// phase 0: the whole skinning backward of the batch on `stream` (skin + per-Gaussian fold).  Phased form for batches
// that are skinned on different streams at once: phase 1 = skin (per-Gaussian gradients in place in the leaders' buffers;
// what it adds to shared words -- the control-point sums -- it adds atomically), phase 2 = the fold over ALL the step's
// renders on one stream.
int lbs_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream, int first_abs,
                         int phase) {
  if (c.N <= 0 || n <= 0) return DIMO_OK;
  const size_t total = (14 * (size_t)c.N + 3) / 4 + 5;  // float4s of the fold's five segments
  if (c.stage1) {
    if (phase == 2) return DIMO_OK;  // (stage s1 has no control points: phase 1 does everything)
    if (!c.log_r || !c.g_log_r) return DIMO_E_ARG;
    ScopedTimer tm(T_DEFORM_BWD, stream);
    hipLaunchKernelGGL(s1_bwd_batched_kernel, dim3((c.N + 255) / 256, b.n_groups), dim3(256), 0, stream, c.N,
                       c.rotation, c.opacity, c.log_r, c.g_log_r, b);
    hipLaunchKernelGGL(accumulate_batched_kernel, dim3(acc_grid(total)), dim3(256), 0, stream, c.N, n, b, c.g_xyz,
                       c.g_rotation, c.g_scaling, c.g_opacity, c.g_f_dc);
    return check_launch();
  }
  if (first_abs < 0 || phase < 0 || phase > 2) return DIMO_E_ARG;
  if (!c.g_xyz || !c.g_rotation || !c.g_scaling || !c.g_opacity || !c.g_f_dc || !c.g_c_xyz || !c.g_c_log_radius)
    return DIMO_E_ARG;
  ScopedTimer tm(T_DEFORM_BWD, stream);
  if (phase != 2) {
    GaussIO g{c.xyz, c.rotation, c.scaling, c.opacity, c.nn_dist, c.nn_idx};
    const size_t lds = 2 * (size_t)c.M * CP_STRIDE * sizeof(float);
    allow_big_lds();
    // 768 = three workgroups (45 KB of LDS, 144 VGPRs) on each of the 256 CUs.  The kernel is a latency chain per wave
    // (1 700 VALU instructions and 32 LDS-atomic instructions per 64 Gaussians) with N / 64 waves per group in all, so
    // it wants every CU slot: with 128 workgroups per group (round 1) a launch of two groups ran one wave per SIMD.
    const int bwd_total = 768;
    const int grid = batched_grid(c.N, b.n_groups, bwd_total);
    if (c.local_frame)
      hipLaunchKernelGGL(lbs_bwd_batched_kernel<true>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M,
                         g, c.c_xyz, c.c_log_radius, b, c.g_c_xyz, c.g_c_log_radius);
    else
      hipLaunchKernelGGL(lbs_bwd_batched_kernel<false>, dim3(grid, b.n_groups), dim3(DEF_BLOCK), lds, stream, c.N, c.M,
                         g, c.c_xyz, c.c_log_radius, b, c.g_c_xyz, c.g_c_log_radius);
  }
  if (phase != 1)
    hipLaunchKernelGGL(accumulate_batched_kernel, dim3(acc_grid(total)), dim3(256), 0, stream, c.N, n, b, c.g_xyz,
                       c.g_rotation, c.g_scaling, c.g_opacity, c.g_f_dc);
  return check_launch();
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_deform_max_ctrl_points(void) { return (160 * 1024 / 2) / (CP_STRIDE * (int)sizeof(float)); }

extern "C" size_t dimo_deform_backward_scratch_bytes(int N, int M) {  // one table per DEF_BLOCK Gaussians at most
  const size_t per = (size_t)((N > 0 ? N : 1) + DEF_BLOCK - 1) / DEF_BLOCK;
  return align_up(per * (size_t)(M > 0 ? M : 1) * CP_STRIDE * sizeof(float));
}

extern "C" int dimo_deform_forward(int N, int M, int local_frame, const float *xyz, const float *rotation,
                                   const float *scaling, const float *opacity, const float *c_xyz,
                                   const float *c_log_radius, const float *d_xyz, const float *d_rot,
                                   const float *nn_dist, const int64_t *nn_idx, float *out_xyz, float *out_rot,
                                   float *out_scales, float *out_opacity, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0 || M <= 0 || M > dimo_deform_max_ctrl_points()) return DIMO_E_ARG;
  if (N == 0) return DIMO_OK;
  if (!xyz || !rotation || !scaling || !opacity || !c_xyz || !c_log_radius || !d_xyz || !d_rot || !nn_dist ||
      !nn_idx || !out_xyz || !out_rot || !out_scales || !out_opacity)
    return DIMO_E_ARG;
  GaussIO g{xyz, rotation, scaling, opacity, nn_dist, nn_idx};
  CtrlTable t{c_xyz, c_log_radius, d_xyz, d_rot};
  const size_t lds = (size_t)M * CP_STRIDE * sizeof(float);
  allow_big_lds();
  ScopedTimer tm(T_DEFORM_FWD, stream);
  if (local_frame)
    hipLaunchKernelGGL(lbs_fwd_kernel<true>, dim3(deform_grid(N)), dim3(DEF_BLOCK), lds, stream, N, M, g, t, out_xyz,
                       out_rot, out_scales, out_opacity);
  else
    hipLaunchKernelGGL(lbs_fwd_kernel<false>, dim3(deform_grid(N)), dim3(DEF_BLOCK), lds, stream, N, M, g, t, out_xyz,
                       out_rot, out_scales, out_opacity);
  return check_launch();
}

extern "C" int dimo_deform_backward(int N, int M, int local_frame, int accumulate, const float *xyz,
                                    const float *rotation,
                                    const float *scaling, const float *opacity, const float *c_xyz,
                                    const float *c_log_radius, const float *d_xyz, const float *d_rot,
                                    const float *nn_dist, const int64_t *nn_idx, const float *g_out_xyz,
                                    const float *g_out_rot, const float *g_out_scales, const float *g_out_opacity,
                                    float *dL_dxyz, float *dL_drotation, float *dL_dscaling, float *dL_dopacity,
                                    float *dL_dc_xyz, float *dL_dc_log_radius, float *dL_dd_xyz, float *dL_dd_rot,
                                    void *scratch, size_t scratch_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0 || M <= 0 || M > dimo_deform_max_ctrl_points()) return DIMO_E_ARG;
  if (!dL_dc_xyz || !dL_dc_log_radius || !dL_dd_xyz || !dL_dd_rot || !scratch) return DIMO_E_ARG;
  if (scratch_bytes < dimo_deform_backward_scratch_bytes(N, M)) return DIMO_E_WORKSPACE;
  if (N > 0 && (!xyz || !rotation || !scaling || !opacity || !c_xyz || !c_log_radius || !d_xyz || !d_rot ||
                !nn_dist || !nn_idx || !g_out_xyz || !g_out_rot || !g_out_scales || !g_out_opacity || !dL_dxyz ||
                !dL_drotation || !dL_dscaling || !dL_dopacity))
    return DIMO_E_ARG;
  GaussIO g{xyz, rotation, scaling, opacity, nn_dist, nn_idx};
  CtrlTable t{c_xyz, c_log_radius, d_xyz, d_rot};
  const int grid = N > 0 ? deform_grid(N) : 0;
  const size_t lds = 2 * (size_t)M * CP_STRIDE * sizeof(float);
  allow_big_lds();
  ScopedTimer tm(T_DEFORM_BWD, stream);
  if (grid > 0) {
#define DIMO_LAUNCH_LBS_BWD(LF, AC)                                                                              \
  hipLaunchKernelGGL((lbs_bwd_kernel<LF, AC>), dim3(grid), dim3(DEF_BLOCK), lds, stream, N, M, g, t, g_out_xyz, \
                     g_out_rot, g_out_scales, g_out_opacity, dL_dxyz, dL_drotation, dL_dscaling, dL_dopacity,   \
                     reinterpret_cast<float *>(scratch))
    if (local_frame && accumulate) DIMO_LAUNCH_LBS_BWD(true, true);
    else if (local_frame) DIMO_LAUNCH_LBS_BWD(true, false);
    else if (accumulate) DIMO_LAUNCH_LBS_BWD(false, true);
    else DIMO_LAUNCH_LBS_BWD(false, false);
#undef DIMO_LAUNCH_LBS_BWD
  }
  hipLaunchKernelGGL(lbs_reduce_kernel, dim3((M * CP_STRIDE + 15) / 16), dim3(256), 0, stream, M, grid,
                     accumulate ? 1 : 0, reinterpret_cast<const float *>(scratch), dL_dc_xyz, dL_dc_log_radius,
                     dL_dd_xyz, dL_dd_rot);
  return check_launch();
}
