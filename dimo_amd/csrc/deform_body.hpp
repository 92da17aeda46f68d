// Device code of the skinning stage (deform.hip); a header since round 5's fused backward tail (projection backward ->
// skinning backward in one kernel: built, parity-green, measured slower and removed -- profiles/r05_fused_tail.txt) used
// it from a second translation unit.  See deform.hip for the maths' provenance (renderer/latent_gs_renderer.py:1187-1219).
#pragma once
#include "common.hpp"
#include "wave_ops.hpp"

namespace dimo {

constexpr int DEF_BLOCK = 256;
constexpr int DEF_K = 4;          // neighbours per Gaussian (find_knn k=4, main_train_dimo.py:257)
constexpr int CP_STRIDE = 11;     // per control point: c(3) lr(1)->r dc(3) dq(4)   |  grads: gc(3) glr(1) gdc(3) gdq(4)
constexpr float LBS_EPS = 1e-7f;
constexpr float NORM_EPS = 1e-12f;

struct CtrlTable {
  const float *c_xyz;     // [M,3]
  const float *c_lr;      // [M]   log radius (_c_radius)
  const float *d_xyz;     // [M,3] TimeNet translation
  const float *d_rot;     // [M,4] TimeNet quaternion (w,x,y,z), not normalised
};

__device__ __forceinline__ void load_ctrl_to_lds(const CtrlTable &t, int M, float *s) {
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    float *r = s + j * CP_STRIDE;
    r[0] = t.c_xyz[3 * j], r[1] = t.c_xyz[3 * j + 1], r[2] = t.c_xyz[3 * j + 2];
    r[3] = __expf(t.c_lr[j]);
    r[4] = t.d_xyz[3 * j], r[5] = t.d_xyz[3 * j + 1], r[6] = t.d_xyz[3 * j + 2];
    r[7] = t.d_rot[4 * j], r[8] = t.d_rot[4 * j + 1], r[9] = t.d_rot[4 * j + 2], r[10] = t.d_rot[4 * j + 3];
  }
}

__device__ __forceinline__ void quat_R(float w, float x, float y, float z, float *R) {
  R[0] = 1.f - 2.f * (y * y + z * z), R[1] = 2.f * (x * y - w * z), R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z), R[4] = 1.f - 2.f * (x * x + z * z), R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y), R[7] = 2.f * (y * z + w * x), R[8] = 1.f - 2.f * (x * x + y * y);
}

struct GaussIO {
  const float *xyz, *rot, *scaling, *opacity;   // [N,3] [N,4] [N,3] [N]  (raw parameters)
  const float *nn_dist;                         // [N,4]
  const int64_t *nn_idx;                        // [N,4]
};

// ACC: add into the per-Gaussian outputs instead of overwriting them (a training step sums the gradients
// of all its renders straight into the flat gradient bucket).
// The batched launch runs this in place (outputs over the inputs of the same Gaussian, each element read before
// it is written by its own thread), hence no __restrict__ on the eight per-Gaussian arrays.
// EXTRA adds the rasterizer gradients of the other renders of a deformation group to the leader's (the backward is
// linear in them); NoExtra for a single render.
struct NoExtra {
  __device__ __forceinline__ void add_all(size_t, float4 &, float (&)[3], float (&)[3], float &) const {}
};
struct GroupExtra {
  const RenderBatch &b;
  unsigned others;  // bitmask of the group's renders without the leader
  // the four gradient arrays of one other render are requested together, before the first add (one memory round trip
  // per member instead of four spread over the body)
  __device__ __forceinline__ void add_all(size_t i, float4 &rot, float (&xyz)[3], float (&sc)[3], float &op) const {
#pragma unroll
    for (int j = 1; j < MAX_BATCH; ++j)
      if ((others >> j) & 1u) {
        const dimo_render_desc &r = b.r[j];
        const float4 t = *reinterpret_cast<const float4 *>(r.g_rot + 4 * i);
        const float p0 = r.g_means3D[3 * i], p1 = r.g_means3D[3 * i + 1], p2 = r.g_means3D[3 * i + 2];
        const float s0 = r.g_scales[3 * i], s1 = r.g_scales[3 * i + 1], s2 = r.g_scales[3 * i + 2];
        const float o = r.g_opac[i];
        rot.x += t.x, rot.y += t.y, rot.z += t.z, rot.w += t.w;
        xyz[0] += p0, xyz[1] += p1, xyz[2] += p2;
        sc[0] += s0, sc[1] += s1, sc[2] += s2;
        op += o;
      }
  }
};

// Skinning backward of ONE Gaussian (all lanes of a wave call this together: the control-point scatter combines lanes).
// In: the Gaussian's canonical position / rotation, its four neighbours (distance, control-point index: `idx` is
// re-ordered in place), the gradients of its skinned position `gp` and rotation `go`.  Out: the gradients of the
// canonical rotation `d_rot` and position `dxs`; columns 3..10 of the neighbours' gradient rows are added to the
// workgroup's LDS table `s_acc` (columns 0..2 follow after the workgroup's loop: lbs_ctrl_position_grad).
template <bool LOCAL_FRAME>
__device__ __forceinline__ void lbs_bwd_math(const float *s_cp, float *s_acc, float x0, float x1, float x2, float4 q0,
                                             float4 dd, int (&idx)[DEF_K], float4 go, const float (&gp)[3], bool valid,
                                             int lane, float4 &d_rot, float (&dxs)[3]) {
  float wt[DEF_K], ex[DEF_K], W = 0.f;
  float dist[DEF_K] = {dd.x, dd.y, dd.z, dd.w};
  // The neighbour slots are re-ordered by control-point index (a 5-exchange network on (index, distance)): the sums
  // over the four neighbours do not care, and Gaussians that are neighbours in memory (Morton order) then hold the
  // same control point in the same slot, so the run-combining of the LDS atomics below merges far more lanes -- the
  // kernel is bound by the LDS atomic unit (~4 cycles per active lane).
#define DIMO_CX(A, B)                                                                 \
{                                                                                   \
  const bool sw = idx[A] > idx[B];                                                  \
  const int ia = sw ? idx[B] : idx[A], ib = sw ? idx[A] : idx[B];                   \
  const float da = sw ? dist[B] : dist[A], db = sw ? dist[A] : dist[B];             \
  idx[A] = ia, idx[B] = ib, dist[A] = da, dist[B] = db;                             \
}
  DIMO_CX(0, 1) DIMO_CX(2, 3) DIMO_CX(0, 2) DIMO_CX(1, 3) DIMO_CX(1, 2)
#undef DIMO_CX
#pragma unroll
  for (int k = 0; k < DEF_K; ++k) {
    const float r = s_cp[idx[k] * CP_STRIDE + 3];
    ex[k] = __expf(-1.0f * dist[k] * dist[k] / (2.0f * (r * r)));
    wt[k] = ex[k] + LBS_EPS;
    W += wt[k];
  }
  const float invW = 1.0f / fmaxf(W, NORM_EPS);
  // recompute the blended quaternion
  float sw = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int k = 0; k < DEF_K; ++k) {
    const float *cp = s_cp + idx[k] * CP_STRIDE;
    const float w = wt[k] * invW;
    sw += w * cp[7], sx += w * cp[8], sy += w * cp[9], sz += w * cp[10];
  }
  const float bw = q0.x, bx = q0.y, by = q0.z, bz = q0.w;
  const float ow = sw * bw - sx * bx - sy * by - sz * bz;
  const float ox = sw * bx + sx * bw + sy * bz - sz * by;
  const float oy = sw * by - sx * bz + sy * bw + sz * bx;
  const float oz = sw * bz + sx * by - sy * bx + sz * bw;
  const float nrm = fmaxf(sqrtf(ow * ow + ox * ox + oy * oy + oz * oz), NORM_EPS);
  const float inv_n = 1.0f / nrm;
  const float uw = ow * inv_n, ux = ox * inv_n, uy = oy * inv_n, uz = oz * inv_n;
  // normalisation backward
  const float dotg = uw * go.x + ux * go.y + uy * go.z + uz * go.w;
  const float gw = (go.x - uw * dotg) * inv_n, gx = (go.y - ux * dotg) * inv_n;
  const float gy = (go.z - uy * dotg) * inv_n, gz = (go.w - uz * dotg) * inv_n;
  // quaternion product backward: a = blended (sw..), b = the Gaussian's own rotation
  const float gaw = gw * bw + gx * bx + gy * by + gz * bz;
  const float gax = -gw * bx + gx * bw - gy * bz + gz * by;
  const float gay = -gw * by + gx * bz + gy * bw - gz * bx;
  const float gaz = -gw * bz - gx * by + gy * bx + gz * bw;
  const float gbw = gw * sw + gx * sx + gy * sy + gz * sz;
  const float gbx = -gw * sx + gx * sw + gy * sz - gz * sy;
  const float gby = -gw * sy - gx * sz + gy * sw + gz * sx;
  const float gbz = -gw * sz + gx * sy - gy * sx + gz * sw;
  d_rot = make_float4(gbw, gbx, gby, gbz);

  const float gp0 = gp[0], gp1 = gp[1], gp2 = gp[2];
  float dx0 = LOCAL_FRAME ? 0.f : gp0, dx1 = LOCAL_FRAME ? 0.f : gp1, dx2 = LOCAL_FRAME ? 0.f : gp2;
  float gwk[DEF_K], sum_wg = 0.f;
  // this Gaussian's contribution to columns 3..10 of the gradient row of neighbour k (columns 0..2, the gradient
  // of the control point's position, follow from columns 4..6 after the reduction: see the end of the kernel)
  static_assert(CP_STRIDE - 3 == 8, "wave_scatter_add_match8 takes eight values per neighbour");
  float cpg[DEF_K][CP_STRIDE - 3];
#pragma unroll
  for (int k = 0; k < DEF_K; ++k) {
    const float *cp = s_cp + idx[k] * CP_STRIDE;
    const float w = wt[k] * invW;
    const float qw = cp[7], qx = cp[8], qy = cp[9], qz = cp[10];
    // d/d(dq) through the blended quaternion
    float gq_w = w * gaw, gq_x = w * gax, gq_y = w * gay, gq_z = w * gaz;
    float gwt = qw * gaw + qx * gax + qy * gay + qz * gaz;  // dL/dw_k
    if (LOCAL_FRAME) {
      const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), inv = 1.0f / qn;
      const float r_ = qw * inv, x_ = qx * inv, y_ = qy * inv, z_ = qz * inv;
      float R[9];
      quat_R(r_, x_, y_, z_, R);
      const float l0 = x0 - cp[0], l1 = x1 - cp[1], l2 = x2 - cp[2];
      const float y0 = R[0] * l0 + R[1] * l1 + R[2] * l2 + cp[0] + cp[4];
      const float y1 = R[3] * l0 + R[4] * l1 + R[5] * l2 + cp[1] + cp[5];
      const float y2 = R[6] * l0 + R[7] * l1 + R[8] * l2 + cp[2] + cp[6];
      gwt += y0 * gp0 + y1 * gp1 + y2 * gp2;
      const float gy0 = w * gp0, gy1 = w * gp1, gy2 = w * gp2;  // dL/dy_k
      // R^T gy
      const float rt0 = R[0] * gy0 + R[3] * gy1 + R[6] * gy2;
      const float rt1 = R[1] * gy0 + R[4] * gy1 + R[7] * gy2;
      const float rt2 = R[2] * gy0 + R[5] * gy1 + R[8] * gy2;
      dx0 += rt0, dx1 += rt1, dx2 += rt2;
      cpg[k][1] = gy0, cpg[k][2] = gy1, cpg[k][3] = gy2;
      // dL/dR = gy (x - c)^T  -> unit quaternion -> raw quaternion
      const float d00 = gy0 * l0, d01 = gy0 * l1, d02 = gy0 * l2;
      const float d10 = gy1 * l0, d11 = gy1 * l1, d12 = gy1 * l2;
      const float d20 = gy2 * l0, d21 = gy2 * l1, d22 = gy2 * l2;
      const float gur = 2.f * (-z_ * d01 + y_ * d02 + z_ * d10 - x_ * d12 - y_ * d20 + x_ * d21);
      const float gux = 2.f * (y_ * d01 + z_ * d02 + y_ * d10 - 2.f * x_ * d11 - r_ * d12 + z_ * d20 + r_ * d21 -
                               2.f * x_ * d22);
      const float guy = 2.f * (-2.f * y_ * d00 + x_ * d01 + r_ * d02 + x_ * d10 + z_ * d12 - r_ * d20 + z_ * d21 -
                               2.f * y_ * d22);
      const float guz = 2.f * (-2.f * z_ * d00 - r_ * d01 + x_ * d02 + r_ * d10 - 2.f * z_ * d11 + y_ * d12 +
                               x_ * d20 + y_ * d21);
      const float du = r_ * gur + x_ * gux + y_ * guy + z_ * guz;
      gq_w += (gur - r_ * du) * inv, gq_x += (gux - x_ * du) * inv;
      gq_y += (guy - y_ * du) * inv, gq_z += (guz - z_ * du) * inv;
    } else {
      gwt += cp[4] * gp0 + cp[5] * gp1 + cp[6] * gp2;
      cpg[k][1] = w * gp0, cpg[k][2] = w * gp1, cpg[k][3] = w * gp2;
    }
    cpg[k][4] = gq_w, cpg[k][5] = gq_x, cpg[k][6] = gq_y, cpg[k][7] = gq_z;
    gwk[k] = gwt;
    sum_wg += w * gwt;
  }
  // L1 normalisation and radial weight backward -> log radius
#pragma unroll
  for (int k = 0; k < DEF_K; ++k) {
    const float r = s_cp[idx[k] * CP_STRIDE + 3];
    const float g_wt = (W > NORM_EPS) ? (gwk[k] - sum_wg) * invW : 0.f;
    // wt = exp(-d^2/(2 r^2)) + eps ; d(wt)/dr = ex * d^2 / r^3 ; r = exp(lr) -> * r
    const float g_lr = g_wt * ex[k] * dist[k] * dist[k] / (r * r);
    cpg[k][0] = g_lr;
  }
#pragma unroll
  for (int k = 0; k < DEF_K; ++k)
    wave_scatter_add_match8(s_acc + 3, CP_STRIDE, idx[k], cpg[k], valid, lane);
  dxs[0] = dx0, dxs[1] = dx1, dxs[2] = dx2;
}

// Gradient of the control points' positions (columns 0..2 of the LDS table) from the summed dL/dy of columns 4..6:
// sum_i (gy_i - R^T gy_i) = S - R^T S -- R depends on the control point only, so three of the eleven LDS atomics per
// (Gaussian, neighbour) are replaced by one 3x3 product per control point and workgroup (linear in S: exact for
// the per-workgroup partial sums too).  The caller synchronises before and after.
__device__ __forceinline__ void lbs_ctrl_position_grad(int M, const float *s_cp, float *s_acc) {
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const float *cp = s_cp + m * CP_STRIDE;
    float *ac = s_acc + m * CP_STRIDE;
    const float qw = cp[7], qx = cp[8], qy = cp[9], qz = cp[10];
    const float inv = 1.0f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    float R[9];
    quat_R(qw * inv, qx * inv, qy * inv, qz * inv, R);
    const float s0 = ac[4], s1 = ac[5], s2 = ac[6];
    ac[0] = s0 - (R[0] * s0 + R[3] * s1 + R[6] * s2);
    ac[1] = s1 - (R[1] * s0 + R[4] * s1 + R[7] * s2);
    ac[2] = s2 - (R[2] * s0 + R[5] * s1 + R[8] * s2);
  }
}

}  // namespace dimo
