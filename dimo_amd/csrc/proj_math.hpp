// Projection maths of the per-Gaussian kernels (preprocess.hip); a header since round 5's fused backward tail used it
// from a second translation unit (removed: profiles/r05_fused_tail.txt).
// NOTE on floating-point contraction: preprocess.hip is compiled with contraction OFF (tile rects, radii and depth
// bits must match the CPU oracle bit for bit) and states so with a file-level pragma BEFORE including this header; the
// functions below contain no pragma of their own and follow their includer.
#pragma once
#include "common.hpp"

namespace dimo {

__device__ __forceinline__ void xform43(const float *p, const float *m, float *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform44(const float *p, const float *m, float *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
__device__ __forceinline__ void quat_to_R(const float *q, float *R) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.0f - 2.0f * (y * y + z * z);
  R[1] = 2.0f * (x * y - r * z);
  R[2] = 2.0f * (x * z + r * y);
  R[3] = 2.0f * (x * y + r * z);
  R[4] = 1.0f - 2.0f * (x * x + z * z);
  R[5] = 2.0f * (y * z - r * x);
  R[6] = 2.0f * (x * z - r * y);
  R[7] = 2.0f * (y * z + r * x);
  R[8] = 1.0f - 2.0f * (x * x + y * y);
}
__device__ __forceinline__ void cov3d_from_scale_rot(const float *s, float mod, const float *R, float *c) {
  const float s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
  const float M0 = R[0] * s0, M1 = R[1] * s1, M2 = R[2] * s2;
  const float M3 = R[3] * s0, M4 = R[4] * s1, M5 = R[5] * s2;
  const float M6 = R[6] * s0, M7 = R[7] * s1, M8 = R[8] * s2;
  c[0] = M0 * M0 + M1 * M1 + M2 * M2;
  c[1] = M0 * M3 + M1 * M4 + M2 * M5;
  c[2] = M0 * M6 + M1 * M7 + M2 * M8;
  c[3] = M3 * M3 + M4 * M4 + M5 * M5;
  c[4] = M3 * M6 + M4 * M7 + M5 * M8;
  c[5] = M6 * M6 + M7 * M7 + M8 * M8;
}
// T = J * Wm (2x3); Wm = world->view rotation = transpose of V[:3,:3] in row-vector convention
__device__ __forceinline__ void ewa_T(const float *t, float fx, float fy, const float *V, float *T) {
  const float itz = 1.0f / t[2];
  const float j00 = fx * itz;
  const float j02 = -(fx * t[0]) * itz * itz;
  const float j11 = fy * itz;
  const float j12 = -(fy * t[1]) * itz * itz;
  T[0] = j00 * V[0] + j02 * V[2];
  T[1] = j00 * V[4] + j02 * V[6];
  T[2] = j00 * V[8] + j02 * V[10];
  T[3] = j11 * V[1] + j12 * V[2];
  T[4] = j11 * V[5] + j12 * V[6];
  T[5] = j11 * V[9] + j12 * V[10];
}
__device__ __forceinline__ void T_sigma(const float *T, const float *c, float *u) {
  u[0] = T[0] * c[0] + T[1] * c[1] + T[2] * c[2];
  u[1] = T[0] * c[1] + T[1] * c[3] + T[2] * c[4];
  u[2] = T[0] * c[2] + T[1] * c[4] + T[2] * c[5];
  u[3] = T[3] * c[0] + T[4] * c[1] + T[5] * c[2];
  u[4] = T[3] * c[1] + T[4] * c[3] + T[5] * c[4];
  u[5] = T[3] * c[2] + T[4] * c[4] + T[5] * c[5];
}
__device__ __forceinline__ void cov2d_from_T(const float *T, const float *c, float *abc) {
  float u[6];
  T_sigma(T, c, u);
  abc[0] = u[0] * T[0] + u[1] * T[1] + u[2] * T[2] + LOWPASS;
  abc[1] = u[0] * T[3] + u[1] * T[4] + u[2] * T[5];
  abc[2] = u[3] * T[3] + u[4] * T[4] + u[5] * T[5] + LOWPASS;
}

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

// colour of one channel c from SH coefficients sh[k*3 + c]
__device__ __forceinline__ float eval_sh_channel(int deg, const float *sh, int c, float x, float y, float z) {
  float res = SH_C0 * sh[0 * 3 + c];
  if (deg > 0) {
    res = res - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      res = res + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
            SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
            SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
      if (deg > 2) {
        res = res + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] + SH_C3[1] * xy * z * sh[10 * 3 + c] +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] + SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
              SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
      }
    }
  }
  return res;
}

__device__ __forceinline__ int argmin3(const float *s) {
  int k = 0;
  if (s[1] < s[k]) k = 1;
  if (s[2] < s[k]) k = 2;
  return k;
}

// Sum of the flagged instance records [lo, hi) of one Gaussian by its own thread (the records of a Gaussian are
// contiguous: emission order).  GR instances per round, every load of the round issued before the first use: the flags
// first, then the four float4s of each record -- from the record when its flag is set (an instance no pixel reached has
// no record), from ONE dummy line otherwise (a cache hit; an unconditional clamped load instead of a branch around it).
// Same order of additions as a one-at-a-time loop.
__device__ __forceinline__ void sum_instance_records(const SplatGrad *__restrict__ inst_grad,
                                                     const uint8_t *__restrict__ inst_flag, uint32_t lo, uint32_t hi,
                                                     float &m0, float &mx, float &my, float &mxx, float &mxy, float &myy,
                                                     float (&dfeat)[NFEAT]) {
  constexpr int GR = 4;
  const float4 *const dummy = reinterpret_cast<const float4 *>(inst_grad);
  for (uint32_t e0 = lo; e0 < hi; e0 += GR) {
    uint32_t fl[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) fl[k] = inst_flag[min(e0 + k, hi - 1)];
    float4 ra[GR], rb[GR], rc[GR], rd[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) {
      const bool on = e0 + k < hi && fl[k] != 0;
      fl[k] = on;
      const float4 *rp = on ? reinterpret_cast<const float4 *>(inst_grad + e0 + k) : dummy;
      ra[k] = rp[0], rb[k] = rp[1], rc[k] = rp[2], rd[k] = rp[3];
    }
#pragma unroll
    for (int k = 0; k < GR; ++k) {
      if (fl[k]) {
        m0 += ra[k].x, mx += ra[k].y, my += ra[k].z, mxx += ra[k].w;
        mxy += rb[k].x, myy += rb[k].y, dfeat[0] += rb[k].z, dfeat[1] += rb[k].w;
        dfeat[2] += rc[k].x, dfeat[3] += rc[k].y, dfeat[4] += rc[k].z, dfeat[5] += rc[k].w;
        dfeat[6] += rd[k].x;
      }
    }
  }
}
// The same sums from the Gaussian's HIT MASK (bit k = instance lo + k has a record; Gaussians of up to 64 instances):
// no flag bytes are read -- the word arrives with the Gaussian's own inputs -- and only records that exist are fetched,
// four per round in ascending order (the order of additions of the loop above).  In the trained regime one instance
// in nine has a record: the flag loop issued 16 record loads per round whatever the flags said.
__device__ __forceinline__ void sum_hit_records(const SplatGrad *__restrict__ inst_grad, uint32_t lo,
                                                unsigned long long hm, float &m0, float &mx, float &my, float &mxx,
                                                float &mxy, float &myy, float (&dfeat)[NFEAT]) {
  constexpr int GR = 4;
  const float4 *const dummy = reinterpret_cast<const float4 *>(inst_grad);
  while (hm != 0ull) {
    bool on[GR];
    float4 ra[GR], rb[GR], rc[GR], rd[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) {
      on[k] = hm != 0ull;
      const uint32_t e = lo + (uint32_t)(on[k] ? __builtin_ctzll(hm) : 0);
      hm &= hm - 1ull;  // (0 stays 0)
      const float4 *rp = on[k] ? reinterpret_cast<const float4 *>(inst_grad + e) : dummy;
      ra[k] = rp[0], rb[k] = rp[1], rc[k] = rp[2], rd[k] = rp[3];
    }
#pragma unroll
    for (int k = 0; k < GR; ++k) {
      if (on[k]) {
        m0 += ra[k].x, mx += ra[k].y, my += ra[k].z, mxx += ra[k].w;
        mxy += rb[k].x, myy += rb[k].y, dfeat[0] += rb[k].z, dfeat[1] += rb[k].w;
        dfeat[2] += rc[k].x, dfeat[3] += rc[k].y, dfeat[4] += rc[k].z, dfeat[5] += rc[k].w;
        dfeat[6] += rd[k].x;
      }
    }
  }
}
// ... and by a whole WAVE (lanes stride over the records, butterfly reduction): a Gaussian blown up over hundreds of
// tiles would keep its thread in the loop above long after the rest of the chip has finished.  Every lane returns the
// 13 sums in a[].
__device__ __forceinline__ void wave_sum_instance_records(const SplatGrad *__restrict__ inst_grad,
                                                          const uint8_t *__restrict__ inst_flag, uint32_t blo,
                                                          uint32_t bhi, int lane, float (&a)[13]) {
#pragma unroll
  for (int k = 0; k < 13; ++k) a[k] = 0.0f;
  for (uint32_t e = blo + (uint32_t)lane; e < bhi; e += 64) {
    if (inst_flag[e]) {
      const float4 *rp = reinterpret_cast<const float4 *>(inst_grad + e);
      const float4 ra = rp[0], rb = rp[1], rc = rp[2], rd = rp[3];
      a[0] += ra.x, a[1] += ra.y, a[2] += ra.z, a[3] += ra.w, a[4] += rb.x, a[5] += rb.y, a[6] += rb.z;
      a[7] += rb.w, a[8] += rc.x, a[9] += rc.y, a[10] += rc.z, a[11] += rc.w, a[12] += rd.x;
    }
  }
#pragma unroll
  for (int k = 0; k < 13; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
  }
}

// Backward of the projection of ONE visible Gaussian, from the sums of its instance records (moments of g = G dL/dG
// over its tiles' pixels, summed feature gradients) to the gradients of its 3D mean, scale, rotation (or 3D covariance),
// opacity and screen-space mean: conic -> cov2D -> (Sigma, t) -> (scale, quaternion, mean), the perspective projection,
// the depth feature, the normal.  The SH colour backward is the caller's (it differs per entry point).  `o` arrives
// zeroed.
struct ProjGrad {
  float dmean[3], dm2d[2], dop, dsc[3], dq[4], dSig[6];
};
__device__ __forceinline__ void proj_backward_math(int i, int W, int H, float tanfovx, float tanfovy, float scale_mod,
                                                   const Splat &sp, const float (&p)[3], const float (&q)[4],
                                                   const float (&s)[3], const float *__restrict__ cov3D_precomp,
                                                   const float (&V)[16], const float (&P)[16], const float (&cam)[3],
                                                   float m0, float mx, float my, float mxx, float mxy, float myy,
                                                   const float (&dfeat)[NFEAT], ProjGrad &o) {
  float(&dmean)[3] = o.dmean;
  float(&dm2d)[2] = o.dm2d;
  float &dop = o.dop;
  float(&dsc)[3] = o.dsc;
  float(&dq)[4] = o.dq;
  float(&dSig)[6] = o.dSig;
  {
    // moments -> gradients of (pixel mean, conic, opacity)
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    dm2d[0] = -(sp.A * mx + sp.B * my) * ddelx_dx;
    dm2d[1] = -(sp.C * my + sp.B * mx) * ddely_dy;
    const float dLA = -0.5f * mxx, dLB = -mxy, dLC = -0.5f * myy;
    dop = (m0 != 0.0f) ? m0 / sp.opacity : 0.0f;

    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    float c6[6], R[9];
    if (cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
    } else {
      quat_to_R(q, R);
      cov3d_from_scale_rot(s, scale_mod, R, c6);
    }
    float pv[3];
    xform43(p, V, pv);
    float t[3] = {pv[0], pv[1], pv[2]};
    const float limx = FOV_CLAMP * tanfovx, limy = FOV_CLAMP * tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    const float ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    float T[6], abc[3];
    ewa_T(t, fx, fy, V, T);
    cov2d_from_T(T, c6, abc);
    const float a = abc[0], b = abc[1], c = abc[2];
    const float det = a * c - b * b;
    if (det != 0.0f) {
      const float d2 = 1.0f / (det * det + 0.0000001f);
      const float dLa = d2 * (-c * c * dLA + b * c * dLB - b * b * dLC);
      const float dLc = d2 * (-b * b * dLA + a * b * dLB - a * a * dLC);
      const float dLb = d2 * (2.0f * b * c * dLA - (det + 2.0f * b * b) * dLB + 2.0f * a * b * dLC);
      dSig[0] = T[0] * T[0] * dLa + T[0] * T[3] * dLb + T[3] * T[3] * dLc;
      dSig[3] = T[1] * T[1] * dLa + T[1] * T[4] * dLb + T[4] * T[4] * dLc;
      dSig[5] = T[2] * T[2] * dLa + T[2] * T[5] * dLb + T[5] * T[5] * dLc;
      dSig[1] = 2.0f * T[0] * T[1] * dLa + (T[0] * T[4] + T[1] * T[3]) * dLb + 2.0f * T[3] * T[4] * dLc;
      dSig[2] = 2.0f * T[0] * T[2] * dLa + (T[0] * T[5] + T[2] * T[3]) * dLb + 2.0f * T[3] * T[5] * dLc;
      dSig[4] = 2.0f * T[2] * T[1] * dLa + (T[1] * T[5] + T[2] * T[4]) * dLb + 2.0f * T[4] * T[5] * dLc;
      float u[6];
      T_sigma(T, c6, u);
      const float dT0 = 2.0f * dLa * u[0] + dLb * u[3], dT1 = 2.0f * dLa * u[1] + dLb * u[4];
      const float dT2 = 2.0f * dLa * u[2] + dLb * u[5], dT3 = dLb * u[0] + 2.0f * dLc * u[3];
      const float dT4 = dLb * u[1] + 2.0f * dLc * u[4], dT5 = dLb * u[2] + 2.0f * dLc * u[5];
      const float dJ00 = dT0 * V[0] + dT1 * V[4] + dT2 * V[8];
      const float dJ02 = dT0 * V[2] + dT1 * V[6] + dT2 * V[10];
      const float dJ11 = dT3 * V[1] + dT4 * V[5] + dT5 * V[9];
      const float dJ12 = dT3 * V[2] + dT4 * V[6] + dT5 * V[10];
      const float tz = 1.0f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      const float dtx = xmul * -fx * tz2 * dJ02;
      const float dty = ymul * -fy * tz2 * dJ12;
      const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * t[0]) * tz3 * dJ02 +
                        (2.0f * fy * t[1]) * tz3 * dJ12;
      dmean[0] += V[0] * dtx + V[1] * dty + V[2] * dtz;
      dmean[1] += V[4] * dtx + V[5] * dty + V[6] * dtz;
      dmean[2] += V[8] * dtx + V[9] * dty + V[10] * dtz;
    }
    {  // NDC mean -> world mean through the projection
      float ph[4];
      xform44(p, P, ph);
      const float mw = 1.0f / (ph[3] + W_EPS);
      const float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
      dmean[0] += (P[0] * mw - P[3] * mul1) * dm2d[0] + (P[1] * mw - P[3] * mul2) * dm2d[1];
      dmean[1] += (P[4] * mw - P[7] * mul1) * dm2d[0] + (P[5] * mw - P[7] * mul2) * dm2d[1];
      dmean[2] += (P[8] * mw - P[11] * mul1) * dm2d[0] + (P[9] * mw - P[11] * mul2) * dm2d[1];
    }
    // depth feature
    dmean[0] += V[2] * dfeat[3], dmean[1] += V[6] * dfeat[3], dmean[2] += V[10] * dfeat[3];
    // Sigma -> scale, rotation ; normal -> rotation
    if (!cov3D_precomp) {
      const float sm[3] = {scale_mod * s[0], scale_mod * s[1], scale_mod * s[2]};
      const float Gm[9] = {dSig[0], 0.5f * dSig[1], 0.5f * dSig[2], 0.5f * dSig[1], dSig[3],
                           0.5f * dSig[4], 0.5f * dSig[2], 0.5f * dSig[4], dSig[5]};
      float dR[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float dMk[3];
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_) {
          float acc = 0.0f;
#pragma unroll
          for (int j = 0; j < 3; ++j) acc += Gm[r_ * 3 + j] * (R[j * 3 + k] * sm[k]);
          dMk[r_] = 2.0f * acc;
        }
        dsc[k] = (dMk[0] * R[0 * 3 + k] + dMk[1] * R[1 * 3 + k] + dMk[2] * R[2 * 3 + k]) * scale_mod;
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_) dR[r_ * 3 + k] = dMk[r_] * sm[k];
      }
      {
        const int k = argmin3(s);
        const float n0 = R[0 + k], n1 = R[3 + k], n2 = R[6 + k];
        const float dot = n0 * (cam[0] - p[0]) + n1 * (cam[1] - p[1]) + n2 * (cam[2] - p[2]);
        const float sgn = dot < 0.0f ? -1.0f : 1.0f;
        const float g0 = dfeat[4], g1 = dfeat[5], g2 = dfeat[6];
        const float wn[3] = {V[0] * g0 + V[1] * g1 + V[2] * g2, V[4] * g0 + V[5] * g1 + V[6] * g2,
                             V[8] * g0 + V[9] * g1 + V[10] * g2};
#pragma unroll
        for (int r_ = 0; r_ < 3; ++r_) {
          // static index selection keeps dR in registers
          if (k == 0) dR[r_ * 3 + 0] += sgn * wn[r_];
          else if (k == 1) dR[r_ * 3 + 1] += sgn * wn[r_];
          else dR[r_ * 3 + 2] += sgn * wn[r_];
        }
      }
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      dq[0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      dq[1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] -
                      2.0f * x * dR[8]);
      dq[2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] -
                      2.0f * y * dR[8]);
      dq[3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] +
                      x * dR[6] + y * dR[7]);
    }
  }
}

}  // namespace dimo
