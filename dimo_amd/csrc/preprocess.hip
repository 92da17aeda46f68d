// Per-Gaussian stages of the rasterizer: projection forward and its backward.
// One thread per Gaussian, 256 per block; pure streaming (HBM-bound) kernels.
//
// This translation unit is compiled with -ffp-contract=off: tile rects, radii and the fp32
// depth bits that form the sort keys must match the CPU oracle bit for bit, so every
// multiply-add below is evaluated unfused, left to right, exactly as written.
#pragma clang fp contract(off)
#include "proj_math.hpp"

namespace dimo {


__device__ __forceinline__ void preprocess_fwd_body(
    int N, int deg, int M, int H, int W, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ colors_precomp, const float *__restrict__ opacities, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, float scale_mod,
    const float *__restrict__ Vg, const float *__restrict__ Pg, const float *__restrict__ camg, float tanfovx,
    float tanfovy, int32_t *__restrict__ radii, Splat *__restrict__ splat, uint16_t *__restrict__ rect,
    uint32_t *__restrict__ tiles_touched, uint8_t *__restrict__ flags, uint32_t *__restrict__ block_sums,
    uint32_t *__restrict__ key32, uint32_t *__restrict__ bk, int ss_shift) {
  __shared__ uint32_t wave_sums[PRE_BLOCK / 64], wave_min[PRE_BLOCK / 64], wave_max[PRE_BLOCK / 64];
  __shared__ uint32_t wave_ent[PRE_BLOCK / 64];
  const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
  // the binning's per-bucket entry counters start every frame at zero: cleared here, one launch ahead of
  // the level-1 kernel that adds to them (binning.hip)
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t < 2 * MAX_BUCKETS; t += PRE_BLOCK) bk[BK_TOT + t] = 0u;  // (64-bit words)
    if (threadIdx.x == 0) bk[BK_DONE] = 0u, bk[BK_OVF] = 0u;  // (level-1 workgroups done; overflow cursor)
  }
  // camera: uniform loads (scalar cache)
  float V[16], P[16], cam[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) V[k] = Vg[k], P[k] = Pg[k];
  cam[0] = camg[0], cam[1] = camg[1], cam[2] = camg[2];

  uint32_t my_tiles = 0, my_key = 0xffffffffu;  // (key = depth bits of a Gaussian that touches a tile)
  uint32_t my_ent = 0;                          // supertiles it touches = its level-1 entries (binning.hip)
  if (i < N) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    Splat out = {};
    int my_radius = 0;
    uint16_t rc[4] = {0, 0, 0, 0};
    uint8_t fl = 0;
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
    xform43(p, V, pv);
    if (pv[2] > NEAR_CULL) {
      float ph[4];
      xform44(p, P, ph);
      const float pw = 1.0f / (ph[3] + W_EPS);
      const float px = ph[0] * pw, py = ph[1] * pw;
      float c6[6];
      float R[9];
      if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
      } else {
        const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        quat_to_R(q, R);
        cov3d_from_scale_rot(s, scale_mod, R, c6);
      }
      float t[3] = {pv[0], pv[1], pv[2]};
      const float limx = FOV_CLAMP * tanfovx, limy = FOV_CLAMP * tanfovy;
      const float txtz = t[0] / t[2], tytz = t[1] / t[2];
      t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
      t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
      float T[6], abc[3];
      ewa_T(t, fx, fy, V, T);
      cov2d_from_T(T, c6, abc);
      const float det = abc[0] * abc[2] - abc[1] * abc[1];
      if (det != 0.0f) {
        const float det_inv = 1.0f / det;
        const float cA = abc[2] * det_inv, cB = -abc[1] * det_inv, cC = abc[0] * det_inv;
        const float mid = 0.5f * (abc[0] + abc[2]);
        const float disc = sqrtf(fmaxf(LAMBDA_FLOOR, mid * mid - det));
        const float lam1 = mid + disc, lam2 = mid - disc;
        const float rad = ceilf(RADIUS_SIGMA * sqrtf(fmaxf(lam1, lam2)));
        const int rr = (int)rad;
        const float pix_x = ((px + 1.0f) * (float)W - 1.0f) * 0.5f;
        const float pix_y = ((py + 1.0f) * (float)H - 1.0f) * 0.5f;
        const int rx0 = min(gx, max(0, (int)((pix_x - (float)rr) / (float)TILE)));
        const int ry0 = min(gy, max(0, (int)((pix_y - (float)rr) / (float)TILE)));
        const int rx1 = min(gx, max(0, (int)((pix_x + (float)rr + (float)(TILE - 1)) / (float)TILE)));
        const int ry1 = min(gy, max(0, (int)((pix_y + (float)rr + (float)(TILE - 1)) / (float)TILE)));
        const int cnt = (rx1 - rx0) * (ry1 - ry0);
        if (cnt != 0) {
          float rgb[3];
          if (colors_precomp) {
            rgb[0] = colors_precomp[3 * i], rgb[1] = colors_precomp[3 * i + 1], rgb[2] = colors_precomp[3 * i + 2];
          } else {
            const float d[3] = {p[0] - cam[0], p[1] - cam[1], p[2] - cam[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float dx = d[0] / len, dy = d[1] / len, dz = d[2] / len;
            const float *sh = shs + (size_t)i * M * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float v = eval_sh_channel(deg, sh, c, dx, dy, dz) + 0.5f;
              if (v < 0.0f) fl |= (uint8_t)(1u << c);
              rgb[c] = fmaxf(v, 0.0f);
            }
          }
          float nv[3] = {0.0f, 0.0f, 0.0f};
          if (!cov3D_precomp) {
            const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
            const int k = argmin3(s);
            float n[3] = {R[0 + k], R[3 + k], R[6 + k]};
            const float dot = n[0] * (cam[0] - p[0]) + n[1] * (cam[1] - p[1]) + n[2] * (cam[2] - p[2]);
            const float sgn = dot < 0.0f ? -1.0f : 1.0f;
            n[0] *= sgn, n[1] *= sgn, n[2] *= sgn;
            nv[0] = V[0] * n[0] + V[4] * n[1] + V[8] * n[2];
            nv[1] = V[1] * n[0] + V[5] * n[1] + V[9] * n[2];
            nv[2] = V[2] * n[0] + V[6] * n[1] + V[10] * n[2];
          }
          my_radius = rr;
          my_tiles = (uint32_t)cnt;
          my_key = __float_as_uint(pv[2]);
          my_ent = (uint32_t)((((rx1 - 1) >> ss_shift) - (rx0 >> ss_shift) + 1) *
                              (((ry1 - 1) >> ss_shift) - (ry0 >> ss_shift) + 1));
          out.x = pix_x, out.y = pix_y, out.A = cA, out.B = cB, out.C = cC, out.opacity = opacities[i];
          out.r = rgb[0], out.g = rgb[1], out.b = rgb[2], out.depth = pv[2];
          out.nx = nv[0], out.ny = nv[1], out.nz = nv[2];
          out.As = CONIC_HALF * cA, out.Bs = CONIC_FULL * cB, out.Cs = CONIC_HALF * cC;
          rc[0] = (uint16_t)rx0, rc[1] = (uint16_t)ry0, rc[2] = (uint16_t)rx1, rc[3] = (uint16_t)ry1;
        }
      }
    }
    radii[i] = my_radius;
    tiles_touched[i] = my_tiles;
    key32[i] = my_key;
    flags[i] = fl;
    // 64-B record and 8-B rect as wide stores
    float4 *dst = reinterpret_cast<float4 *>(splat + i);
    const float4 *src = reinterpret_cast<const float4 *>(&out);
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
    *reinterpret_cast<uint2 *>(rect + 4 * (size_t)i) =
        make_uint2((uint32_t)rc[0] | ((uint32_t)rc[1] << 16), (uint32_t)rc[2] | ((uint32_t)rc[3] << 16));
  }
  // block sum of tiles_touched (feeds the 2-level scan)
  uint32_t v = my_tiles, ve = my_ent;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64), ve += __shfl_down(ve, o, 64);
  if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = v, wave_ent[threadIdx.x >> 6] = ve;
  // ... and the block's range of depth-sort keys (the depth bits of the Gaussians that touch a tile): the sort cuts
  // [min, max] into buckets (binning.hip)
  uint32_t kmn = my_key, kmx = my_key == 0xffffffffu ? 0u : my_key;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmn = min(kmn, (uint32_t)__shfl_down((int)kmn, o, 64));
    kmx = max(kmx, (uint32_t)__shfl_down((int)kmx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) wave_min[threadIdx.x >> 6] = kmn, wave_max[threadIdx.x >> 6] = kmx;
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    block_sums[(gridDim.x + 1) + blockIdx.x] = min(min(wave_min[0], wave_min[1]), min(wave_min[2], wave_min[3]));
    block_sums[2 * (gridDim.x + 1) + blockIdx.x] = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
    block_sums[3 * (gridDim.x + 1) + blockIdx.x] = wave_ent[0] + wave_ent[1] + wave_ent[2] + wave_ent[3];
  }
}

// ----------------------------------------------------------------------------------------------
// Backward: sums the per-instance records of the blend backward (contiguous per Gaussian, indexed
// by emission position -> deterministic, atomic-free), then differentiates conic -> cov2D ->
// (Sigma, t) -> (scale, quaternion, mean), the perspective projection, depth, SH colour, normal.
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_fwd_kernel(
    int N, int deg, int M, int H, int W, const float *__restrict__ means3D, const float *__restrict__ shs,
    const float *__restrict__ colors_precomp, const float *__restrict__ opacities, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, float scale_mod,
    const float *__restrict__ Vg, const float *__restrict__ Pg, const float *__restrict__ camg, float tanfovx,
    float tanfovy, int32_t *__restrict__ radii, Splat *__restrict__ splat, uint16_t *__restrict__ rect,
    uint32_t *__restrict__ tiles_touched, uint8_t *__restrict__ flags, uint32_t *__restrict__ block_sums,
    uint32_t *__restrict__ key32, uint32_t *__restrict__ bk, int ss_shift) {
  preprocess_fwd_body(N, deg, M, H, W, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                      scale_mod, Vg, Pg, camg, tanfovx, tanfovy, radii, splat, rect, tiles_touched, flags, block_sums,
                      key32, bk, ss_shift);
}
// total[0] = R = sum of the blocks' tile counts: only for callers that size their buffers from a read-back of R
// (dimo_raster_preprocess_forward with R_host); the binning's level-1 kernel writes the same word
__global__ void __launch_bounds__(256) total_instances_kernel(int nb, const uint32_t *__restrict__ sums,
                                                              uint32_t *__restrict__ total) {
  __shared__ uint32_t part[4];
  uint32_t v = 0;
  for (int i = threadIdx.x; i < nb; i += 256) v += sums[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) total[0] = part[0] + part[1] + part[2] + part[3], total[1] = 0, total[2] = 0, total[3] = 0;
}
// blockIdx.y = render of the batch: degree-0 colour from the shared f_dc, scale/rotation covariance
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_fwd_batched_kernel(int N, int H, int W,
                                                                           const float *__restrict__ f_dc,
                                                                           float scale_mod, GeomLayout L,
                                                                           int ss_shift, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  preprocess_fwd_body(N, 0, 1, H, W, r.pts, f_dc, nullptr, r.opac, r.scales, r.rot, nullptr, scale_mod, r.view, r.proj,
                      r.campos, r.tanfovx, r.tanfovy, r.radii, at<Splat>(r.geom, L.splat), at<uint16_t>(r.geom, L.rect),
                      at<uint32_t>(r.geom, L.tiles), at<uint8_t>(r.geom, L.flags), at<uint32_t>(r.geom, L.block_sums),
                      at<uint32_t>(r.geom, L.key32), at<uint32_t>(r.geom, L.bk), ss_shift);
}

constexpr uint32_t PRESUM_MIN = 64;   // instances above which a Gaussian's records are summed cooperatively
constexpr int PRESUM_MAX_N = 8192;    // Gaussians up to which that is a launch of its own (one workgroup per Gaussian)
__device__ __forceinline__ void preprocess_bwd_body(
    int N, int deg, int M, int H, int W, uint32_t R_cap, const float *__restrict__ means3D,
    const float *__restrict__ shs, const float *__restrict__ colors_precomp, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, float scale_mod,
    const float *__restrict__ Vg, const float *__restrict__ Pg, const float *__restrict__ camg, float tanfovx,
    float tanfovy, const int32_t *__restrict__ radii, const Splat *__restrict__ splat,
    const uint32_t *__restrict__ offsets, const uint8_t *__restrict__ flags, const SplatGrad *__restrict__ inst_grad,
    const uint8_t *__restrict__ inst_flag, const unsigned long long *__restrict__ hitmask,
    float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dshs,
    float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacity, float *__restrict__ dL_dscales,
    float *__restrict__ dL_drot, float *__restrict__ dL_dcov3D, const float *__restrict__ presum = nullptr) {
  // A Gaussian with MANY instances (a splat blown up over hundreds of tiles: random targets produce them within a
  // few hundred steps) would keep its thread in the gather loop long after the rest of the chip has finished --
  // the kernel went from 80 to 230 us per 8 renders over a 1400-step run.  Such Gaussians are summed by a whole WAVE
  // (lanes stride over the records, butterfly reduction) before the per-thread part; everything else is unchanged.
  // `presum` (stage-s1 shapes: a few hundred Gaussians, every one of them over hundreds of tiles): the sums of the big
  // Gaussians were formed by instance_sums_batched_kernel, one workgroup per Gaussian, [16 N] floats.
  constexpr uint32_t BIG = PRESUM_MIN;  // instances above which the wave takes over
  __shared__ uint32_t s_big_n;
  __shared__ uint32_t s_big_lo[PRE_BLOCK], s_big_hi[PRE_BLOCK];
  __shared__ float s_big_sum[PRE_BLOCK][13];
  const int i_raw = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const bool valid = i_raw < N;
  const int i = valid ? i_raw : (N > 0 ? N - 1 : 0);
  float V[16], P[16], cam[3];
#pragma unroll
  for (int k = 0; k < 16; ++k) V[k] = Vg[k], P[k] = Pg[k];
  cam[0] = camg[0], cam[1] = camg[1], cam[2] = camg[2];

  float dmean[3] = {0, 0, 0}, dm2d[2] = {0, 0}, dop = 0.0f;
  float dsc[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0}, dSig[6] = {0, 0, 0, 0, 0, 0};
  float dfeat[NFEAT] = {0, 0, 0, 0, 0, 0, 0};
  const bool visible = valid && N > 0 && radii[i] > 0;
  // the Gaussian's own inputs are requested before the gather loop (they are only needed after it)
  uint32_t lo = N > 0 ? offsets[i == 0 ? 0 : i - 1] : 0u, hi = N > 0 ? offsets[i] : 0u;
  const unsigned long long hm = N > 0 ? hitmask[i] : 0ull;
  const Splat sp = N > 0 ? splat[i] : Splat{};
  const float p[3] = {N > 0 ? means3D[3 * i] : 0.f, N > 0 ? means3D[3 * i + 1] : 0.f, N > 0 ? means3D[3 * i + 2] : 0.f};
  float q[4] = {1, 0, 0, 0}, s[3] = {0, 0, 0};
  if (!cov3D_precomp && N > 0) {  // (kernel-uniform)
    q[0] = rotations[4 * i], q[1] = rotations[4 * i + 1], q[2] = rotations[4 * i + 2], q[3] = rotations[4 * i + 3];
    s[0] = scales[3 * i], s[1] = scales[3 * i + 1], s[2] = scales[3 * i + 2];
  }
  lo = i == 0 ? 0u : lo;
  const uint32_t n_inst = hi - lo;  // (before the capacity clamp: what the blend backward's choice of mask / flags went by)
  lo = min(lo, R_cap), hi = min(hi, R_cap);
  if (threadIdx.x == 0) s_big_n = 0u;
  __syncthreads();
  const bool big = visible && hi > lo && n_inst > BIG;
  uint32_t big_slot = 0;
  if (big && !presum) {
    big_slot = atomicAdd(&s_big_n, 1u);
    s_big_lo[big_slot] = lo, s_big_hi[big_slot] = hi;
  }
  __syncthreads();
  if (!presum) {
    const uint32_t n_big = s_big_n;  // (workgroup-uniform)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b = (uint32_t)wave; b < n_big; b += PRE_BLOCK / 64) {
      const uint32_t blo = s_big_lo[b], bhi = s_big_hi[b];
      float a[13];
      wave_sum_instance_records(inst_grad, inst_flag, blo, bhi, lane, a);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 13; ++k) s_big_sum[b][k] = a[k];
      }
    }
  }
  __syncthreads();
  if (!valid) return;

  if (visible) {
    // ---- gather the instance records
    float m0 = 0, mx = 0, my = 0, mxx = 0, mxy = 0, myy = 0;
    if (big) {
      const float *a = presum ? presum + 16 * (size_t)i : s_big_sum[big_slot];
      m0 = a[0], mx = a[1], my = a[2], mxx = a[3], mxy = a[4], myy = a[5];
#pragma unroll
      for (int k = 0; k < NFEAT; ++k) dfeat[k] = a[6 + k];
      lo = hi;  // (nothing left for the loop below)
    }
    // Several instances per round, every load of the round issued before the first use: the flags first, then the four
    // float4s of each record -- from the record when its flag is set (an instance no pixel reached has no record),
    // from ONE dummy line otherwise (a cache hit; an unconditional clamped load instead of a branch around it).
    // The one-entry-at-a-time loop was two dependent memory round trips per instance with ~1.5 waves per SIMD to
    // hide them: 24 us per wave of pure latency.
    // (FOUR per round: eight held 104 registers of records in flight, 149 VGPRs = three waves per SIMD, and a stamp per
    // workgroup showed 1.65 workgroups per CU alive on average; four: 122 VGPRs, four waves per SIMD, the same number
    // of loads in flight per SIMD -- 100 -> 94 us per 8 renders in the timed step.)
    // (a Gaussian of up to 64 instances -- every one the wave or the presum did not take -- is summed from its hit mask;
    // the mask only names instances below the capacity: the blend backward set no bit beyond it)
    if (hi > lo) sum_hit_records(inst_grad, lo, hm, m0, mx, my, mxx, mxy, myy, dfeat);
    // conic -> cov2D -> (Sigma, t) -> (scale, quaternion, mean), projection, depth feature, normal (proj_math.hpp)
    ProjGrad pg = {};
    proj_backward_math(i, W, H, tanfovx, tanfovy, scale_mod, sp, p, q, s, cov3D_precomp, V, P, cam, m0, mx, my, mxx, mxy,
                       myy, dfeat, pg);
#pragma unroll
    for (int k = 0; k < 3; ++k) dmean[k] = pg.dmean[k], dsc[k] = pg.dsc[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) dq[k] = pg.dq[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) dSig[k] = pg.dSig[k];
    dm2d[0] = pg.dm2d[0], dm2d[1] = pg.dm2d[1], dop = pg.dop;

    // colour
    if (!colors_precomp) {
      const float d[3] = {p[0] - cam[0], p[1] - cam[1], p[2] - cam[2]};
      const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const float x = d[0] / len, y = d[1] / len, z = d[2] / len;
      float ddir[3] = {0, 0, 0};
      const float *sh = shs + (size_t)i * M * 3;
      float *dsh = dL_dshs + (size_t)i * M * 3;
      const uint8_t fl = flags[i];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const float g = ((fl >> cc) & 1) ? 0.0f : dfeat[cc];
        float dx_ = 0, dy_ = 0, dz_ = 0;
        dsh[0 * 3 + cc] = SH_C0 * g;
        if (deg > 0) {
          dsh[1 * 3 + cc] = -SH_C1 * y * g;
          dsh[2 * 3 + cc] = SH_C1 * z * g;
          dsh[3 * 3 + cc] = -SH_C1 * x * g;
          dx_ = -SH_C1 * sh[3 * 3 + cc];
          dy_ = -SH_C1 * sh[1 * 3 + cc];
          dz_ = SH_C1 * sh[2 * 3 + cc];
          if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dsh[4 * 3 + cc] = SH_C2[0] * xy * g;
            dsh[5 * 3 + cc] = SH_C2[1] * yz * g;
            dsh[6 * 3 + cc] = SH_C2[2] * (2.0f * zz - xx - yy) * g;
            dsh[7 * 3 + cc] = SH_C2[3] * xz * g;
            dsh[8 * 3 + cc] = SH_C2[4] * (xx - yy) * g;
            dx_ += SH_C2[0] * y * sh[4 * 3 + cc] + SH_C2[2] * 2.0f * -x * sh[6 * 3 + cc] +
                   SH_C2[3] * z * sh[7 * 3 + cc] + SH_C2[4] * 2.0f * x * sh[8 * 3 + cc];
            dy_ += SH_C2[0] * x * sh[4 * 3 + cc] + SH_C2[1] * z * sh[5 * 3 + cc] +
                   SH_C2[2] * 2.0f * -y * sh[6 * 3 + cc] + SH_C2[4] * 2.0f * -y * sh[8 * 3 + cc];
            dz_ += SH_C2[1] * y * sh[5 * 3 + cc] + SH_C2[2] * 2.0f * 2.0f * z * sh[6 * 3 + cc] +
                   SH_C2[3] * x * sh[7 * 3 + cc];
            if (deg > 2) {
              dsh[9 * 3 + cc] = SH_C3[0] * y * (3.0f * xx - yy) * g;
              dsh[10 * 3 + cc] = SH_C3[1] * xy * z * g;
              dsh[11 * 3 + cc] = SH_C3[2] * y * (4.0f * zz - xx - yy) * g;
              dsh[12 * 3 + cc] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * g;
              dsh[13 * 3 + cc] = SH_C3[4] * x * (4.0f * zz - xx - yy) * g;
              dsh[14 * 3 + cc] = SH_C3[5] * z * (xx - yy) * g;
              dsh[15 * 3 + cc] = SH_C3[6] * x * (xx - 3.0f * yy) * g;
              dx_ += SH_C3[0] * sh[9 * 3 + cc] * 3.0f * 2.0f * xy + SH_C3[1] * sh[10 * 3 + cc] * yz +
                     SH_C3[2] * sh[11 * 3 + cc] * -2.0f * xy + SH_C3[3] * sh[12 * 3 + cc] * -3.0f * 2.0f * xz +
                     SH_C3[4] * sh[13 * 3 + cc] * (-3.0f * xx + 4.0f * zz - yy) +
                     SH_C3[5] * sh[14 * 3 + cc] * 2.0f * xz + SH_C3[6] * sh[15 * 3 + cc] * 3.0f * (xx - yy);
              dy_ += SH_C3[0] * sh[9 * 3 + cc] * 3.0f * (xx - yy) + SH_C3[1] * sh[10 * 3 + cc] * xz +
                     SH_C3[2] * sh[11 * 3 + cc] * (-3.0f * yy + 4.0f * zz - xx) +
                     SH_C3[3] * sh[12 * 3 + cc] * -3.0f * 2.0f * yz + SH_C3[4] * sh[13 * 3 + cc] * -2.0f * xy +
                     SH_C3[5] * sh[14 * 3 + cc] * -2.0f * yz + SH_C3[6] * sh[15 * 3 + cc] * -3.0f * 2.0f * xy;
              dz_ += SH_C3[1] * sh[10 * 3 + cc] * xy + SH_C3[2] * sh[11 * 3 + cc] * 4.0f * 2.0f * yz +
                     SH_C3[3] * sh[12 * 3 + cc] * 3.0f * (2.0f * zz - xx - yy) +
                     SH_C3[4] * sh[13 * 3 + cc] * 4.0f * 2.0f * xz + SH_C3[5] * sh[14 * 3 + cc] * (xx - yy);
            }
          }
        }
        ddir[0] += dx_ * g, ddir[1] += dy_ * g, ddir[2] += dz_ * g;
      }
      if (deg > 0) {
        const float sum2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((sum2 - d[0] * d[0]) * ddir[0] - d[1] * d[0] * ddir[1] - d[2] * d[0] * ddir[2]) * inv32;
        dmean[1] += (-d[0] * d[1] * ddir[0] + (sum2 - d[1] * d[1]) * ddir[1] - d[2] * d[1] * ddir[2]) * inv32;
        dmean[2] += (-d[0] * d[2] * ddir[0] - d[1] * d[2] * ddir[1] + (sum2 - d[2] * d[2]) * ddir[2]) * inv32;
      }
    }

  } else if (!colors_precomp) {
    float *dsh = dL_dshs + (size_t)i * M * 3;
    for (int k = 0; k < M * 3; ++k) dsh[k] = 0.0f;
  }

  dL_dmeans3D[3 * i] = dmean[0], dL_dmeans3D[3 * i + 1] = dmean[1], dL_dmeans3D[3 * i + 2] = dmean[2];
  dL_dmeans2D[3 * i] = dm2d[0], dL_dmeans2D[3 * i + 1] = dm2d[1], dL_dmeans2D[3 * i + 2] = 0.0f;
  dL_dopacity[i] = dop;
  if (colors_precomp) {
    dL_dcolors[3 * i] = dfeat[0], dL_dcolors[3 * i + 1] = dfeat[1], dL_dcolors[3 * i + 2] = dfeat[2];
  } else if (visible) {
    float *dsh = dL_dshs + (size_t)i * M * 3;
    const int used = (deg + 1) * (deg + 1);
    for (int k = used * 3; k < M * 3; ++k) dsh[k] = 0.0f;  // coefficients above the active degree
  }
  if (cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = dSig[k];
  } else {
    dL_dscales[3 * i] = dsc[0], dL_dscales[3 * i + 1] = dsc[1], dL_dscales[3 * i + 2] = dsc[2];
    dL_drot[4 * i] = dq[0], dL_drot[4 * i + 1] = dq[1], dL_drot[4 * i + 2] = dq[2], dL_drot[4 * i + 3] = dq[3];
  }
}

__global__ void __launch_bounds__(PRE_BLOCK) preprocess_bwd_kernel(
    int N, int deg, int M, int H, int W, uint32_t R_cap, const float *__restrict__ means3D,
    const float *__restrict__ shs, const float *__restrict__ colors_precomp, const float *__restrict__ scales,
    const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp, float scale_mod,
    const float *__restrict__ Vg, const float *__restrict__ Pg, const float *__restrict__ camg, float tanfovx,
    float tanfovy, const int32_t *__restrict__ radii, const Splat *__restrict__ splat,
    const uint32_t *__restrict__ offsets, const uint8_t *__restrict__ flags, const SplatGrad *__restrict__ inst_grad,
    const uint8_t *__restrict__ inst_flag, const unsigned long long *__restrict__ hitmask,
    float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
    float *__restrict__ dL_dshs, float *__restrict__ dL_dcolors, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dscales, float *__restrict__ dL_drot, float *__restrict__ dL_dcov3D) {
  preprocess_bwd_body(N, deg, M, H, W, R_cap, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, scale_mod,
                      Vg, Pg, camg, tanfovx, tanfovy, radii, splat, offsets, flags, inst_grad, inst_flag, hitmask,
                      dL_dmeans3D,
                      dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacity, dL_dscales, dL_drot, dL_dcov3D);
}
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_bwd_batched_kernel(int N, int H, int W, uint32_t R_cap,
                                                                           const float *__restrict__ f_dc,
                                                                           float scale_mod, GeomLayout L,
                                                                           size_t flag_offset, size_t presum_offset,
                                                                           RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  preprocess_bwd_body(N, 0, 1, H, W, R_cap, r.pts, f_dc, nullptr, r.scales, r.rot, nullptr, scale_mod, r.view, r.proj,
                      r.campos, r.tanfovx, r.tanfovy, r.radii, at<Splat>(r.geom, L.splat),
                      at<uint32_t>(r.geom, L.offsets), at<uint8_t>(r.geom, L.flags),
                      reinterpret_cast<const SplatGrad *>(r.bwd_scratch), at<uint8_t>(r.bwd_scratch, flag_offset),
                      at<unsigned long long>(r.geom, L.hitmask), r.g_means3D, r.g_means2D, r.g_shs, nullptr, r.g_opac, r.g_scales, r.g_rot, nullptr,
                      presum_offset != ~(size_t)0 ? at<float>(r.bin, presum_offset) : nullptr);
}

// Stage-s1 shapes (main_train_dimo.py: 512 Gaussians of radius exp(_r), each over hundreds of tiles): the projection
// backward is 2 workgroups per render there and its 8 waves walked ~10^6 instance records per render between them
// (229 us per 8 renders).  One WORKGROUP per Gaussian sums the records first -- four records of a thread in flight,
// waves and lanes reduced in a fixed order -- into [16 N] floats that live in the binning's fallback-sort scratch
// (free during the backward).
__global__ void __launch_bounds__(256) instance_sums_batched_kernel(int N, uint32_t R_cap, GeomLayout L, size_t flag_offset,
                                                                    size_t presum_offset, RenderBatch b) {
  __shared__ float s_part[4][13];
  const dimo_render_desc &r = b.r[blockIdx.y];
  const int g = blockIdx.x;
  const uint32_t *__restrict__ offsets = at<uint32_t>(r.geom, L.offsets);
  const uint32_t lo_raw = g == 0 ? 0u : offsets[g - 1], hi_raw = offsets[g];
  const uint32_t lo = min(lo_raw, R_cap), hi = min(hi_raw, R_cap);
  if (hi <= lo || hi_raw - lo_raw <= PRESUM_MIN) return;  // (workgroup-uniform; the count BEFORE the capacity clamp: preprocess_bwd_body)
  const SplatGrad *__restrict__ inst_grad = reinterpret_cast<const SplatGrad *>(r.bwd_scratch);
  const uint8_t *__restrict__ inst_flag = at<uint8_t>(r.bwd_scratch, flag_offset);
  const float4 *const dummy = reinterpret_cast<const float4 *>(inst_grad);
  float a[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int GR = 4;
  for (uint32_t e0 = lo + threadIdx.x; e0 < hi; e0 += GR * 256) {
    uint32_t fl[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) fl[k] = inst_flag[min(e0 + k * 256, hi - 1)];
    float4 ra[GR], rb[GR], rc[GR], rd[GR];
#pragma unroll
    for (int k = 0; k < GR; ++k) {
      const bool on = e0 + k * 256 < hi && fl[k] != 0;
      fl[k] = on;
      const float4 *rp = on ? reinterpret_cast<const float4 *>(inst_grad + e0 + k * 256) : dummy;
      ra[k] = rp[0], rb[k] = rp[1], rc[k] = rp[2], rd[k] = rp[3];
    }
#pragma unroll
    for (int k = 0; k < GR; ++k)
      if (fl[k]) {
        a[0] += ra[k].x, a[1] += ra[k].y, a[2] += ra[k].z, a[3] += ra[k].w, a[4] += rb[k].x, a[5] += rb[k].y;
        a[6] += rb[k].z, a[7] += rb[k].w, a[8] += rc[k].x, a[9] += rc[k].y, a[10] += rc[k].z, a[11] += rc[k].w;
        a[12] += rd[k].x;
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 13; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
    if (lane == 0) s_part[wave][k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x < 13)
    at<float>(r.bin, presum_offset)[16 * (size_t)g + threadIdx.x] =
        ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
}

// scan of tiles_touched for every render of a batch (binning.hip)

int preprocess_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout L(c.N);
  if (c.geom_bytes < L.bytes) return DIMO_E_WORKSPACE;
  BinGrid gi;
  if (!make_bin_grid(c.H, c.W, gi)) return DIMO_E_ARG;  // more than MAX_SUPER * 64 tiles
  // (N = 0 still runs one block: it clears the binning's counters)
  const int nb = c.N > 0 ? (c.N + PRE_BLOCK - 1) / PRE_BLOCK : 1;
  ScopedTimer tm(T_PREPROCESS_FWD, stream);
  hipLaunchKernelGGL(preprocess_fwd_batched_kernel, dim3(nb, n), dim3(PRE_BLOCK), 0, stream, c.N, c.H, c.W, c.f_dc,
                     c.scale_modifier, L, gi.ss_shift, b);
  return check_launch();
}

int preprocess_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  const int nb = (c.N + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb == 0 || n <= 0) return DIMO_OK;
  GeomLayout L(c.N);
  const uint32_t cap = (uint32_t)(c.R_cap > 0xffffffffLL ? 0xffffffffu : (uint32_t)c.R_cap);
  const size_t flag_offset = align_up((size_t)(c.R_cap > 0 ? c.R_cap : 1) * sizeof(SplatGrad));
  // few Gaussians (stage s1): their instance records are summed one workgroup per Gaussian first
  BinLayout B(c.R_cap, c.H, c.W, c.N);
  const bool presum = c.N <= PRESUM_MAX_N && B.l1b - B.l1a >= 64 * (size_t)c.N && c.bin_bytes >= B.bytes;
  ScopedTimer tm(T_PREPROCESS_BWD, stream);
  if (presum)
    hipLaunchKernelGGL(instance_sums_batched_kernel, dim3(c.N, n), dim3(256), 0, stream, c.N, cap, L, flag_offset,
                       B.l1a, b);
  hipLaunchKernelGGL(preprocess_bwd_batched_kernel, dim3(nb, n), dim3(PRE_BLOCK), 0, stream, c.N, c.H, c.W, cap,
                     c.f_dc, c.scale_modifier, L, flag_offset, presum ? B.l1a : ~(size_t)0, b);
  return check_launch();
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_raster_preprocess_forward(int N, int sh_degree, int M, int H, int W, const float *means3D,
                                              const float *shs, const float *colors_precomp, const float *opacities,
                                              const float *scales, const float *rotations,
                                              const float *cov3D_precomp, float scale_modifier,
                                              const float *viewmatrix, const float *projmatrix, const float *campos,
                                              float tanfovx, float tanfovy, int32_t *radii, void *geom,
                                              size_t geom_bytes, int64_t *R_host, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0 || H <= 0 || W <= 0 || !geom || !viewmatrix || !projmatrix || !campos) return DIMO_E_ARG;
  if (N > 0 && (shs == nullptr) == (colors_precomp == nullptr)) return DIMO_E_ARG;
  if (N > 0 && !cov3D_precomp && (!scales || !rotations)) return DIMO_E_ARG;
  if (shs && (sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1))) return DIMO_E_ARG;
  if ((W + TILE - 1) / TILE > 65535 || (H + TILE - 1) / TILE > 65535) return DIMO_E_ARG;
  GeomLayout L(N);
  if (geom_bytes < L.bytes) return DIMO_E_WORKSPACE;
  if (N > 0 && (!means3D || !opacities || !radii)) return DIMO_E_ARG;
  BinGrid gi;
  if (!make_bin_grid(H, W, gi)) return DIMO_E_ARG;  // more than MAX_SUPER * 64 tiles
  const int nb = N > 0 ? (N + PRE_BLOCK - 1) / PRE_BLOCK : 1;  // (N = 0: one block clears the binning's counters)
  {
    ScopedTimer tm(T_PREPROCESS_FWD, stream);
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(nb), dim3(PRE_BLOCK), 0, stream, N, sh_degree, M, H, W, means3D,
                       shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, scale_modifier, viewmatrix,
                       projmatrix, campos, tanfovx, tanfovy, radii, at<Splat>(geom, L.splat),
                       at<uint16_t>(geom, L.rect), at<uint32_t>(geom, L.tiles), at<uint8_t>(geom, L.flags),
                       at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.key32), at<uint32_t>(geom, L.bk),
                       gi.ss_shift);
  }
  if (R_host) {
    hipLaunchKernelGGL(total_instances_kernel, dim3(1), dim3(256), 0, stream, nb, at<uint32_t>(geom, L.block_sums),
                       at<uint32_t>(geom, L.total));
    uint32_t tot[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(tot, at<uint32_t>(geom, L.total), sizeof(tot), hipMemcpyDeviceToHost, stream) != hipSuccess)
      return DIMO_E_LAUNCH;
    if (hipStreamSynchronize(stream) != hipSuccess) return DIMO_E_LAUNCH;
    *R_host = (int64_t)tot[0];
  }
  return check_launch();
}

int dimo::preprocess_backward_launch(
    int N, int sh_degree, int M, int H, int W, int64_t R_cap, const float *means3D, const float *shs,
    const float *colors_precomp, const float *scales, const float *rotations, const float *cov3D_precomp,
    float scale_modifier, const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
    float tanfovy, const int32_t *radii, const void *geom, const void *inst_grad, float *dL_dmeans3D,
    float *dL_dmeans2D, float *dL_dshs, float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drot,
    float *dL_dcov3D, hipStream_t stream) {
  GeomLayout L(N);
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb == 0) return DIMO_OK;
  ScopedTimer tm(T_PREPROCESS_BWD, stream);
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(nb), dim3(PRE_BLOCK), 0, stream, N, sh_degree, M, H, W,
                     (uint32_t)(R_cap > 0xffffffffLL ? 0xffffffffu : (uint32_t)R_cap), means3D, shs, colors_precomp,
                     scales, rotations, cov3D_precomp, scale_modifier, viewmatrix, projmatrix, campos, tanfovx,
                     tanfovy, radii, at<Splat>(geom, L.splat), at<uint32_t>(geom, L.offsets),
                     at<uint8_t>(geom, L.flags), reinterpret_cast<const SplatGrad *>(inst_grad),
                     reinterpret_cast<const uint8_t *>(inst_grad) +
                         align_up((size_t)(R_cap > 0 ? R_cap : 1) * sizeof(SplatGrad)),
                     at<unsigned long long>(geom, L.hitmask), dL_dmeans3D,
                     dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacity, dL_dscales, dL_drot, dL_dcov3D);
  return check_launch();
}
