// Per-tile alpha compositing, forward and backward.
//
// One workgroup (256 threads = 4 wave64) per 16x16 tile.  Each WAVE owns an 8x8 pixel quadrant
// (not a 16x4 strip): splats are small and round, so an 8x8 footprint keeps more of a wave's
// lanes doing the same thing.  The tile's depth-sorted instance list is staged through LDS in
// batches of 256 records (all lanes then read the same record: conflict-free LDS broadcast).
//
// Quadrant culling: while staging, the loading thread computes the bounding box of the region where the
// splat's alpha can reach 1/255 (Mahalanobis radius^2 <= 2 ln(255 opacity)) and a 4-bit mask of the
// quadrants it overlaps; every wave then compacts the batch into its own index list (ballot + popcount)
// and only visits the records that can touch its 64 pixels.  The box is conservative, so results are
// unchanged; at ~1e5 small splats per frame it removes most of the per-record rejection tests.
//
// Backward is free of global atomics: per (tile, instance) sums are reduced inside the wave
// (halving DPP butterfly -> LDS) and written as ONE 64-byte record per instance at the
// instance's emission position, so that the per-Gaussian kernel can sum a contiguous run.
// In the backward a wave owns a whole tile (four pixels per lane, one per quadrant): see blend_bwd_body.
//
// Backward work items are BUCKETS of 64 consecutive list entries, not tiles.  A tile's list is a sequential
// recurrence (transmittance), a trained scene saturates after ~160 of ~1200 entries on average but after 550 on
// the worst tile, and 800 tile-sized workgroups on 256 CUs leave the chip waiting for that one chain.  The forward
// pass therefore checkpoints every pixel's compositing state (T, the 7 accumulated features, the accumulated
// weight) at each bucket boundary it crosses and queues one item per bucket some pixel of the tile reaches; the
// backward runs each bucket front-to-back from its checkpoint:
//     dL/dalpha_i = D_i T_i - (S - P_i) / (1 - alpha_i),   P_i = sum_{j<=i} D_j w_j,  S = P_last + T_final (bg . dL/dC)
// with P at the bucket start = dL/dout . (checkpointed accumulators) and S from the final accumulators.  T_i is
// the forward's own product (no division chain), and ~2700 equal-sized items replace ~800 unequal ones.
#include "common.hpp"
#include "wave_ops.hpp"
#include <cstdlib>

namespace dimo {

constexpr int BLEND_BLOCK = 256;
constexpr int BATCH = 256;
constexpr int BWD_GRID = 16384;  // persistent single-wave workgroups looping over the (tile, bucket) items

__device__ __forceinline__ void pixel_of_thread(int tile_x, int tile_y, int &px, int &py) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  px = tile_x * TILE + (wave & 1) * 8 + (lane & 7);
  py = tile_y * TILE + (wave >> 1) * 8 + (lane >> 3);
}

// 4-bit mask of the tile's 8x8 quadrants (bit = wave index) that the alpha >= 1/255 region of a splat can reach:
//     alpha >= 1/255  <=>  q(d) = A dx^2 + 2 B dx dy + C dy^2 <= tau = 2 ln(255 opacity),
// so a quadrant is reachable iff the minimum of the convex form q over its pixel-centre rectangle is <= tau.  That
// minimum is 0 if the centre lies inside, else it sits on one of the four edges (a clamped 1-D minimiser each).
// Conservative: tau carries slack for the exp / log rounding of the kernels' own test.  (A first version tested the
// ellipse's bounding box: a quarter of the quadrant visits it let through had no active pixel -- the box corners.)
__device__ __forceinline__ float edge_min(float P, float Q, float R, float fixed, float lo, float hi) {
  // min over t in [lo, hi] of P fixed^2 + 2 Q fixed t + R t^2   (R > 0)
  const float t = fminf(fmaxf(-Q * fixed / R, lo), hi);
  return P * fixed * fixed + 2.0f * Q * fixed * t + R * t * t;
}
__device__ __forceinline__ bool rect_reach(float A, float B, float C, float tau, float u0, float u1, float v0,
                                           float v1) {
  if (u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f) return true;
  const float m = fminf(fminf(edge_min(A, B, C, u0, v0, v1), edge_min(A, B, C, u1, v0, v1)),
                        fminf(edge_min(C, B, A, v0, u0, u1), edge_min(C, B, A, v1, u0, u1)));
  return m <= tau;
}
__device__ __forceinline__ uint32_t quadrant_mask(float gx, float gy, float A, float B, float C, float opacity,
                                                  int tile_x, int tile_y) {
  if (!(opacity * 255.0f >= 1.0f)) return 0u;  // alpha = min(.99, o * G) with G <= 1 can never reach 1/255
  const float det = A * C - B * B;
  if (!(det > 0.0f) || !(A > 0.0f) || !(C > 0.0f)) return 0xfu;  // degenerate conic: no culling
  const float tau = 2.0f * __logf(opacity * 255.0f) * 1.02f + 0.05f;  // bound on q with slack for exp/log rounding
  // pixel centres of the quadrants relative to the splat centre, half a pixel of margin on each side
  const float x0 = (float)(tile_x * TILE) - gx, y0 = (float)(tile_y * TILE) - gy;
  const float ul0 = x0 - 0.5f, ul1 = x0 + 7.5f, ur0 = x0 + 7.5f, ur1 = x0 + 15.5f;
  const float vt0 = y0 - 0.5f, vt1 = y0 + 7.5f, vb0 = y0 + 7.5f, vb1 = y0 + 15.5f;
  return (uint32_t)rect_reach(A, B, C, tau, ul0, ul1, vt0, vt1) |
         ((uint32_t)rect_reach(A, B, C, tau, ur0, ur1, vt0, vt1) << 1) |
         ((uint32_t)rect_reach(A, B, C, tau, ul0, ul1, vb0, vb1) << 2) |
         ((uint32_t)rect_reach(A, B, C, tau, ur0, ur1, vb0, vb1) << 3);
}

// Every wave builds the ascending list of batch entries whose mask has its bit set.  Returns the count.
__device__ __forceinline__ int compact_for_wave(const uint32_t *s_mask, uint16_t *my_list, int count, int wave,
                                                int lane) {
  int cnt = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int c = 0; c < count; c += 64) {
    const int j = c + lane;
    const bool hit = j < count && ((s_mask[j] >> wave) & 1u);
    const unsigned long long bal = __ballot(hit);
    if (hit) my_list[cnt + __popcll(bal & lt)] = (uint16_t)j;
    cnt += __popcll(bal);
  }
  return cnt;
}

// ---------------------------------------------------------------------------------- forward
template <bool NORMAL>
__device__ __forceinline__ void blend_fwd_body(
    int H, int W, int tiles_x, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ vals_sorted,
    const Splat *__restrict__ splat, const float *__restrict__ bg, float *__restrict__ out_color,
    float *__restrict__ out_depth, float *__restrict__ out_normal, float *__restrict__ out_alpha,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ final_acc,
    float *__restrict__ ckpt, uint32_t *__restrict__ work) {
  __shared__ uint32_t s_last[BLEND_BLOCK / 64];
  __shared__ float4 s_geo[BATCH];   // x y A B
  __shared__ float4 s_col[BATCH];   // C opacity r g
  __shared__ float4 s_aux[BATCH];   // b depth nx ny
  __shared__ float s_nz[BATCH];
  __shared__ uint32_t s_mask[BATCH];
  __shared__ uint16_t s_list[BLEND_BLOCK / 64][BATCH];

  const int tile = blockIdx.x;
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int px, py;
  pixel_of_thread(tile_x, tile_y, px, py);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1];

  float T = 1.0f, wsum = 0.0f;
  float acc[NFEAT] = {0, 0, 0, 0, 0, 0, 0};
  uint32_t last = 0;
  bool done = !inside;
  // checkpoint slots of this tile: (lo / BUCKET + tile) + bucket -- strictly increasing with the tile index and
  // never overlapping (ceil(len / BUCKET) <= floor((lo + len) / BUCKET) - floor(lo / BUCKET) + 1)
  float *const ck_base = ckpt + ((size_t)(lo / BUCKET) + tile) * (CKPT_FLOATS * TILE * TILE) + threadIdx.x;
  uint32_t next_ck = 0;  // first bucket whose start state this wave has not stored yet (wave-uniform)

  for (uint32_t start = lo; start < hi; start += BATCH) {
    if (__syncthreads_count(done) == BLEND_BLOCK) break;
    const uint32_t idx = start + threadIdx.x;
    if (idx < hi) {
      const float4 *rp = reinterpret_cast<const float4 *>(splat + vals_sorted[idx]);
      const float4 a = rp[0], b = rp[1], c = rp[2];
      s_geo[threadIdx.x] = a;
      s_col[threadIdx.x] = b;
      s_aux[threadIdx.x] = c;
      if (NORMAL) s_nz[threadIdx.x] = rp[3].x;
      s_mask[threadIdx.x] = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, tile_x, tile_y);
    }
    __syncthreads();
    const int count = (int)min((uint32_t)BATCH, hi - start);
    const int mine = compact_for_wave(s_mask, s_list[wave], count, wave, lane);
    // Software pipeline over the wave's list: the index of visit t+2 and the record of visit t+1 are in flight
    // from LDS while visit t computes (index -> readfirstlane -> record is a dependent two-hop chain otherwise
    // paid in full on every visit of this sequential loop).
    int raw_next = (int)s_list[wave][mine > 1 ? 1 : 0];
    int jn = mine > 0 ? __builtin_amdgcn_readfirstlane((int)s_list[wave][0]) : 0;
    float4 gn = s_geo[jn], cn = s_col[jn], an = s_aux[jn];
    float nzn = NORMAL ? s_nz[jn] : 0.0f;
    for (int t = 0; t < mine; ++t) {
      if (__ballot(!done) == 0) break;  // whole wave finished
      const int j = jn;
      const float4 g = gn, c = cn, a = an;
      const float nz = nzn;
      jn = __builtin_amdgcn_readfirstlane(raw_next);
      raw_next = (int)s_list[wave][min(t + 2, mine - 1)];
      gn = s_geo[jn], cn = s_col[jn], an = s_aux[jn];
      if (NORMAL) nzn = s_nz[jn];
      for (const uint32_t eb = ((start - lo) + (uint32_t)j) / BUCKET; next_ck <= eb; ++next_ck) {
        float *ck = ck_base + (size_t)next_ck * (CKPT_FLOATS * TILE * TILE);
        ck[0] = T;
#pragma unroll
        for (int k = 0; k < NFEAT; ++k) ck[(1 + k) * TILE * TILE] = acc[k];
        ck[8 * TILE * TILE] = wsum;
      }
      // Predicated visit: the rejection tests of the published loop are one predicate, one wave-level skip and
      // selects.  Per-lane `continue`s cost as many scalar instructions (exec save / restore, branches) as there
      // were vector ones; in the batched launches, which are issue bound, this form is 6 % faster.
      const float dx = g.x - pxf, dy = g.y - pyf;
      const float power = -0.5f * (g.z * dx * dx + c.x * dy * dy) - g.w * dx * dy;
      const float alpha = fminf(ALPHA_MAX, c.y * __expf(power));
      const float test_T = T * (1.0f - alpha);
      const bool hit = !done && power <= 0.0f && alpha >= ALPHA_MIN;
      if (__ballot(hit) == 0) continue;
      const bool add = hit && !(test_T < T_STOP);
      done = done || (hit && !add);
      const float w = add ? alpha * T : 0.0f;
      acc[0] += c.z * w, acc[1] += c.w * w, acc[2] += a.x * w, acc[3] += a.y * w;
      if (NORMAL) acc[4] += a.z * w, acc[5] += a.w * w, acc[6] += nz * w;
      wsum += w;
      T = add ? test_T : T;
      last = add ? (start - lo) + (uint32_t)j + 1u : last;
    }
  }
  if (inside) {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    final_T[pix] = T;
    n_contrib[pix] = last;
#pragma unroll
    for (int k = 0; k < NFEAT; ++k) final_acc[k * HW + pix] = acc[k];
    final_acc[7 * HW + pix] = wsum;
    out_color[pix] = acc[0] + T * bg[0];
    out_color[HW + pix] = acc[1] + T * bg[1];
    out_color[2 * HW + pix] = acc[2] + T * bg[2];
    out_depth[pix] = acc[3];
    if (NORMAL) {
      out_normal[pix] = acc[4];
      out_normal[HW + pix] = acc[5];
      out_normal[2 * HW + pix] = acc[6];
    }
    out_alpha[pix] = wsum;
  }
  // queue one backward item per bucket some pixel of this tile reaches
  uint32_t m = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if (lane == 0) s_last[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t deepest = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const uint32_t nb = (deepest + BUCKET - 1) / BUCKET;
    if (nb) {
      const uint32_t base = atomicAdd(work, nb);
      // (the item carries its tile's list range: one dependent load less in front of every backward item)
      for (uint32_t b = 0; b < nb; ++b)
        reinterpret_cast<uint4 *>(work)[1 + base + b] = make_uint4(((uint32_t)tile << 12) | b, lo, hi, 0u);
    }
  }
}

// ---------------------------------------------------------------------------------- backward
// (wave reduction helpers: wave_ops.hpp)
//
// ONE WAVE per (tile, bucket) item, four pixels per lane -- lane l owns pixel l of EACH 8x8 quadrant.  The 13
// per-(record) sums are then accumulated over a lane's four pixels in registers and cross the wave ONCE per record
// (one halving butterfly + one plain 16-lane LDS store: the wave is the only writer of its item), while quadrant
// culling stays wave-uniform: a quadrant the record cannot reach, or whose pixels all stopped earlier, is skipped by
// a scalar branch.  The previous layout (one workgroup per item, one wave per quadrant) paid the ~60-instruction
// reduction once per visited QUADRANT -- 2.3 times per record on the C3 workload, more than the ~45 instructions of
// the per-pixel evaluation itself.
//
// QPW = quadrants per wave: 4 -> one wave per item (the batched launches of the step executor, >= 10^4 items in
// flight); 1 or 2 -> 4 or 2 waves per item, each with its own visit list and LDS float atomics into the shared
// sums (a single render has ~2700 items for 1024 SIMDs: one wave per SIMD is latency bound, 0.185 ms against
// 0.129 ms with four waves per item).
template <bool NORMAL, int QPW>
__device__ __forceinline__ void blend_bwd_body(
    int H, int W, int tiles_x, uint32_t R_cap, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ vals_sorted, const Splat *__restrict__ splat, const uint16_t *__restrict__ rect,
    const uint32_t *__restrict__ offsets, const float *__restrict__ bg, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ final_acc, const float *__restrict__ ckpt,
    const uint32_t *__restrict__ work, const float *__restrict__ dL_dcolor, const float *__restrict__ dL_ddepth,
    const float *__restrict__ dL_dnormal, const float *__restrict__ dL_dalpha, SplatGrad *__restrict__ inst_grad,
    uint8_t *__restrict__ inst_flag) {
  __shared__ float4 s_geo[BUCKET];
  __shared__ float4 s_col[BUCKET];
  __shared__ float4 s_aux[BUCKET];
  __shared__ float s_nz[BUCKET];
  __shared__ uint32_t s_emit[BUCKET];
  __shared__ uint32_t s_mask[BUCKET];
  constexpr int WAVES = 4 / QPW;
  __shared__ uint16_t s_list[WAVES][BUCKET];
  __shared__ float s_acc[BUCKET][16];
  static_assert(BUCKET == 64, "one staged record per lane of the first wave");

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t HW = (size_t)H * W;
  const uint32_t n_items = work[0];
  for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint4 it = reinterpret_cast<const uint4 *>(work)[1 + item];
    const uint32_t code = it.x, lo = it.y, hi = it.z;
    const int tile = (int)(code >> 12);
    const uint32_t blo = (code & 0xfffu) * BUCKET;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int count = (int)min((uint32_t)BUCKET, hi - lo - blo);
    const int bx = tile_x * TILE + (lane & 7), by = tile_y * TILE + (lane >> 3);
    const float bxf = (float)bx, byf = (float)by;
    const float *const ck_item = ckpt + ((size_t)(lo / BUCKET) + tile + blo / BUCKET) * (CKPT_FLOATS * TILE * TILE);

    __syncthreads();  // previous item fully consumed
    // QPW == 4 (one wave per item): the records are staged COMPACTED -- record number rank-among-the-reachable goes
    // to slot rank, with its original list position in the high bits of the mask word -- so the visit loop reads slot
    // t directly instead of chasing an index list (two dependent LDS round trips per record, ~250 cycles of a
    // latency-bound loop); every lane keeps its own rank / emission slot in registers for the epilogue.
    uint32_t my_emit = 0;
    int my_rank = 0, mine = 0;
    bool my_hit = false;
    if (QPW == 4) {
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc4 = ra;
      float rnz = 0.0f;
      uint32_t qmask = 0;
      if (lane < count) {
        const uint32_t g = vals_sorted[lo + blo + lane];
        const float4 *rp = reinterpret_cast<const float4 *>(splat + g);
        ra = rp[0], rb = rp[1], rc4 = rp[2];
        if (NORMAL) rnz = rp[3].x;
        qmask = quadrant_mask(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, tile_x, tile_y);
        const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)g);
        const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff;
        my_emit = (g == 0 ? 0u : offsets[g - 1]) + (uint32_t)((tile_y - y0) * (x1 - x0) + (tile_x - x0));
      }
      my_hit = qmask != 0u;
      const unsigned long long bal = __ballot(my_hit);
      my_rank = __popcll(bal & ((1ull << lane) - 1ull));
      mine = __popcll(bal);
      if (my_hit) {
        s_geo[my_rank] = ra, s_col[my_rank] = rb, s_aux[my_rank] = rc4;
        if (NORMAL) s_nz[my_rank] = rnz;
        s_mask[my_rank] = qmask | ((uint32_t)lane << 8);
      }
    } else if (wave == 0 && lane < count) {
      const uint32_t g = vals_sorted[lo + blo + lane];
      const float4 *rp = reinterpret_cast<const float4 *>(splat + g);
      const float4 a = rp[0], b = rp[1];
      s_geo[lane] = a;
      s_col[lane] = b;
      s_aux[lane] = rp[2];
      if (NORMAL) s_nz[lane] = rp[3].x;
      s_mask[lane] = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, tile_x, tile_y);
      const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)g);
      const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff;
      s_emit[lane] = (g == 0 ? 0u : offsets[g - 1]) + (uint32_t)((tile_y - y0) * (x1 - x0) + (tile_x - x0));
    }
#pragma unroll
    for (int k = 0; k < QPW; ++k)
      reinterpret_cast<float4 *>(&s_acc[0][0])[k * (64 * WAVES) + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the records are staged BEFORE the pixel state is loaded: the quadrant tests need registers, and with the four
    // quadrants' state already live they spilled)
    __builtin_amdgcn_sched_barrier(0);
    // per-quadrant pixel state.  Every load is unconditional (clamped pixel index, substitute pointer for an absent
    // gradient image) and the predicates are applied afterwards with selects: a branch around a load makes the
    // compiler drain the whole memory queue at the join, which put ~16 dependent round trips in front of every item.
    uint32_t last[QPW], deepest[QPW];
    float T[QPW], SP[QPW], dp[QPW][8];  // SP = S - P: what the entries behind the current one still add
    const float *const pc = dL_dcolor ? dL_dcolor : final_T, *const pd = dL_ddepth ? dL_ddepth : final_T;
    const float *const pn = (NORMAL && dL_dnormal) ? dL_dnormal : final_T, *const pa = dL_dalpha ? dL_dalpha : final_T;
    const float kc = dL_dcolor ? 1.0f : 0.0f, kd = dL_ddepth ? 1.0f : 0.0f;
    const float kn = (NORMAL && dL_dnormal) ? 1.0f : 0.0f, ka = dL_dalpha ? 1.0f : 0.0f;
    // 32-bit element offsets from wave-uniform base pointers (SGPR base + VGPR offset addressing: a 64-bit address
    // pair per load is what overflowed the register file here)
    const uint32_t HW32 = (uint32_t)HW;
    const uint32_t HWc = dL_dcolor ? HW32 : 0u, HWn = (NORMAL && dL_dnormal) ? HW32 : 0u;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      const int quad = wave * QPW + q;
      const int px = bx + (quad & 1) * 8, py = by + (quad >> 1) * 8;
      const bool inside = px < W && py < H;
      const uint32_t pix = inside ? (uint32_t)py * (uint32_t)W + (uint32_t)px : 0u;
      const uint32_t nc = n_contrib[pix];
      const float *c = ck_item + (uint32_t)(quad * 64 + lane);  // the forward's thread index = quadrant * 64 + lane
      float cv[9], fa[8];
#pragma unroll
      for (int k = 0; k < 9; ++k) cv[k] = c[k * TILE * TILE];
#pragma unroll
      for (int k = 0; k < 8; ++k) fa[k] = final_acc[(uint32_t)k * HW32 + pix];
      const float fT = final_T[pix];
      dp[q][0] = kc * pc[pix], dp[q][1] = kc * pc[HWc + pix], dp[q][2] = kc * pc[2u * HWc + pix];
      dp[q][3] = kd * pd[pix];
      dp[q][4] = kn * pn[pix], dp[q][5] = kn * pn[HWn + pix], dp[q][6] = kn * pn[2u * HWn + pix];
      dp[q][7] = ka * pa[pix];
      last[q] = inside ? nc : 0u;
      const bool need = last[q] > blo;
#pragma unroll
      for (int k = 0; k < 8; ++k) dp[q][k] = need ? dp[q][k] : 0.0f;
      float P = dp[q][7] * cv[8];
      float S = dp[q][7] * fa[7] + fT * (bg0 * dp[q][0] + bg1 * dp[q][1] + bg2 * dp[q][2]);
#pragma unroll
      for (int k = 0; k < (NORMAL ? 7 : 4); ++k) {
        P += dp[q][k] * cv[1 + k];
        S += dp[q][k] * fa[k];
      }
      T[q] = need ? cv[0] : 0.0f;
      SP[q] = need ? S - P : 0.0f;
      // deepest entry any pixel of this quadrant still looks at (wave-uniform)
      uint32_t m = last[q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
      deepest[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
      // one quadrant's ~27 loads in flight at a time: hoisting all four above the first use spills 48 registers
      if (QPW == 4) __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t wlast = deepest[0];
#pragma unroll
    for (int q = 1; q < QPW; ++q) wlast = max(wlast, deepest[q]);

    __syncthreads();
    // ascending list of the records that can reach any pixel of this wave's quadrants
    constexpr uint32_t QBITS = ((1u << QPW) - 1u);
    if (QPW != 4) {
      const bool hit = lane < count && ((s_mask[lane] >> (wave * QPW)) & QBITS) != 0u;
      const unsigned long long bal = __ballot(hit);
      if (hit) s_list[wave][__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)lane;
      mine = __popcll(bal);
    }
    __syncthreads();

    // (QPW == 4) the mask word and the geometry of record t + 1 are read from LDS while record t is processed
    uint32_t m_next = 0;
    float4 g_next = make_float4(0.f, 0.f, 0.f, 0.f), c_next = g_next;
    if (QPW == 4 && mine > 0) m_next = s_mask[0], g_next = s_geo[0], c_next = s_col[0];
    for (int t = 0; t < mine; ++t) {
      // slot of the record in the staged arrays, its position in the bucket, its quadrant bits
      int slot, j;
      uint32_t qm;
      float4 g, c;
      if (QPW == 4) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_next);
        slot = t, j = (int)(m >> 8), qm = m & 0xfu;
        g = g_next, c = c_next;
        const int tn = min(t + 1, mine - 1);
        m_next = s_mask[tn], g_next = s_geo[tn], c_next = s_col[tn];
      } else {
        slot = j = __builtin_amdgcn_readfirstlane((int)s_list[wave][t]);
        qm = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_mask[j]) >> (wave * QPW);
        g = s_geo[slot], c = s_col[slot];
      }
      const uint32_t pos = blo + (uint32_t)j;
      if (pos >= wlast) break;  // the list is ascending: nothing further reaches this tile
      const float4 a = s_aux[slot];
      const float nz = NORMAL ? s_nz[slot] : 0.0f;
      const float dx0 = g.x - bxf - (float)(((wave * QPW) & 1) * 8), dy0 = g.y - byf - (float)(((wave * QPW) >> 1) * 8);
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = 0.0f;
      bool any = false;
#pragma unroll
      for (int q = 0; q < QPW; ++q) {
        if (!((qm >> q) & 1u) || pos >= deepest[q]) continue;  // wave-uniform
        // (QPW = 2: wave 0 owns the top quadrants 0, 1 and wave 1 the bottom ones, so q only moves along x)
        const float dx = dx0 - (float)((q & 1) * 8), dy = dy0 - (float)((QPW == 4 ? (q >> 1) : 0) * 8);
        const float power = -0.5f * (g.z * dx * dx + c.x * dy * dy) - g.w * dx * dy;
        const float G = __expf(power);
        const float alpha = fminf(ALPHA_MAX, c.y * G);
        const bool active = pos < last[q] && power <= 0.0f && alpha >= ALPHA_MIN;
        if (__ballot(active) == 0) continue;  // wave-uniform skip
        any = true;
        if (active) {
          const float w = alpha * T[q];
          float D = dp[q][7] + c.z * dp[q][0] + c.w * dp[q][1] + a.x * dp[q][2] + a.y * dp[q][3];
          if (NORMAL) D += a.z * dp[q][4] + a.w * dp[q][5] + nz * dp[q][6];
          SP[q] -= D * w;
          const float dL_dalpha_i = D * T[q] - SP[q] * __builtin_amdgcn_rcpf(1.0f - alpha);
          T[q] *= 1.0f - alpha;
          const float gg = G * c.y * dL_dalpha_i;  // g = G * dL/dG, dL/dG = opacity * dL/dalpha
          const float gx = gg * dx, gy = gg * dy;
          v[0] += gg, v[1] += gx, v[2] += gy;
          v[3] += gx * dx, v[4] += gx * dy, v[5] += gy * dy;
          v[6] += w * dp[q][0], v[7] += w * dp[q][1], v[8] += w * dp[q][2], v[9] += w * dp[q][3];
          if (NORMAL) v[10] += w * dp[q][4], v[11] += w * dp[q][5], v[12] += w * dp[q][6];
        }
      }
      if (!any) continue;
      const float tot = butterfly16(v, lane);  // lanes 0..15: the wave total of value butterfly16_slot(lane)
      if (lane < 16) {
        if (QPW == 4) s_acc[slot][butterfly16_slot(lane)] = tot;  // this wave is the only writer of the record
        else atomicAdd(&s_acc[j][butterfly16_slot(lane)], tot);
      }
    }
    __syncthreads();
    if (QPW == 4 ? my_hit : (wave == 0 && lane < count)) {
      const uint32_t e = QPW == 4 ? my_emit : s_emit[lane];
      const float4 *src = reinterpret_cast<const float4 *>(&s_acc[QPW == 4 ? my_rank : lane][0]);
      const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
      const bool nonzero = r0.x != 0.f || r0.y != 0.f || r0.z != 0.f || r0.w != 0.f || r1.x != 0.f || r1.y != 0.f ||
                           r1.z != 0.f || r1.w != 0.f || r2.x != 0.f || r2.y != 0.f || r2.z != 0.f || r2.w != 0.f ||
                           r3.x != 0.f;
      if (e < R_cap && nonzero) {
        float4 *dst = reinterpret_cast<float4 *>(inst_grad + e);
        dst[0] = r0, dst[1] = r1, dst[2] = r2, dst[3] = r3;
        inst_flag[e] = 1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- kernel entry points
template <bool NORMAL>
__global__ void __launch_bounds__(BLEND_BLOCK) blend_fwd_kernel(
    int H, int W, int tiles_x, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ vals_sorted,
    const Splat *__restrict__ splat, const float *__restrict__ bg, float *__restrict__ out_color,
    float *__restrict__ out_depth, float *__restrict__ out_normal, float *__restrict__ out_alpha,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ final_acc,
    float *__restrict__ ckpt, uint32_t *__restrict__ work) {
  blend_fwd_body<NORMAL>(H, W, tiles_x, ranges, vals_sorted, splat, bg, out_color, out_depth, out_normal, out_alpha,
                         final_T, n_contrib, final_acc, ckpt, work);
}
template <bool NORMAL, int QPW>
__global__ void __launch_bounds__(256 / QPW, QPW == 4 ? 3 : 4) blend_bwd_kernel(
    int H, int W, int tiles_x, uint32_t R_cap, const uint32_t *__restrict__ ranges,
    const uint32_t *__restrict__ vals_sorted, const Splat *__restrict__ splat, const uint16_t *__restrict__ rect,
    const uint32_t *__restrict__ offsets, const float *__restrict__ bg, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ final_acc, const float *__restrict__ ckpt,
    const uint32_t *__restrict__ work, const float *__restrict__ dL_dcolor, const float *__restrict__ dL_ddepth,
    const float *__restrict__ dL_dnormal, const float *__restrict__ dL_dalpha, SplatGrad *__restrict__ inst_grad,
    uint8_t *__restrict__ inst_flag) {
  blend_bwd_body<NORMAL, QPW>(H, W, tiles_x, R_cap, ranges, vals_sorted, splat, rect, offsets, bg, final_T, n_contrib,
                         final_acc, ckpt, work, dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha, inst_grad, inst_flag);
}

// Batched entry points (native step executor): blockIdx.y = render of the batch.
struct BlendOffsets {
  size_t splat, rect, offsets, total;           // geom
  size_t ranges, vals, ckpt, work;              // bin
  size_t final_T, n_contrib, final_acc;         // img
  size_t flag;                                  // backward scratch: records at 0, flags here
};
template <bool NORMAL>
__global__ void __launch_bounds__(BLEND_BLOCK) blend_fwd_batched_kernel(int H, int W, int tiles_x,
                                                                        const float *__restrict__ bg, BlendOffsets o,
                                                                        RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  blend_fwd_body<NORMAL>(H, W, tiles_x, at<uint32_t>(r.bin, o.ranges), at<uint32_t>(r.bin, o.vals),
                         at<Splat>(r.geom, o.splat), bg, r.out_color, r.out_depth, NORMAL ? r.out_normal : nullptr,
                         r.out_alpha, at<float>(r.img, o.final_T), at<uint32_t>(r.img, o.n_contrib),
                         at<float>(r.img, o.final_acc), at<float>(r.bin, o.ckpt), at<uint32_t>(r.bin, o.work));
}
template <bool NORMAL, int QPW>
__global__ void __launch_bounds__(256 / QPW, QPW == 4 ? 3 : 4) blend_bwd_batched_kernel(int H, int W, int tiles_x, uint32_t R_cap,
                                                                        const float *__restrict__ bg, BlendOffsets o,
                                                                        RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  blend_bwd_body<NORMAL, QPW>(H, W, tiles_x, R_cap, at<uint32_t>(r.bin, o.ranges), at<uint32_t>(r.bin, o.vals),
                         at<Splat>(r.geom, o.splat), at<uint16_t>(r.geom, o.rect), at<uint32_t>(r.geom, o.offsets), bg,
                         at<float>(r.img, o.final_T), at<uint32_t>(r.img, o.n_contrib), at<float>(r.img, o.final_acc),
                         at<float>(r.bin, o.ckpt), at<uint32_t>(r.bin, o.work), r.g_color, r.g_depth,
                         NORMAL ? r.g_normal : nullptr, r.g_alpha, reinterpret_cast<SplatGrad *>(r.bwd_scratch),
                         at<uint8_t>(r.bwd_scratch, o.flag));
}
// clears the "record written" flags of the instances a render actually has ([0, R), 16 bytes per thread)
__global__ void __launch_bounds__(256) clear_flags_batched_kernel(uint32_t R_cap, BlendOffsets o, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  const uint32_t R = min(*at<uint32_t>(r.geom, o.total), R_cap);
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i < R) *reinterpret_cast<uint4 *>(at<uint8_t>(r.bwd_scratch, o.flag) + i) = make_uint4(0, 0, 0, 0);
}

// quadrants per wave of the backward for a launch over n renders (DIMO_BWD_QPW overrides, for experiments)
static int bwd_quadrants_per_wave(int n) {
  static const int forced = [] {
    const char *e = getenv("DIMO_BWD_QPW");
    const int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2 || v == 4) ? v : 0;
  }();
  if (forced) return forced;
  return n >= 2 ? 4 : 2;
}

static BlendOffsets blend_offsets(const GeomLayout &G, const BinLayout &B, const ImgLayout &I) {
  BlendOffsets o;
  o.splat = G.splat, o.rect = G.rect, o.offsets = G.offsets, o.total = G.total;
  o.ranges = B.ranges, o.vals = B.vals_b, o.ckpt = B.ckpt, o.work = B.work;
  o.final_T = I.final_T, o.n_contrib = I.n_contrib, o.final_acc = I.final_acc;
  o.flag = align_up(B.cap * sizeof(SplatGrad));
  return o;
}

int blend_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W);
  ImgLayout I(c.H, c.W);
  if (c.bin_bytes < B.bytes || c.img_bytes < I.bytes) return DIMO_E_WORKSPACE;
  const BlendOffsets o = blend_offsets(G, B, I);
  ScopedTimer tm(T_BLEND_FWD, stream);
  if (c.with_normal)
    hipLaunchKernelGGL(blend_fwd_batched_kernel<true>, dim3(B.T, n), dim3(BLEND_BLOCK), 0, stream, c.H, c.W, B.tiles_x,
                       c.bg, o, b);
  else
    hipLaunchKernelGGL(blend_fwd_batched_kernel<false>, dim3(B.T, n), dim3(BLEND_BLOCK), 0, stream, c.H, c.W,
                       B.tiles_x, c.bg, o, b);
  return check_launch();
}

int blend_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0 || c.N <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W);
  ImgLayout I(c.H, c.W);
  if (c.bwd_scratch_bytes < align_up(B.cap * sizeof(SplatGrad)) + align_up(B.cap)) return DIMO_E_WORKSPACE;
  const BlendOffsets o = blend_offsets(G, B, I);
  const uint32_t cap = (uint32_t)B.cap;
  ScopedTimer tm(T_BLEND_BWD, stream);
  hipLaunchKernelGGL(clear_flags_batched_kernel, dim3((unsigned)((B.cap / 16 + 255) / 256 + 1), n), dim3(256), 0,
                     stream, cap, o, b);
  // one wave per item when the batch supplies enough items to fill the chip, two waves per item for a lone render
  const int qpw = bwd_quadrants_per_wave(n);
  const int want = BWD_GRID * qpw / 4;
  const int grid = (want + n - 1) / n < 2048 ? 2048 : (want + n - 1) / n;
#define DIMO_LAUNCH_BWD_BATCHED(NORMAL, QPW)                                                                         \
  hipLaunchKernelGGL((blend_bwd_batched_kernel<NORMAL, QPW>), dim3(grid, n), dim3(256 / QPW), 0, stream, c.H, c.W,   \
                     B.tiles_x, cap, c.bg, o, b)
  if (c.with_normal) {
    if (qpw == 4) DIMO_LAUNCH_BWD_BATCHED(true, 4);
    else if (qpw == 2) DIMO_LAUNCH_BWD_BATCHED(true, 2);
    else DIMO_LAUNCH_BWD_BATCHED(true, 1);
  } else {
    if (qpw == 4) DIMO_LAUNCH_BWD_BATCHED(false, 4);
    else if (qpw == 2) DIMO_LAUNCH_BWD_BATCHED(false, 2);
    else DIMO_LAUNCH_BWD_BATCHED(false, 1);
  }
#undef DIMO_LAUNCH_BWD_BATCHED
  return check_launch();
}

}  // namespace dimo

using namespace dimo;

extern "C" int dimo_raster_render_forward(int N, int H, int W, int64_t R_cap, const float *bg, const void *geom,
                                          void *bin, size_t bin_bytes, void *img, size_t img_bytes, float *out_color,
                                          float *out_depth, float *out_normal, float *out_alpha, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (N < 0 || H <= 0 || W <= 0 || R_cap < 0 || R_cap > 0xfffffff0LL) return DIMO_E_ARG;
  if (!bg || !geom || !bin || !img || !out_color || !out_depth || !out_alpha) return DIMO_E_ARG;
  GeomLayout G(N);
  BinLayout B(R_cap, H, W);
  ImgLayout I(H, W);
  if (bin_bytes < B.bytes || img_bytes < I.bytes) return DIMO_E_WORKSPACE;
  int rc = bin_instances(N, H, W, R_cap, geom, bin, stream);
  if (rc) return rc;
  const uint32_t *ranges = at<uint32_t>(bin, B.ranges);
  const uint32_t *vals = at<uint32_t>(bin, B.vals_b);
  const Splat *splat = at<Splat>(geom, G.splat);
  ScopedTimer tm(T_BLEND_FWD, stream);
  if (out_normal)
    hipLaunchKernelGGL(blend_fwd_kernel<true>, dim3(B.T), dim3(BLEND_BLOCK), 0, stream, H, W, B.tiles_x, ranges, vals,
                       splat, bg, out_color, out_depth, out_normal, out_alpha, at<float>(img, I.final_T),
                       at<uint32_t>(img, I.n_contrib), at<float>(img, I.final_acc), at<float>(bin, B.ckpt),
                       at<uint32_t>(bin, B.work));
  else
    hipLaunchKernelGGL(blend_fwd_kernel<false>, dim3(B.T), dim3(BLEND_BLOCK), 0, stream, H, W, B.tiles_x, ranges,
                       vals, splat, bg, out_color, out_depth, out_normal, out_alpha, at<float>(img, I.final_T),
                       at<uint32_t>(img, I.n_contrib), at<float>(img, I.final_acc), at<float>(bin, B.ckpt),
                       at<uint32_t>(bin, B.work));
  return check_launch();
}

// scratch = [R_cap x 64-byte SplatGrad records][R_cap x 1-byte "record written" flags]
extern "C" size_t dimo_raster_backward_scratch_bytes(int N, int64_t R_cap) {
  (void)N;
  const size_t cap = (size_t)(R_cap > 0 ? R_cap : 1);
  return align_up(cap * sizeof(SplatGrad)) + align_up(cap);
}

extern "C" int dimo_raster_backward(int N, int sh_degree, int M, int H, int W, int64_t R_cap, const float *means3D,
                                    const float *shs, const float *colors_precomp, const float *opacities,
                                    const float *scales, const float *rotations, const float *cov3D_precomp,
                                    float scale_modifier, const float *viewmatrix, const float *projmatrix,
                                    const float *campos, const float *bg, float tanfovx, float tanfovy,
                                    const int32_t *radii, const void *geom, const void *bin, const void *img,
                                    const float *dL_dcolor, const float *dL_ddepth, const float *dL_dnormal,
                                    const float *dL_dalpha, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs,
                                    float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drotations,
                                    float *dL_dcov3D, void *scratch, size_t scratch_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  (void)opacities;
  if (N < 0 || H <= 0 || W <= 0 || R_cap < 0 || R_cap > 0xfffffff0LL) return DIMO_E_ARG;
  if (!geom || !bin || !img || !bg || !scratch || !viewmatrix || !projmatrix || !campos) return DIMO_E_ARG;
  if (N > 0 && (shs == nullptr) == (colors_precomp == nullptr)) return DIMO_E_ARG;
  if (N > 0 && (!means3D || !radii || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity)) return DIMO_E_ARG;
  if (N > 0 && shs && !dL_dshs) return DIMO_E_ARG;
  if (N > 0 && colors_precomp && !dL_dcolors) return DIMO_E_ARG;
  if (N > 0 && (cov3D_precomp ? !dL_dcov3D : (!dL_dscales || !dL_drotations || !scales || !rotations)))
    return DIMO_E_ARG;
  if (scratch_bytes < dimo_raster_backward_scratch_bytes(N, R_cap)) return DIMO_E_WORKSPACE;
  GeomLayout G(N);
  BinLayout B(R_cap, H, W);
  ImgLayout I(H, W);
  const uint32_t cap = (uint32_t)B.cap;
  SplatGrad *inst = reinterpret_cast<SplatGrad *>(scratch);
  uint8_t *inst_flag = reinterpret_cast<uint8_t *>(scratch) + align_up(B.cap * sizeof(SplatGrad));
  if (N > 0) {
    ScopedTimer tm(T_BLEND_BWD, stream);
    if (hipMemsetAsync(inst_flag, 0, B.cap, stream) != hipSuccess) return DIMO_E_LAUNCH;
    const int qpw = bwd_quadrants_per_wave(1);
#define DIMO_LAUNCH_BWD(NORMAL, QPW)                                                                                 \
  hipLaunchKernelGGL((blend_bwd_kernel<NORMAL, QPW>), dim3(BWD_GRID * QPW / 4), dim3(256 / QPW), 0, stream, H, W,    \
                     B.tiles_x, cap, at<uint32_t>(bin, B.ranges), at<uint32_t>(bin, B.vals_b),                       \
                     at<Splat>(geom, G.splat), at<uint16_t>(geom, G.rect), at<uint32_t>(geom, G.offsets), bg,        \
                     at<float>(img, I.final_T), at<uint32_t>(img, I.n_contrib), at<float>(img, I.final_acc),         \
                     at<float>(bin, B.ckpt), at<uint32_t>(bin, B.work), dL_dcolor, dL_ddepth, dL_dnormal, dL_dalpha, \
                     inst, inst_flag)
    if (dL_dnormal) {
      if (qpw == 4) DIMO_LAUNCH_BWD(true, 4);
      else if (qpw == 2) DIMO_LAUNCH_BWD(true, 2);
      else DIMO_LAUNCH_BWD(true, 1);
    } else {
      if (qpw == 4) DIMO_LAUNCH_BWD(false, 4);
      else if (qpw == 2) DIMO_LAUNCH_BWD(false, 2);
      else DIMO_LAUNCH_BWD(false, 1);
    }
#undef DIMO_LAUNCH_BWD
    int rc = check_launch();
    if (rc) return rc;
  }
  return preprocess_backward_launch(N, sh_degree, M, H, W, R_cap, means3D, shs, colors_precomp, scales, rotations,
                                    cov3D_precomp, scale_modifier, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                                    radii, geom, scratch, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacity,
                                    dL_dscales, dL_drotations, dL_dcov3D, stream);
}
