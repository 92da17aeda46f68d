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
constexpr int BWD_GRID = 16384;  // single-wave workgroups striding over the work items (most get one)
#ifndef DIMO_BWD_WAVES
#define DIMO_BWD_WAVES 4
#endif
constexpr int BWD_WAVES_PER_SIMD = DIMO_BWD_WAVES;  // register budget of the backward: 128 VGPRs at 4 (build.py passes
                                                    // -DDIMO_BWD_WAVES=<n> when the environment has DIMO_BWD_WAVES)

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
// The blend BACKWARD stages a record's PRE-MULTIPLIED conic (Splat::As, Bs, Cs: A' = -0.5 log2(e) A, B' = -log2(e) B,
// C' = -0.5 log2(e) C, written by the projection next to nz), so that a visit's exponent comes out in base 2,
// power' = dx (A' dx + B' dy) + C' dy dy = log2(e) power, and goes straight into v_exp_f32: one 4-cycle product
// fewer per quadrant visit (262 -> 258 us per 8-render launch).  The sign test `power > 0 -> skip` of the published loop
// is unchanged (log2(e) > 0); the exponent differs from the forward's by the rounding of the re-associated sum, ~1e-7
// relative.  The quadrant masks are computed from the plain conic.  (The same in the FORWARD -- ten instructions in
// front of its v_exp become seven -- measured SLOWER, 147-148 against 143 us per 8 renders and -1.1 % in the step, at
// five and at six waves per SIMD: its staging then reads the record's fourth float4 and re-assembles two of the staged
// vectors, 79 -> 93 VGPRs; the forward keeps the published form.)

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
    float *__restrict__ ckpt, uint32_t *__restrict__ work, uint32_t chain_arg, uint32_t *__restrict__ meta,
    int tile) {
  // Buckets per backward work item (chain).  Fixed by the caller (chain_arg > 0), or ADAPTIVE (chain_arg = 0, the step
  // executor's persistent render slots): read from the workspace, where the previous forward over it left what ITS lists
  // called for -- see the end of this function.  The backward needs no word of it: its items say where they start and
  // how many buckets they walk.
  uint32_t chain = chain_arg;
  if (chain == 0u) {
    const uint32_t cw = meta[META_CHAIN];
    chain = (cw == 2u || cw == 4u) ? cw : 1u;  // (a fresh workspace holds anything)
  }
  __shared__ uint32_t s_last[BLEND_BLOCK / 64];
  __shared__ float4 s_geo[BATCH];   // x y A B
  __shared__ float4 s_col[BATCH];   // C opacity r g
  __shared__ float4 s_aux[BATCH];   // b depth nx ny
  __shared__ float s_nz[BATCH];
  __shared__ uint32_t s_mask[BATCH];

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

  // One visit: the rejection tests of the published loop as ONE predicate and selects (per-lane `continue`s cost as
  // many scalar instructions -- exec save / restore, branches -- as there were vector ones).  `done` needs no test of
  // its own: a finished pixel has Tw = 0, so its products vanish and test_T < T_STOP keeps it from being added.
  float Tw = done ? 0.0f : 1.0f;  // working transmittance: T while the pixel is alive, 0 once it has stopped
#define DIMO_VISIT(G, C, A, NZ, POS)                                                              \
  {                                                                                               \
    const float dx = G.x - pxf, dy = G.y - pyf;                                                   \
    const float power = -0.5f * (G.z * dx * dx + C.x * dy * dy) - G.w * dx * dy;                  \
    const float alpha = fminf(ALPHA_MAX, C.y * __expf(power));                                    \
    const float test_T = Tw * (1.0f - alpha);                                                     \
    const bool hit = power <= 0.0f && alpha >= ALPHA_MIN;                                         \
    const bool add = hit && !(test_T < T_STOP);                                                   \
    const float w = add ? alpha * Tw : 0.0f;                                                      \
    acc[0] += C.z * w, acc[1] += C.w * w, acc[2] += A.x * w, acc[3] += A.y * w;                   \
    if (NORMAL) acc[4] += A.z * w, acc[5] += A.w * w, acc[6] += NZ * w;                           \
    wsum += w;                                                                                    \
    T = add ? test_T : T;                                                                         \
    last = add ? (POS) + 1u : last;                                                               \
    Tw = hit ? (add ? test_T : 0.0f) : Tw;                                                        \
  }
#define DIMO_LOAD(G, C, A, NZ, J) \
  G = s_geo[J], C = s_col[J], A = s_aux[J]; \
  if (NORMAL) NZ = s_nz[J];

  // (the list entry of the NEXT batch is requested while this one is worked on: a batch's staging was two dependent
  // memory round trips -- list entry, then record -- in front of every 256 records)
  uint32_t next_id = lo + threadIdx.x < hi ? vals_sorted[lo + threadIdx.x] : 0u;
  for (uint32_t start = lo; start < hi; start += BATCH) {
    if (__syncthreads_count(Tw == 0.0f) == BLEND_BLOCK) break;
    const uint32_t idx = start + threadIdx.x;
    const uint32_t id = next_id;
    next_id = idx + BATCH < hi ? vals_sorted[idx + BATCH] : 0u;
    if (idx < hi) {
      const float4 *rp = reinterpret_cast<const float4 *>(splat + id);
      const float4 a = rp[0], b = rp[1], c = rp[2];
      s_geo[threadIdx.x] = a;
      s_col[threadIdx.x] = b;
      s_aux[threadIdx.x] = c;
      if (NORMAL) s_nz[threadIdx.x] = rp[3].x;
      s_mask[threadIdx.x] = quadrant_mask(a.x, a.y, a.z, a.w, b.x, b.y, tile_x, tile_y);
    }
    __syncthreads();
    const int count = (int)min((uint32_t)BATCH, hi - start);
    // The batch is walked in four chunks of 64 records = the backward's buckets.  The records of a chunk that can
    // reach this wave's quadrant are the set bits of ONE ballot; the visit loop pulls them off with scalar bit
    // operations (round 1 compacted an index list through LDS and chased it with readfirstlane: two dependent LDS
    // round trips in front of every visit).  Two register sets alternate, the record of the next visit is in flight
    // from LDS while the current one computes.
#pragma unroll 1
    for (int ch = 0; ch < BATCH / 64; ++ch) {
      const int jb = ch * 64;
      if (jb >= count) break;
      if (__builtin_amdgcn_ballot_w64(Tw != 0.0f) == 0ull) break;  // every pixel of the quadrant has stopped
      // the state at the first entry of every bucket that STARTS A CHAIN of the backward (every `chain`-th bucket;
      // not bucket 0, whose state is T = 1 and empty sums)
      const uint32_t bucket = (start - lo) / BUCKET + (uint32_t)ch;
      if (bucket != 0u && bucket % chain == 0u) {
        float *ck = ck_base + (size_t)bucket * (CKPT_FLOATS * TILE * TILE);
        ck[0] = T;
#pragma unroll
        for (int k = 0; k < NFEAT; ++k) ck[(1 + k) * TILE * TILE] = acc[k];
        ck[8 * TILE * TILE] = wsum;
      }
      const int j_lane = jb + lane;
      unsigned long long bits = __builtin_amdgcn_ballot_w64(j_lane < count && ((s_mask[j_lane] >> wave) & 1u));
      if (bits == 0ull) continue;
      const uint32_t pos0 = (start - lo) + (uint32_t)jb;
      float4 g0, c0, a0, g1, c1, a1;
      float nz0 = 0.0f, nz1 = 0.0f;
      int j0 = __builtin_ctzll(bits), j1 = 0;
      bits &= bits - 1ull;
      DIMO_LOAD(g0, c0, a0, nz0, jb + j0)
      while (true) {
        const bool more1 = bits != 0ull;
        if (more1) {
          j1 = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          DIMO_LOAD(g1, c1, a1, nz1, jb + j1)
        }
        DIMO_VISIT(g0, c0, a0, nz0, pos0 + (uint32_t)j0)
        if (!more1) break;
        const bool more0 = bits != 0ull;
        if (more0) {
          j0 = __builtin_ctzll(bits);
          bits &= bits - 1ull;
          DIMO_LOAD(g0, c0, a0, nz0, jb + j0)
        }
        DIMO_VISIT(g1, c1, a1, nz1, pos0 + (uint32_t)j1)
        if (!more0) break;
        if (__builtin_amdgcn_ballot_w64(Tw != 0.0f) == 0ull) break;  // checked once per two visits
      }
    }
  }
#undef DIMO_VISIT
#undef DIMO_LOAD
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
  // queue one backward item per chain of `chain` buckets some pixel of this tile reaches
  uint32_t m = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if (lane == 0) s_last[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t deepest = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    const uint32_t nb = (deepest + BUCKET - 1) / BUCKET;
    if (nb) {
      // Three queues by the chain's position in its list -- head, second, deeper: [1, 1 + T), [1 + T, 1 + 2 T),
      // [1 + 2 T, ...) of the uint4 array, counts in work[0..2].  The head chains see every pixel of the tile and are
      // the expensive ones; the backward takes all heads first, the sparse deep chains fill the tail of the launch.
      // Item = (tile << 12 | first bucket, list start, list end, buckets in the chain): it carries its tile's list
      // range, one dependent load less in front of every backward item.
      const uint32_t ni = (nb + chain - 1) / chain;
      const uint32_t T = gridDim.x;
      uint4 *const q = reinterpret_cast<uint4 *>(work) + 1;
      q[atomicAdd(work, 1u)] = make_uint4(((uint32_t)tile << 12), lo, hi, min(chain, nb));
      if (ni > 1) q[T + atomicAdd(work + 1, 1u)] = make_uint4(((uint32_t)tile << 12) | chain, lo, hi, min(chain, nb - chain));
      if (ni > 2) {
        const uint32_t base = atomicAdd(work + 2, ni - 2);
        for (uint32_t i = 2; i < ni; ++i)
          q[2 * T + base + (i - 2)] = make_uint4(((uint32_t)tile << 12) | (i * chain), lo, hi, min(chain, nb - i * chain));
      }
    }
    // Adaptive chains.  One bucket per item is right where lists saturate early (a trained scene: ~2.3 buckets per tile
    // reached, ~2 400 items per render -- longer items lose more in the launch's tail than they save in state loads: 2 %
    // slower at two buckets per item); where every pixel walks its whole list (the reference's initial state, every
    // opacity 0.05: ~15.5 buckets per tile, 16 000 items per render) the items are plenty, and what an item of four
    // buckets saves -- three of four checkpoint stores here, three of four checkpoint + pixel-state loads there -- shows:
    // forward 659 -> 629 us, backward 1396 -> 1302 us per 8 renders (profiles/r06_blend_init_regime_ablation.txt).  The render's last
    // tile (ONE 64-bit atomic per tile carries the count of tiles done and the buckets they reached: whoever sees
    // T - 1 tiles before it holds the total) leaves the length for the next forward over this workspace.
    if (chain_arg == 0u) {
      const uint32_t n_tiles = gridDim.x;  // (a launch has one workgroup per tile and render)
      const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(meta + META_TICK),
                                               1ull | ((unsigned long long)nb << 32));
      if ((uint32_t)old == n_tiles - 1u) {
        const uint32_t total_nb = (uint32_t)(old >> 32) + nb;
        meta[META_CHAIN] = total_nb >= 8u * n_tiles ? 4u : (total_nb >= 4u * n_tiles ? 2u : 1u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------- backward
// (wave reduction: wave_ops.hpp)
//
// ONE WAVE per work item, four pixels per lane -- lane l owns pixel l of EACH 8x8 quadrant of the tile.  The 13
// per-record sums are accumulated over a lane's four pixels in registers and cross the wave ONCE per record (one
// halving reduction + one plain 16-lane LDS store: the wave is the only writer of its item), while quadrant culling
// stays wave-uniform: a quadrant the record cannot reach, or whose pixels all stopped earlier, is skipped by a scalar
// branch.
//
// A work item is a CHAIN of up to `chain` consecutive buckets of one tile (bucket = BUCKET list entries): the pixel
// state (transmittance, the running sums, the 8 gradient values of each of the lane's four pixels) is loaded once
// per item and carried through the chain in registers.  Only a chain that does not start at the head of the list
// reads a checkpoint; round 1 ran one bucket per item and re-read ~112 B of state per pixel for every bucket
// (1.86x the algorithmic traffic, SQ_WAIT_ANY a third of a wave's life).
//
// The per-pixel evaluation is predicated (selects on alpha and on g), not branched: with half of a quadrant's lanes
// inactive a branch saves no issue slot, and every `if` on a vector condition costs a VALU -> SALU round trip.
// Instruction classes are chosen by their measured issue cost (profiles/r02_valu_issue_rates.txt).
// Optional per-item trace of the backward (diagnostics; compiled in with -DDIMO_BWD_TRACE -- build.py does that when
// the environment has DIMO_BWD_TRACE=1 -- and then off unless dimo_debug_blend_trace set a buffer): 4 x u64 per
// item -- s_memrealtime (the chip-wide 100 MHz clock) at its start and end, (XCC id << 56 | render << 48 | item code), (records visited << 48 | quadrant visits <<
// 32 | HW_ID[31:16] | visits that some pixel used) -- appended with one atomic per item.
__device__ unsigned long long *g_bwd_trace = nullptr;
__device__ unsigned int g_bwd_trace_cap = 0;
__device__ unsigned int g_bwd_trace_n = 0;

struct BwdView {  // one render's buffers as the backward sees them
  const uint32_t *vals;
  const Splat *splat;
  const uint16_t *rect;
  const uint32_t *offsets;
  const float *final_T;
  const uint32_t *n_contrib;
  const float *final_acc;
  const float *ckpt;
  const uint32_t *work;
  const float *dL_dcolor, *dL_ddepth, *dL_dnormal, *dL_dalpha;
  SplatGrad *inst_grad;
  uint8_t *inst_flag;
  unsigned long long *hitmask;  // per Gaussian: bit k = its k-th instance has a record (k < 64; geom workspace)
  const float *dot;  // optional: per pixel sum over the channels of gradient x rendered value (= S below)
};

template <bool NORMAL>
__device__ __forceinline__ void blend_bwd_item(int H, int W, int tiles_x, uint32_t R_cap, const float *__restrict__ bg,
                                               const BwdView &r, uint4 it, float4 *s_geo, float4 *s_col, float4 *s_aux,
                                               float *s_nz, uint32_t *s_mask, float (*s_acc)[16], int render) {
  const int lane = threadIdx.x;
#ifdef DIMO_BWD_TRACE
  unsigned long long *const trace = g_bwd_trace;
#else
  unsigned long long *const trace = nullptr;  // (everything below that depends on it folds away)
#endif
  const unsigned long long t_start = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  uint32_t n_rec = 0, n_quad = 0, n_useful = 0;  // (n_useful: visits in which some pixel took the entry)
  const uint32_t HW32 = (uint32_t)H * (uint32_t)W;
  const uint32_t code = it.x, lo = it.y, hi = it.z;
  const int tile = (int)(code >> 12);
  const uint32_t b0 = code & 0xfffu, nbk = it.w;
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int bx = tile_x * TILE + (lane & 7), by = tile_y * TILE + (lane >> 3);
  const float bxf = (float)bx, byf = (float)by;

  // ---- pixel state of the lane's four pixels at the head of the chain.  Every load is unconditional (clamped pixel
  // index, substitute pointer for an absent gradient image) and the predicates are applied afterwards with selects:
  // a branch around a load makes the compiler drain the whole memory queue at the join.
  uint32_t last[4], deepest[4];
  float T[4], SP[4], dp[4][8];  // SP = S - P: what the entries behind the current one still add
  const float *const pc = r.dL_dcolor ? r.dL_dcolor : r.final_T, *const pd = r.dL_ddepth ? r.dL_ddepth : r.final_T;
  const float *const pn = (NORMAL && r.dL_dnormal) ? r.dL_dnormal : r.final_T;
  const float *const pa = r.dL_dalpha ? r.dL_dalpha : r.final_T;
  const float kc = r.dL_dcolor ? 1.0f : 0.0f, kd = r.dL_ddepth ? 1.0f : 0.0f;
  const float kn = (NORMAL && r.dL_dnormal) ? 1.0f : 0.0f, ka = r.dL_dalpha ? 1.0f : 0.0f;
  const uint32_t HWc = r.dL_dcolor ? HW32 : 0u, HWn = (NORMAL && r.dL_dnormal) ? HW32 : 0u;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const bool head = b0 == 0u;  // the chain starts at the head of the list: T = 1, nothing accumulated yet
  const float *const ck_item = r.ckpt + ((size_t)(lo / BUCKET) + tile + b0) * (CKPT_FLOATS * TILE * TILE);
  const uint32_t blo0 = b0 * BUCKET;
  // (the first bucket's list entries are requested here, in front of the pixel state: the records they name are
  // gathered right behind it -- one dependent memory round trip less per item; a chain requests the next bucket's
  // while it works on the current one)
  uint32_t next_g = blo0 + (uint32_t)lane < hi - lo ? r.vals[lo + blo0 + (uint32_t)lane] : 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int px = bx + (q & 1) * 8, py = by + (q >> 1) * 8;
    const bool inside = px < W && py < H;
    const uint32_t pix = inside ? (uint32_t)py * (uint32_t)W + (uint32_t)px : 0u;
    const uint32_t nc = r.n_contrib[pix];
    dp[q][0] = kc * pc[pix], dp[q][1] = kc * pc[HWc + pix], dp[q][2] = kc * pc[2u * HWc + pix];
    dp[q][3] = kd * pd[pix];
    dp[q][4] = kn * pn[pix], dp[q][5] = kn * pn[HWn + pix], dp[q][6] = kn * pn[2u * HWn + pix];
    dp[q][7] = ka * pa[pix];
    last[q] = inside ? nc : 0u;
    const bool need = last[q] > blo0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dp[q][k] = need ? dp[q][k] : 0.0f;
    // S = dL/dout . (final accumulators) + final_T (bg . dL/dcolour) = sum over the channels of gradient x RENDERED
    // value: the loss kernel hands it over as one plane (4 B per pixel); without it, from the forward's nine planes
    float S;
    if (r.dot) {  // wave-uniform
      S = need ? r.dot[pix] : 0.0f;
    } else {
      float fa[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) fa[k] = r.final_acc[(uint32_t)k * HW32 + pix];
      const float fT = r.final_T[pix];
      S = dp[q][7] * fa[7] + fT * (bg0 * dp[q][0] + bg1 * dp[q][1] + bg2 * dp[q][2]);
#pragma unroll
      for (int k = 0; k < (NORMAL ? 7 : 4); ++k) S += dp[q][k] * fa[k];
    }
    float P = 0.0f, Tq = 1.0f;
    if (!head) {  // wave-uniform
      const float *c = ck_item + (uint32_t)(q * 64 + lane);  // the forward's thread index = quadrant * 64 + lane
      float cv[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) cv[k] = c[k * TILE * TILE];
      P = dp[q][7] * cv[8];
#pragma unroll
      for (int k = 0; k < (NORMAL ? 7 : 4); ++k) P += dp[q][k] * cv[1 + k];
      Tq = cv[0];
    }
    T[q] = need ? Tq : 0.0f;
    SP[q] = need ? S - P : 0.0f;
    // deepest entry any pixel of this quadrant still looks at (wave-uniform)
    uint32_t m = last[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    deepest[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    // (round 2 kept one quadrant's loads in flight at a time with a scheduling barrier here: hoisting all four above
    // the first use spilled then.  It does not any more -- 120 VGPRs, no scratch -- and the four quadrants' ~76 loads in
    // flight together are three memory round trips less per item: 259.6 -> 255.9 us per 8-render launch, +0.7 % in
    // the step, profiles/r06_hit_mask_and_early_list_entries.txt)
  }
  const uint32_t wlast = max(max(deepest[0], deepest[1]), max(deepest[2], deepest[3]));

  for (uint32_t bk = 0; bk < nbk; ++bk) {
    const uint32_t blo = blo0 + bk * BUCKET;
    if (blo >= wlast) break;  // no pixel of the tile looks this deep
    const int count = (int)min((uint32_t)BUCKET, hi - lo - blo);
    // ---- stage the bucket's records COMPACTED: the records some quadrant can reach, in list order, slot = rank;
    // every lane keeps the rank / emission slot of the record it staged for the epilogue
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc4 = ra;
    float rnz = 0.0f;
    uint32_t qmask = 0, my_emit = 0, my_hit_word = 0;  // (Gaussian << 7 | min(instance of the Gaussian, 127))
    const uint32_t g = next_g;
    next_g = bk + 1u < nbk && blo + BUCKET + (uint32_t)lane < hi - lo ? r.vals[lo + blo + BUCKET + (uint32_t)lane] : 0u;
    if (lane < count) {
      const float4 *rp = reinterpret_cast<const float4 *>(r.splat + g);
      ra = rp[0], rb = rp[1], rc4 = rp[2];
      const float4 rd = rp[3];  // nz, A', B', C'
      rnz = rd.x;
      qmask = quadrant_mask(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, tile_x, tile_y);
      ra.z = rd.y, ra.w = rd.z, rb.x = rd.w;  // (staged pre-multiplied: see CONIC_HALF)
      const uint2 rc = *reinterpret_cast<const uint2 *>(r.rect + 4 * (size_t)g);
      const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
      const uint32_t local = (uint32_t)((tile_y - y0) * (x1 - x0) + (tile_x - x0));
      my_emit = (g == 0 ? 0u : r.offsets[g - 1]) + local;
      // (a Gaussian of up to 64 instances is found by its hit mask, a larger one by the record flags: preprocess.hip)
      my_hit_word = (g << 7) | ((x1 - x0) * (y1 - y0) <= 64 ? local : 127u);
    }
    const bool my_hit = qmask != 0u;
    const unsigned long long bal = __ballot(my_hit);
    const int my_rank = __popcll(bal & ((1ull << lane) - 1ull));
    const int mine = __popcll(bal);
    __syncthreads();  // the previous bucket's epilogue has read s_acc
    if (my_hit) {
      s_geo[my_rank] = ra, s_col[my_rank] = rb, s_aux[my_rank] = rc4;
      if (NORMAL) s_nz[my_rank] = rnz;
      s_mask[my_rank] = qmask | ((uint32_t)lane << 8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      reinterpret_cast<float4 *>(&s_acc[0][0])[k * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // Two register sets alternate (the loop is unrolled by two): the record of visit t + 1 -- mask word and all 13
    // floats -- is read from LDS into the idle set while visit t computes from the other, with no copies at the
    // back edge (a rotating single set cost 12 v_mov per record).
#define DIMO_BWD_LOAD(M, G, C, A, NZ, TT)                      \
  {                                                            \
    const int tt_ = min((TT), mine - 1);                       \
    M = s_mask[tt_], G = s_geo[tt_], C = s_col[tt_], A = s_aux[tt_]; \
    if (NORMAL) NZ = s_nz[tt_];                                \
  }
#define DIMO_BWD_VISIT(q, G, C, A, NZ)                                                                              \
    {                                                                                                              \
      const float dx = dx0 - (float)(((q) & 1) * 8), dy = dy0 - (float)(((q) >> 1) * 8);                           \
      const float power = fmaf(C.x * dy, dy, fmaf(G.w, dy, G.z * dx) * dx); /* log2(e) x the exponent */          \
      const float Gs = __builtin_amdgcn_exp2f(power);                                                              \
      const float alpha = fminf(ALPHA_MAX, C.y * Gs);                                                              \
      DIMO_BWD_PREDICATE(q)                                                                                        \
      if (trace && __ballot(active) != 0ull) ++n_useful; /* (trace builds only) */                                 \
      const float w = ae * T[q];                                                                                   \
      float D = dp[q][7] + C.z * dp[q][0] + C.w * dp[q][1] + A.x * dp[q][2] + A.y * dp[q][3];                      \
      if (NORMAL) D += A.z * dp[q][4] + A.w * dp[q][5] + NZ * dp[q][6];                                            \
      SP[q] -= D * w;                                                                                              \
      const float oma = 1.0f - ae;                                                                                 \
      const float dL_dalpha_i = D * T[q] - SP[q] * __builtin_amdgcn_rcpf(oma);                                     \
      T[q] *= oma;                                                                                                 \
      float gg = Gs * C.y * dL_dalpha_i; /* g = G * dL/dG, dL/dG = opacity * dL/dalpha */                           \
      gg = active ? gg : 0.0f;           /* (a select, not a product: G of a rejected lane may be inf) */            \
      const float gx = gg * dx, gy = gg * dy;                                                                      \
      v[0] += gg, v[1] += gx, v[2] += gy;                                                                          \
      v[3] += gx * dx, v[4] += gx * dy, v[5] += gy * dy;                                                           \
      v[6] += w * dp[q][0], v[7] += w * dp[q][1], v[8] += w * dp[q][2], v[9] += w * dp[q][3];                      \
      if (NORMAL) v[10] += w * dp[q][4], v[11] += w * dp[q][5], v[12] += w * dp[q][6];                             \
    }
#define DIMO_BWD_RECORD(M, G, C, A, NZ, TT)                                                                        \
  {                                                                                                                \
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)M);                                           \
    const uint32_t pos = blo + (m >> 8), qm = m & 0xfu;                                                            \
    const float dx0 = G.x - bxf, dy0 = G.y - byf;                                                                  \
    float v[16];                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                                               \
      v[k] = 0.0f;                                                                                                 \
      if (k < (NORMAL ? 13 : 10)) asm volatile("" : "+v"(v[k])); /* every quadrant accumulates the same way */      \
    }                                                                                                              \
    bool any = false;                                                                                              \
    DIMO_BWD_QUADRANTS(G, C, A, NZ)                                                                                \
    if (any) {                                                                                                     \
      ++n_rec;                                                                                                     \
      const float tot = wave_reduce16<13>(v); /* lane l: the wave total of value reduce16_slot(l); 13 in use */      \
      if ((lane & 3) == 0) s_acc[(TT)][reduce16_slot(lane)] = tot; /* this wave is the only writer of the record */ \
    } else if (pos >= wlast) {                                                                                     \
      break; /* the list is ascending: nothing further reaches this tile */                                         \
    }                                                                                                              \
  }
    /* Which quadrants a record is evaluated on: the ones its mask names and some pixel still looks at (wave-uniform
       branches).  Round 6 measured two other forms of this part and dropped both (profiles/r06_blend_bwd_visits.txt):
       BOTH quadrants of a tile half in ONE straight-line block whenever either qualifies (an unneeded quadrant
       evaluates to nothing, and two independent chains interleave): a visit got ~13 % cheaper, but 3.35 instead of
       2.51 of them per record: 283 against 257 us per 8-render launch, 1483 against 1315 in the init regime; and the
       three rejection tests as a chain of selects on alpha instead of compares joined on the scalar unit: equal
       (254.9 / 1317 us). */
#define DIMO_BWD_NEED(q) (((qm >> (q)) & 1u) && pos < deepest[(q)])
#define DIMO_BWD_QUADRANTS(G, C, A, NZ)                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                         \
      if (!DIMO_BWD_NEED(q)) continue;                                       \
      any = true;                                                           \
      ++n_quad;                                                             \
      DIMO_BWD_VISIT(q, G, C, A, NZ)                                         \
    }
#define DIMO_BWD_PREDICATE(q)                                                          \
      const bool active = pos < last[q] && power <= 0.0f && alpha >= ALPHA_MIN;       \
      const float ae = active ? alpha : 0.0f;
    uint32_t m0 = 0, m1 = 0;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), c0 = g0, a0 = g0, g1 = g0, c1 = g0, a1 = g0;
    float nz0 = 0.0f, nz1 = 0.0f;
    if (mine > 0) DIMO_BWD_LOAD(m0, g0, c0, a0, nz0, 0)
    for (int t = 0; t < mine; t += 2) {
      DIMO_BWD_LOAD(m1, g1, c1, a1, nz1, t + 1)
      DIMO_BWD_RECORD(m0, g0, c0, a0, nz0, t)
      if (t + 1 >= mine) break;
      DIMO_BWD_LOAD(m0, g0, c0, a0, nz0, t + 2)
      DIMO_BWD_RECORD(m1, g1, c1, a1, nz1, t + 1)
    }
#undef DIMO_BWD_LOAD
#undef DIMO_BWD_RECORD
#undef DIMO_BWD_PREDICATE
#undef DIMO_BWD_QUADRANTS
#undef DIMO_BWD_NEED
#undef DIMO_BWD_VISIT
    __syncthreads();
    if (my_hit) {
      const float4 *src = reinterpret_cast<const float4 *>(&s_acc[my_rank][0]);
      const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
      const bool nonzero = r0.x != 0.f || r0.y != 0.f || r0.z != 0.f || r0.w != 0.f || r1.x != 0.f || r1.y != 0.f ||
                           r1.z != 0.f || r1.w != 0.f || r2.x != 0.f || r2.y != 0.f || r2.z != 0.f || r2.w != 0.f ||
                           r3.x != 0.f;
      if (my_emit < R_cap && nonzero) {
        float4 *dst = reinterpret_cast<float4 *>(r.inst_grad + my_emit);
        dst[0] = r0, dst[1] = r1, dst[2] = r2, dst[3] = r3;
        // the projection backward finds a Gaussian's records by ONE word (its instances are contiguous from
        // offsets[g - 1]): two dependent rounds of scattered flag bytes per four instances before (round 6)
        if ((my_hit_word & 127u) < 64u) atomicOr(r.hitmask + (my_hit_word >> 7), 1ull << (my_hit_word & 127u));
        else r.inst_flag[my_emit] = 1;
      }
    }
  }
  if (trace && lane == 0) {
    const unsigned int at_ = atomicAdd(&g_bwd_trace_n, 1u);
    if (at_ < g_bwd_trace_cap) {
      unsigned int hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long *t = trace + 4ull * at_;
      t[0] = t_start, t[1] = __builtin_amdgcn_s_memrealtime();
      t[2] = ((unsigned long long)(xcc & 0xfu) << 56) | ((unsigned long long)render << 48) | code;
      t[3] = ((unsigned long long)n_rec << 48) | ((unsigned long long)n_quad << 32) | (hw & 0xffff0000u) | (n_useful & 0xffffu);
    }
  }
}

// Items are taken class by class -- the deep chains, then the second ones, then the heads (see the forward's queues)
// -- in the hardware's in-order workgroup dispatch over blockIdx: no atomics.  Measured on the C3 batch: deep-first
// 196-203 us per launch, heads-first 200-223 us (the sparse deep items are latency bound and mix well with the
// VALU-bound heads when they start together; started last they leave the chip half idle).  Inside a class the items are
// taken from the FRONT of the queue: the forward starts its tiles longest list first, so the queues fill roughly by
// descending list length and the launch ends on its cheapest items (from the back -- the order of round 2, when
// the queues filled in tile order -- the blend backward took 278 us per 8 renders, from the front 270).  Virtual
// item v = blockIdx.x, + gridDim.x, ...: render v % n, that render's (v / n)-th item, so the renders of a batch
// interleave.
// (Round 4 measured every other order of the three classes -- heads / second / deep first, a proportional interleave of
// the heads with the rest -- and grids of 4 096 / 8 192 / 12 288 workgroups: 259-316 us per 8-render launch against
// 256.5 for this one on 16 384; the switches are gone.)
template <bool NORMAL, class VIEW>
__device__ __forceinline__ void blend_bwd_loop(int H, int W, int tiles_x, uint32_t R_cap, const float *__restrict__ bg,
                                               int n, VIEW view) {
  __shared__ float4 s_geo[BUCKET];
  __shared__ float4 s_col[BUCKET];
  __shared__ float4 s_aux[BUCKET];
  __shared__ float s_nz[BUCKET];
  __shared__ uint32_t s_mask[BUCKET];
  __shared__ float s_acc[BUCKET][16];
  static_assert(BUCKET == 64, "one staged record per lane");
  const uint32_t T = (uint32_t)tiles_x * (uint32_t)((H + TILE - 1) / TILE);
  uint32_t most = 0;
  for (int i = 0; i < n; ++i) {
    const uint32_t *w = view(i).work;
    most = max(most, w[0] + w[1] + w[2]);
  }
  for (uint32_t v = blockIdx.x; v < most * (uint32_t)n; v += gridDim.x) {
    const BwdView r = view((int)(v % (uint32_t)n));
    const uint32_t local = v / (uint32_t)n, c0 = r.work[0], c1 = r.work[1], c2 = r.work[2];
    if (local >= c0 + c1 + c2) continue;
    const uint32_t slot = local < c2 ? 2u * T + local : (local < c2 + c1 ? T + (local - c2) : local - c2 - c1);
    const uint4 it = reinterpret_cast<const uint4 *>(r.work)[1 + slot];
    blend_bwd_item<NORMAL>(H, W, tiles_x, R_cap, bg, r, it, s_geo, s_col, s_aux, s_nz, s_mask, s_acc,
                           (int)(v % (uint32_t)n));
  }
}

// ---------------------------------------------------------------------------------- kernel entry points
template <bool NORMAL>
__global__ void __launch_bounds__(BLEND_BLOCK) blend_fwd_kernel(
    int H, int W, int tiles_x, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ vals_sorted,
    const Splat *__restrict__ splat, const float *__restrict__ bg, float *__restrict__ out_color,
    float *__restrict__ out_depth, float *__restrict__ out_normal, float *__restrict__ out_alpha,
    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib, float *__restrict__ final_acc,
    float *__restrict__ ckpt, uint32_t *__restrict__ work, uint32_t chain) {
  blend_fwd_body<NORMAL>(H, W, tiles_x, ranges, vals_sorted, splat, bg, out_color, out_depth, out_normal, out_alpha,
                         final_T, n_contrib, final_acc, ckpt, work, chain, nullptr, (int)blockIdx.x);
}
// (Round 4 also built the visit's eight accumulations on the matrix pipe -- two v_mfma_f32_4x4x1_16b_f32 per visit,
// accumulators transposed: 27 vector instructions + 2 MFMAs instead of 37, parity identical -- and measured it slower,
// 124 us against 115 per 4-render launch in the step: a 4x4x1 MFMA holds the wave's issue for its two passes, which is
// what its four v_fmac_f32 cost.  profiles/r04_blend_fwd_mfma.txt; removed in round 5.)
struct SingleView {
  BwdView v;
  __device__ __forceinline__ const BwdView &operator()(int) const { return v; }
};
template <bool NORMAL>
__global__ void __launch_bounds__(64, BWD_WAVES_PER_SIMD) blend_bwd_kernel(int H, int W, int tiles_x, uint32_t R_cap,
                                                                            const float *__restrict__ bg, SingleView sv) {
  blend_bwd_loop<NORMAL>(H, W, tiles_x, R_cap, bg, 1, sv);
}

// Batched entry points (native step executor): blockIdx.y = render of the batch (forward); the backward interleaves
// the renders' items over a one-dimensional grid.
struct BlendOffsets {
  size_t splat, rect, offsets, total;           // geom
  size_t ranges, vals, ckpt, work, order, meta; // bin
  size_t final_T, n_contrib, final_acc;         // img
  size_t flag;                                  // backward scratch: records at 0, flags here
  size_t hit;                                   // geom: per-Gaussian hit masks
};
template <bool NORMAL>
__global__ void __launch_bounds__(BLEND_BLOCK) blend_fwd_batched_kernel(int H, int W, int tiles_x,
                                                                        const float *__restrict__ bg, BlendOffsets o,
                                                                        uint32_t chain, RenderBatch b) {
  // Workgroups start in the order of their linear index: the renders of the batch interleaved, and inside a render the
  // tiles by descending list length (the order the level-2 fill left in the bin workspace) -- the long lists of every
  // render first, the empty tiles of the border as the launch's tail.
  const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x;
  const dimo_render_desc &r = b.r[lin % gridDim.y];
  const int tile = (int)at<uint32_t>(r.bin, o.order)[lin / gridDim.y];
  blend_fwd_body<NORMAL>(H, W, tiles_x, at<uint32_t>(r.bin, o.ranges), at<uint32_t>(r.bin, o.vals),
                         at<Splat>(r.geom, o.splat), bg, r.out_color, r.out_depth, NORMAL ? r.out_normal : nullptr,
                         r.out_alpha, at<float>(r.img, o.final_T), at<uint32_t>(r.img, o.n_contrib),
                         at<float>(r.img, o.final_acc), at<float>(r.bin, o.ckpt), at<uint32_t>(r.bin, o.work),
                         chain, at<uint32_t>(r.bin, o.meta), tile);
}
template <bool NORMAL>
struct BatchView {
  const RenderBatch &b;
  const BlendOffsets &o;
  __device__ __forceinline__ BwdView operator()(int i) const {
    const dimo_render_desc &r = b.r[i];
    return BwdView{at<uint32_t>(r.bin, o.vals),      at<Splat>(r.geom, o.splat),       at<uint16_t>(r.geom, o.rect),
                   at<uint32_t>(r.geom, o.offsets),  at<float>(r.img, o.final_T),      at<uint32_t>(r.img, o.n_contrib),
                   at<float>(r.img, o.final_acc),    at<float>(r.bin, o.ckpt),         at<uint32_t>(r.bin, o.work),
                   r.g_color,                        r.g_depth,                        NORMAL ? r.g_normal : nullptr,
                   r.g_alpha,                        reinterpret_cast<SplatGrad *>(r.bwd_scratch),
                   at<uint8_t>(r.bwd_scratch, o.flag), at<unsigned long long>(r.geom, o.hit), r.g_dot};
  }
};
// JOINT only names the launch (no code depends on it): the joint launch over ALL the step's renders on the caller's
// stream -- the launch bench.py takes its roofline clock on, alone on the device -- appears in a kernel trace as
// blend_bwd_batched_kernel<.., true>, apart from the per-motion launches of the default schedule (<.., false>), which
// overlap the other motion's kernels and read longer than they are.
template <bool NORMAL, bool JOINT>
__global__ void __launch_bounds__(64, BWD_WAVES_PER_SIMD) blend_bwd_batched_kernel(int H, int W, int tiles_x,
                                                                                    uint32_t R_cap,
                                                                                    const float *__restrict__ bg,
                                                                                    BlendOffsets o, int n,
                                                                                    RenderBatch b) {
  blend_bwd_loop<NORMAL>(H, W, tiles_x, R_cap, bg, n, BatchView<NORMAL>{b, o});
}
// Buckets per backward item.  Rounds 2-5 ran ONE: measured on the trained C3 batch (4 renders, ~10^4 buckets for 4096
// wave slots) 189 / 231 / 245 / 273 us per launch at 1 / 2 / 3 / 4 buckets per item -- what a longer chain saves in state
// loads it loses several times over in the tail of the launch.  Round 6 measured the reference's INITIAL state (every
// opacity 0.05: every pixel walks its whole list, 127 000 items per 8-render launch): 1396 / 1314 / 1302 us at 1 / 2 / 4
// (forward 659 / 635 / 629), the step +4.9 % / +6.9 %, and the trained step -2.5 % at 2.  So the length follows the
// lists: the batched forward picks it per render slot from what the slot's previous render looked like (blend_fwd_body).
#ifndef DIMO_BWD_CHAIN
#define DIMO_BWD_CHAIN 0
#endif
constexpr uint32_t BWD_CHAIN = DIMO_BWD_CHAIN;          // batched launches: 0 = adaptive (a build may fix 1, 2 or 4)
constexpr uint32_t BWD_CHAIN_SINGLE = DIMO_BWD_CHAIN ? DIMO_BWD_CHAIN : 1;  // single-render launches (fresh workspaces)

static BlendOffsets blend_offsets(const GeomLayout &G, const BinLayout &B, const ImgLayout &I) {
  BlendOffsets o;
  o.splat = G.splat, o.rect = G.rect, o.offsets = G.offsets, o.total = G.total;
  o.ranges = B.ranges, o.vals = B.vals_b, o.ckpt = B.ckpt, o.work = B.work, o.order = B.order, o.meta = B.meta;
  o.final_T = I.final_T, o.n_contrib = I.n_contrib, o.final_acc = I.final_acc;
  o.flag = align_up(B.cap * sizeof(SplatGrad));
  o.hit = G.hitmask;
  return o;
}

int blend_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W, c.N);
  ImgLayout I(c.H, c.W);
  if (c.bin_bytes < B.bytes || c.img_bytes < I.bytes) return DIMO_E_WORKSPACE;
  const BlendOffsets o = blend_offsets(G, B, I);
  const uint32_t chain = BWD_CHAIN;  // (0: adaptive, the length the slot's previous forward left in its workspace)
  ScopedTimer tm(T_BLEND_FWD, stream);
#define DIMO_LAUNCH_FWD(N_)                                                                                    \
  hipLaunchKernelGGL((blend_fwd_batched_kernel<N_>), dim3(B.T, n), dim3(BLEND_BLOCK), 0, stream, c.H, c.W, B.tiles_x, \
                     c.bg, o, chain, b)
  if (c.with_normal) DIMO_LAUNCH_FWD(true);
  else DIMO_LAUNCH_FWD(false);
#undef DIMO_LAUNCH_FWD
  return check_launch();
}

int blend_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream, bool joint) {
  if (n <= 0 || c.N <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W, c.N);
  ImgLayout I(c.H, c.W);
  if (c.bwd_scratch_bytes < align_up(B.cap * sizeof(SplatGrad)) + align_up(B.cap)) return DIMO_E_WORKSPACE;
  const BlendOffsets o = blend_offsets(G, B, I);
  const uint32_t cap = (uint32_t)B.cap;
  // (the "record written" flags were cleared by the placement's fill pass of this batch's forward: binning.hip)
  ScopedTimer tm(T_BLEND_BWD, stream);
#define DIMO_LAUNCH_BWD(N_, J_)                                                                                     \
  hipLaunchKernelGGL((blend_bwd_batched_kernel<N_, J_>), dim3(BWD_GRID), dim3(64), 0, stream, c.H, c.W, B.tiles_x, \
                     cap, c.bg, o, n, b)
  if (c.with_normal) {
    if (joint) DIMO_LAUNCH_BWD(true, true);
    else DIMO_LAUNCH_BWD(true, false);
  } else {
    if (joint) DIMO_LAUNCH_BWD(false, true);
    else DIMO_LAUNCH_BWD(false, false);
  }
#undef DIMO_LAUNCH_BWD
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
  BinLayout B(R_cap, H, W, N);
  ImgLayout I(H, W);
  if (bin_bytes < B.bytes || img_bytes < I.bytes) return DIMO_E_WORKSPACE;
  int rc = bin_instances(N, H, W, R_cap, geom, bin, stream);
  if (rc) return rc;
  const uint32_t *ranges = at<uint32_t>(bin, B.ranges);
  const uint32_t *vals = at<uint32_t>(bin, B.vals_b);
  const Splat *splat = at<Splat>(geom, G.splat);
  ScopedTimer tm(T_BLEND_FWD, stream);
#define DIMO_LAUNCH_FWD(N_)                                                                                     \
  hipLaunchKernelGGL((blend_fwd_kernel<N_>), dim3(B.T), dim3(BLEND_BLOCK), 0, stream, H, W, B.tiles_x, ranges, vals, \
                     splat, bg, out_color, out_depth, out_normal, out_alpha, at<float>(img, I.final_T),              \
                     at<uint32_t>(img, I.n_contrib), at<float>(img, I.final_acc), at<float>(bin, B.ckpt),            \
                     at<uint32_t>(bin, B.work), BWD_CHAIN_SINGLE)
  if (out_normal) DIMO_LAUNCH_FWD(true);
  else DIMO_LAUNCH_FWD(false);
#undef DIMO_LAUNCH_FWD
  return check_launch();
}

// Diagnostic: per-item trace of the blend backward (see g_bwd_trace).  buffer = device memory for `capacity` records
// of 4 x u64, or null to switch the trace off; returns the number of records written since the last call.
extern "C" int64_t dimo_debug_blend_trace(void *buffer, int64_t capacity) {
#ifndef DIMO_BWD_TRACE
  if (buffer) return DIMO_E_ARG;  // not compiled in
#endif
  unsigned int n = 0, zero = 0, cap = (unsigned int)(capacity > 0 ? capacity : 0);
  unsigned long long *p = reinterpret_cast<unsigned long long *>(buffer);
  if (hipDeviceSynchronize() != hipSuccess) return DIMO_E_LAUNCH;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_bwd_trace_n), sizeof(n)) != hipSuccess) return DIMO_E_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace_n), &zero, sizeof(zero)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace_cap), &cap, sizeof(cap)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), &p, sizeof(p)) != hipSuccess)
    return DIMO_E_LAUNCH;
  return (int64_t)n;
}

// Diagnostic: the wave reduction of the backward on caller-supplied data (in: 64 lanes x 16 floats, lane-major;
// out: 3 x 16 wave totals -- all 16 values in use, then the 13-value and the 10-value forms the blend backward runs,
// whose slots 13 .. 15 / 10 .. 15 are unspecified).  Lets a test pin the DPP / permlane sequences of wave_ops.hpp
// against a plain sum.
__global__ void __launch_bounds__(64) wave_reduce16_selftest_kernel(const float *__restrict__ in, float *__restrict__ out) {
  float v[16], w[16], u[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = w[k] = u[k] = in[threadIdx.x * 16 + k];
  const float t16 = dimo::wave_reduce16<16>(v), t13 = dimo::wave_reduce16<13>(w), t10 = dimo::wave_reduce16<10>(u);
  if ((threadIdx.x & 3) == 0) {
    const int s = dimo::reduce16_slot(threadIdx.x);
    out[s] = t16, out[16 + s] = t13, out[32 + s] = t10;
  }
}
extern "C" int dimo_selftest_wave_reduce16(const float *in, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  clear_errors();
  if (!in || !out) return DIMO_E_ARG;
  hipLaunchKernelGGL(wave_reduce16_selftest_kernel, dim3(1), dim3(64), 0, stream, in, out);
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
  BinLayout B(R_cap, H, W, N);
  ImgLayout I(H, W);
  const uint32_t cap = (uint32_t)B.cap;
  SplatGrad *inst = reinterpret_cast<SplatGrad *>(scratch);
  uint8_t *inst_flag = reinterpret_cast<uint8_t *>(scratch) + align_up(B.cap * sizeof(SplatGrad));
  if (N > 0) {
    ScopedTimer tm(T_BLEND_BWD, stream);
    if (hipMemsetAsync(inst_flag, 0, B.cap, stream) != hipSuccess) return DIMO_E_LAUNCH;
    unsigned long long *hitmask = at<unsigned long long>(const_cast<void *>(geom), G.hitmask);
    if (hipMemsetAsync(hitmask, 0, (size_t)N * sizeof(unsigned long long), stream) != hipSuccess) return DIMO_E_LAUNCH;
    SingleView sv{BwdView{at<uint32_t>(bin, B.vals_b), at<Splat>(geom, G.splat), at<uint16_t>(geom, G.rect),
                          at<uint32_t>(geom, G.offsets), at<float>(img, I.final_T), at<uint32_t>(img, I.n_contrib),
                          at<float>(img, I.final_acc), at<float>(bin, B.ckpt), at<uint32_t>(bin, B.work), dL_dcolor,
                          dL_ddepth, dL_dnormal, dL_dalpha, inst, inst_flag, hitmask, nullptr}};
    if (dL_dnormal)
      hipLaunchKernelGGL(blend_bwd_kernel<true>, dim3(BWD_GRID), dim3(64), 0, stream, H, W, B.tiles_x, cap, bg, sv);
    else
      hipLaunchKernelGGL(blend_bwd_kernel<false>, dim3(BWD_GRID), dim3(64), 0, stream, H, W, B.tiles_x, cap, bg, sv);
    int rc = check_launch();
    if (rc) return rc;
  }
  return preprocess_backward_launch(N, sh_degree, M, H, W, R_cap, means3D, shs, colors_precomp, scales, rotations,
                                    cov3D_precomp, scale_modifier, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                                    radii, geom, scratch, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacity,
                                    dL_dscales, dL_drotations, dL_dcov3D, stream);
}
