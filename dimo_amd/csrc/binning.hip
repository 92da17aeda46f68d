// Tile binning: produces, per tile, the list of Gaussians that touch it in (depth, id) order -- bit for bit the
// arrays a stable radix sort of the (tile | fp32 depth bits) keys of all (Gaussian, tile) instances yields (the
// published rasterizer's cub::DeviceRadixSort) -- without ever sorting the instances:
//   1. the Gaussians of the frame that touch a tile are sorted by (their 32 depth bits, index): coarse bins over the
//      key range, each sorted in LDS by one workgroup (see "depth sort" below);
//   2. ORDERED FILTERS cut that list first into per-supertile lists, then every supertile list into the lists of
//      its tiles (count pass, scan, fill pass per level: see "placement" below).  A filter keeps the input order,
//      so every tile list comes out in (depth, id) order, and all writes are long contiguous runs.
// A frame has ~10x fewer Gaussians than instances (1e5 vs 1e6 at the benchmark configuration): sorting the
// Gaussians and filtering replaces 6 radix passes over the instances (first version), later 2 passes + a per-tile
// LDS sort, each a chain of launches whose cost was latency rather than bandwidth.
//
// Everything reads live counts from device memory (geom.total) and is sized by capacities, so the whole chain is
// enqueued without a host round trip.
#include "common.hpp"

namespace dimo {

// Depth-bin map (see "depth sort"): keys at or below `lo` fall into bin 0, keys past the last bin into the last one
__device__ __forceinline__ uint32_t depth_bin(uint32_t key, uint32_t lo, uint32_t shift, uint32_t nbins) {
  return key <= lo ? 0u : min((key - lo) >> shift, nbins - 1u);
}

// Sub-bins of a bin: 256 over its 2^shift keys; bin 0 also holds every key below the binned range, so its sub-bins
// start at the smallest key and are correspondingly wider.
__device__ __forceinline__ void sub_bin_map(const uint32_t *bk, uint32_t bin, uint32_t &klo, uint32_t &sub_shift) {
  const uint32_t shift = bk[BK_SHIFT];
  klo = bk[BK_KMIN] + (bin << shift);
  uint32_t width = 1u << shift;
  if (bin == 0u) {
    width += klo - bk[BK_KMIN0];
    klo = bk[BK_KMIN0];
  }
  const int nbits = 32 - __clz((int)(width - 1u) | 1);
  sub_shift = (uint32_t)(nbits > 8 ? nbits - 8 : 0);
}

// ------------------------------------------------------------------------------------ scan
// Exclusive scan of the per-block sums (nb <= a few thousand) by one workgroup; writes the
// grand total R to total[0] and clears the overflow flag total[1].
__device__ __forceinline__ void scan_block_sums_body(int nb, int N, uint32_t *__restrict__ sums,
                                                     uint32_t *__restrict__ total, uint32_t *__restrict__ bk) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  __shared__ uint32_t wave_mn[16], wave_mx[16], wave_a[16], wave_b[16];
  if (threadIdx.x == 0) carry_s = 0;
  // the depth sort's bin map (see "depth sort" below).  The binned range is NOT [min, max] of the keys: a few floaters
  // far behind (or in front of) the scene would stretch it until the scene's bulk shares a handful of bins.  A block's
  // MIN ignores a far outlier inside it and its MAX a near one, so A = the largest block minimum and B = the smallest
  // block maximum bracket the bulk whatever the order of the Gaussians (Morton order: A ~ far end, B ~ near end;
  // random order: the other way round); the range is [min(A, B), max(A, B)] widened by a quarter on both sides,
  // inside [min, max].  Keys outside it fall into the first / last bin (whose sub-bins start at the smallest key).
  {
    uint32_t mn = 0xffffffffu, mx = 0u, a_ = 0u, b_ = 0xffffffffu;
    for (int i = threadIdx.x; i < nb; i += 1024) {
      const uint32_t bmin = sums[(nb + 1) + i], bmax = sums[2 * (nb + 1) + i];
      if (bmin != 0xffffffffu) mn = min(mn, bmin), mx = max(mx, bmax), a_ = max(a_, bmin), b_ = min(b_, bmax);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = min(mn, (uint32_t)__shfl_down((int)mn, o, 64));
      mx = max(mx, (uint32_t)__shfl_down((int)mx, o, 64));
      a_ = max(a_, (uint32_t)__shfl_down((int)a_, o, 64));
      b_ = min(b_, (uint32_t)__shfl_down((int)b_, o, 64));
    }
    if ((threadIdx.x & 63) == 0)
      wave_mn[threadIdx.x >> 6] = mn, wave_mx[threadIdx.x >> 6] = mx, wave_a[threadIdx.x >> 6] = a_,
      wave_b[threadIdx.x >> 6] = b_;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t mn = wave_mn[0], mx = wave_mx[0], a_ = wave_a[0], b_ = wave_b[0];
    for (int w = 1; w < 16; ++w)
      mn = min(mn, wave_mn[w]), mx = max(mx, wave_mx[w]), a_ = max(a_, wave_a[w]), b_ = min(b_, wave_b[w]);
    uint32_t lo = mn, span = 0u;
    if (mx >= mn) {  // (some Gaussian touches a tile)
      const uint32_t rl = min(a_, b_), rh = max(a_, b_), ext = (rh - rl) >> 2;
      lo = rl - mn > ext ? rl - ext : mn;
      const uint32_t hi = mx - rh > ext ? rh + ext : mx;
      span = hi - lo;
    }
    const int nbits = span ? 32 - __clz((int)span) : 0, lg = depth_bins_log2(N);
    bk[BK_KMIN] = lo, bk[BK_SHIFT] = (uint32_t)(nbits > lg ? nbits - lg : 0), bk[BK_KMIN0] = min(mn, lo);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < nb) sums[i] = carry + wave_off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[nb] = carry_s;
    total[0] = carry_s;
    total[1] = 0;
    total[2] = 0;
    total[3] = 0;  // number of depth-sorted Gaussians (depth_bin_scan writes it)
  }
}

// offsets[i] = inclusive scan of tiles_touched (block prefix + in-block scan)
// ... and the input of the depth sort: key = depth bits (0xffffffff for a Gaussian without tiles), value = id
__device__ __forceinline__ void write_offsets_body(int N, const uint32_t *__restrict__ tiles,
                                                   const uint32_t *__restrict__ block_prefix,
                                                   uint32_t *__restrict__ offsets, const Splat *__restrict__ splat,
                                                   uint64_t *__restrict__ nkeys, uint32_t *__restrict__ nvals,
                                                   uint32_t *__restrict__ bk) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / 64];
  __shared__ uint32_t s_hist[NC_MAX];
  const int nc = 1 << depth_bins_log2(N);
  if ((int)threadIdx.x < nc) s_hist[threadIdx.x] = 0u;  // (PRE_BLOCK >= NC_MAX)
  const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t v = i < N ? tiles[i] : 0u;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t off = block_prefix[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += wave_tot[w];
  if (i < N) {
    offsets[i] = off + inc;
    const uint32_t key = __float_as_uint(splat[i].depth);
    nkeys[i] = v ? (uint64_t)key : 0xffffffffull;
    nvals[i] = (uint32_t)i;
    // this block's share of its depth bin (the clamp is insurance: min / max come from the same keys)
    if (v) atomicAdd(&s_hist[depth_bin(key, bk[BK_KMIN], bk[BK_SHIFT], (uint32_t)nc)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < nc) bk[BK_HIST + (size_t)blockIdx.x * nc + threadIdx.x] = s_hist[threadIdx.x];
}

// ------------------------------------------------------------------------------------ depth sort
// The Gaussians of a frame that touch a tile, ordered by (depth bits, index) -- what a stable sort by the 32 depth bits
// gives.  A frame has ~1e5 of them: an LSD radix sort was 4 passes x 3 launches of pure launch / ramp latency (99 us
// per batch of 8 renders), and device-scope atomics (a first bucket sort: 2 per key) are slow on this chip (one
// counter bump per key cost 27 us per batch).  So, without a single global atomic:
//   1. preprocess leaves the min / max key per block and scan_block_sums derives the bin map (128 or 256 coarse bins
//      of width 2^shift over a range that brackets the BULK of the keys -- not [min, max]: see there --, monotone in
//      the key); write_offsets counts each block's keys per bin in LDS;
//   2. depth_bin_scan (one workgroup): column sums over the blocks -> bin bases, every block's first slot per bin;
//   3. depth_bin_scatter: a block drops its (key, id) pairs into their bins (LDS cursor per bin: unordered inside);
//   4. depth_bin_sort: ONE WORKGROUP PER BIN (~1 600 entries at 1e5 Gaussians) sorts its bin in LDS: a counting pass
//      over 256 sub-bins of the bin's key range, then every entry ranks itself inside its sub-bin (a handful of
//      entries) by the 64-bit word (key << 32 | id) -- all distinct because the ids are.  A bin above 6144 entries is
//      cut along its sub-bins into slices for extra workgroups of the same launch; only a sub-bin above 2048 entries
//      (thousands of Gaussians at nearly ONE depth) sends its bin to nine stable byte passes of a single workgroup
//      through global memory: slow, correct, rare.
// The result does not depend on the order the LDS atomics resolved in.
constexpr int BIN_CAP = 6144;      // entries a workgroup sorts in LDS (48 KB)
constexpr int SUB_BINS = 256;
constexpr int SUB_MAX = 2048;      // largest sub-bin ranked quadratically
// A bin above BIN_CAP (a few far outliers stretch [min, max] and the scene's bulk lands in a handful of bins) is cut
// into SLICES of ~SLICE_TARGET entries along its sub-bins, one extra workgroup each: without them 8 floaters 30 units
// behind a 1e5-Gaussian scene cost 0.6 ms per render in the single-workgroup fallback.
constexpr int SLICE_TARGET = BIN_CAP - SUB_MAX;
constexpr int SLICE_GRID = 64;     // extra workgroups of a launch that work the slice list off

// one workgroup of 1024 threads: thread (part, c) first sums its slice of column c of the [blocks][bins] counts, then
// (bases known) rewrites the slice as each block's first slot
__device__ __forceinline__ void depth_bin_scan_body(int N, uint32_t *__restrict__ bk, uint32_t *__restrict__ total) {
  __shared__ uint32_t s_part[16][NC_MAX];
  __shared__ uint32_t s_scan[NC_MAX];
  const int lg = depth_bins_log2(N), nc = 1 << lg, parts = 1024 >> lg;
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  const int c = threadIdx.x & (nc - 1), part = threadIdx.x >> lg;
  const int per = (nb + parts - 1) / parts;
  const int b_lo = min(nb, part * per), b_hi = min(nb, b_lo + per);
  uint32_t *p = bk + BK_HIST + c;
  uint32_t sum = 0;
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = b0 + j < b_hi ? p[(size_t)(b0 + j) * nc] : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += v[j];
  }
  s_part[part][c] = sum;
  __syncthreads();
  uint32_t before = 0, len = 0;
  for (int q = 0; q < parts; ++q) {
    before += q < part ? s_part[q][c] : 0u;
    len += s_part[q][c];
  }
  if (part == 0) s_scan[c] = len;
  __syncthreads();
  for (int o = 1; o < nc; o <<= 1) {  // Hillis-Steele inclusive scan of the bin sizes
    const uint32_t add = (part == 0 && c >= o) ? s_scan[c - o] : 0u;
    __syncthreads();
    if (part == 0) s_scan[c] += add;
    __syncthreads();
  }
  const uint32_t base = s_scan[c] - len;
  __shared__ uint32_t s_nslice;
  if (threadIdx.x == 0) s_nslice = 0u;
  __syncthreads();
  if (part == 0) {
    bk[BK_BASE + c] = base;
    if (c == nc - 1) bk[BK_BASE + nc] = s_scan[c], total[3] = s_scan[c];
    uint32_t J = 0;
    if (len > (uint32_t)BIN_CAP) {  // an oversized bin: its slices go on the list (if they fit; else J stays 0)
      const uint32_t want = (len + SLICE_TARGET - 1) / SLICE_TARGET;
      const uint32_t pos = want <= 255u ? atomicAdd(&s_nslice, want) : (uint32_t)MAX_SLICES;
      if (pos + want <= (uint32_t)MAX_SLICES) {
        J = want;
        for (uint32_t j = 0; j < want; ++j) bk[BK_SLICE + pos + j] = ((uint32_t)c << 16) | (want << 8) | j;
      } else if (want <= 255u) {
        // (list full: the entries this bin reserved stay unused; mark them so)
        for (uint32_t j = pos; j < min(pos + want, (uint32_t)MAX_SLICES); ++j) bk[BK_SLICE + j] = 0xffffffffu;
      }
    }
    bk[BK_BINJ + c] = J;
  }
  __syncthreads();
  if (threadIdx.x == 0) bk[BK_NSLICE] = min(s_nslice, (uint32_t)MAX_SLICES);
  uint32_t run = base + before;
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = b0 + j < b_hi ? p[(size_t)(b0 + j) * nc] : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (b0 + j < b_hi) p[(size_t)(b0 + j) * nc] = run;
      run += v[j];
    }
  }
}

// blocks of PRE_BLOCK keys, the SAME blocks write_offsets counted
__device__ __forceinline__ void depth_bin_scatter_body(int N, const uint64_t *__restrict__ keys_in,
                                                       uint64_t *__restrict__ keys_out,
                                                       uint32_t *__restrict__ vals_out,
                                                       const uint32_t *__restrict__ bk) {
  __shared__ uint32_t s_cur[NC_MAX];
  const int nc = 1 << depth_bins_log2(N);
  if ((int)threadIdx.x < nc) s_cur[threadIdx.x] = bk[BK_HIST + (size_t)blockIdx.x * nc + threadIdx.x];
  __syncthreads();
  const int idx = blockIdx.x * PRE_BLOCK + (int)threadIdx.x;
  if (idx >= N) return;
  const uint64_t k = keys_in[idx];
  if ((uint32_t)k == 0xffffffffu) return;  // touches no tile: not sorted at all
  const uint32_t pos = atomicAdd(&s_cur[depth_bin((uint32_t)k, bk[BK_KMIN], bk[BK_SHIFT], (uint32_t)nc)], 1u);
  keys_out[pos] = k;
  vals_out[pos] = (uint32_t)idx;
}

// One stable counting pass of a single workgroup over n (key, id) pairs: digit = byte `byte` of the 9-byte word
// (id bytes 0..3, key bytes 0..4 -- the last one is zero, a copy pass that makes the number of passes odd).
__device__ __forceinline__ uint32_t pair_digit(uint64_t k, uint32_t v, int byte) {
  return byte < 4 ? (v >> (8 * byte)) & 255u : (uint32_t)(k >> (8 * (byte - 4))) & 255u;
}
__device__ __forceinline__ void wg_radix_pass(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout,
                                              uint32_t n, int byte, uint32_t *s_run, uint32_t (*s_cnt)[256]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  s_run[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += SORT_BLOCK) atomicAdd(&s_run[pair_digit(kin[i], vin[i], byte)], 1u);
  __syncthreads();
  {  // exclusive scan of the 256 digit counts (thread d owns digit d)
    const uint32_t tot = s_run[threadIdx.x];
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_cnt[0][wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_cnt[0][w];
    __syncthreads();
    s_run[threadIdx.x] = off + inc - tot;
  }
  for (uint32_t i0 = 0; i0 < n; i0 += SORT_BLOCK) {  // chunks in order: stable
#pragma unroll
    for (int w = 0; w < SORT_BLOCK / 64; ++w) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = i0 + threadIdx.x;
    const bool valid = i < n;
    const uint64_t k = valid ? kin[i] : 0;
    const uint32_t v = valid ? vin[i] : 0;
    const uint32_t d = pair_digit(k, v, byte);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long bal = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    if (valid && before == 0) s_cnt[wave][d] = (uint32_t)__popcll(peers);
    __syncthreads();
    {  // thread d: first slot of every wave's entries with digit d, then the running offset moves on
      uint32_t run = s_run[threadIdx.x];
#pragma unroll
      for (int w = 0; w < SORT_BLOCK / 64; ++w) {
        const uint32_t cw = s_cnt[w][threadIdx.x];
        s_cnt[w][threadIdx.x] = run;
        run += cw;
      }
      s_run[threadIdx.x] = run;
    }
    __syncthreads();
    if (valid) {
      const uint32_t pos = s_cnt[wave][d] + before;
      kout[pos] = k, vout[pos] = v;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void depth_bin_sort_body(int N, uint64_t *__restrict__ keys_b, uint32_t *__restrict__ vals_b,
                                                    uint64_t *__restrict__ keys_a, uint32_t *__restrict__ vals_a,
                                                    const uint32_t *__restrict__ bk) {
  __shared__ unsigned long long s_k[BIN_CAP];
  __shared__ uint32_t s_start[SUB_BINS + 1], s_cur[SUB_BINS];
  __shared__ uint32_t s_run[256];
  __shared__ uint32_t s_cnt[SORT_BLOCK / 64][256];
  __shared__ uint32_t s_big;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nc = 1u << depth_bins_log2(N);
  if (blockIdx.x >= nc) {
    // ---- slices of oversized bins: every slice counts the bin's sub-bins itself, takes the run of sub-bins whose
    // first entry falls into its share of the bin, and sorts those (<= SLICE_TARGET + SUB_MAX = BIN_CAP entries)
    const uint32_t n_slices = bk[BK_NSLICE];
    for (uint32_t t = blockIdx.x - nc; t < n_slices; t += gridDim.x - nc) {
      const uint32_t code = bk[BK_SLICE + t];
      if (code == 0xffffffffu) continue;
      const uint32_t b = code >> 16, J = (code >> 8) & 255u, j = code & 255u;
      const uint32_t base = bk[BK_BASE + b], n = bk[BK_BASE + b + 1] - base;
      uint32_t klo, sub_shift;
      sub_bin_map(bk, b, klo, sub_shift);
      __syncthreads();  // (the previous slice's LDS has been consumed)
      s_cur[threadIdx.x] = 0u;
      if (threadIdx.x == 0) s_big = 0u;
      __syncthreads();
      for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {
        uint32_t kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = (uint32_t)keys_b[base + min(e0 + u * SORT_BLOCK + threadIdx.x, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (e0 + u * SORT_BLOCK + threadIdx.x < n)
            atomicAdd(&s_cur[depth_bin(kk[u], klo, sub_shift, (uint32_t)SUB_BINS)], 1u);
      }
      __syncthreads();
      {
        const uint32_t cnt = s_cur[threadIdx.x];
        if (cnt > (uint32_t)SUB_MAX) s_big = 1u;
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t tt = __shfl_up(inc, o, 64);
          if (lane >= o) inc += tt;
        }
        if (lane == 63) s_run[wave] = inc;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += s_run[w];
        s_start[threadIdx.x] = off + inc - cnt;
        if (threadIdx.x == SORT_BLOCK - 1) s_start[SUB_BINS] = off + inc;
      }
      __syncthreads();
      if (s_big != 0u) {  // a sub-bin too large to rank: slice 0 sorts the whole bin the slow way
        if (j == 0) {
          for (int byte = 0; byte < 9; ++byte) {
            const bool b2a = (byte & 1) == 0;
            wg_radix_pass((b2a ? keys_b : keys_a) + base, (b2a ? vals_b : vals_a) + base,
                          (b2a ? keys_a : keys_b) + base, (b2a ? vals_a : vals_b) + base, n, byte, s_run, s_cnt);
            __threadfence_block();
            __syncthreads();
          }
        }
        continue;
      }
      // this slice's sub-bins: those whose first entry lies in [j T, (j + 1) T)
      const uint32_t T = (n + J - 1) / J;
      const uint32_t my0 = s_start[threadIdx.x];
      const bool in = my0 >= j * T && my0 < (j + 1) * T && s_start[threadIdx.x + 1] > my0;
      const uint32_t f_lo_key = in ? threadIdx.x : 0xffffu, f_hi_key = in ? threadIdx.x + 1 : 0u;
      uint32_t mn = f_lo_key, mx = f_hi_key;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
      }
      if (lane == 0) s_run[wave] = mn, s_run[4 + wave] = mx;
      __syncthreads();
      const uint32_t f0 = min(min(s_run[0], s_run[1]), min(s_run[2], s_run[3]));
      const uint32_t f1 = max(max(s_run[4], s_run[5]), max(s_run[6], s_run[7]));
      __syncthreads();
      if (f0 >= f1) continue;  // (an empty share)
      const uint32_t first = s_start[f0];
      s_cur[threadIdx.x] = s_start[threadIdx.x] - first;  // LDS cursors of the slice's sub-bins
      __syncthreads();
      for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {
        unsigned long long kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t e = min(e0 + u * SORT_BLOCK + threadIdx.x, n - 1);
          kv[u] = (keys_b[base + e] << 32) | (unsigned long long)vals_b[base + e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t f = depth_bin((uint32_t)(kv[u] >> 32), klo, sub_shift, (uint32_t)SUB_BINS);
          if (e0 + u * SORT_BLOCK + threadIdx.x < n && f >= f0 && f < f1) s_k[atomicAdd(&s_cur[f], 1u)] = kv[u];
        }
      }
      __syncthreads();
      const uint32_t m = s_start[f1] - first;  // entries of the slice (<= BIN_CAP)
      for (uint32_t e = threadIdx.x; e < m; e += SORT_BLOCK) {
        const unsigned long long c = s_k[e];
        const uint32_t f = depth_bin((uint32_t)(c >> 32), klo, sub_shift, (uint32_t)SUB_BINS);
        const uint32_t lo = s_start[f] - first, hi = s_start[f + 1] - first;
        uint32_t r = 0;
        for (uint32_t tt = lo; tt < hi; tt += 8) {
          unsigned long long v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = s_k[min(tt + u, hi - 1)];
#pragma unroll
          for (int u = 0; u < 8; ++u) r += (tt + u < hi && v[u] < c) ? 1u : 0u;
        }
        keys_a[base + first + lo + r] = c >> 32;
        vals_a[base + first + lo + r] = (uint32_t)c;
      }
    }
    return;
  }
  const uint32_t bin = blockIdx.x;
  const uint32_t base = bk[BK_BASE + bin], n = bk[BK_BASE + bin + 1] - base;
  if (n == 0) return;
  if (n > (uint32_t)BIN_CAP && bk[BK_BINJ + bin] != 0u) return;  // cut into slices: the extra workgroups sort it
  uint32_t klo, sub_shift;
  sub_bin_map(bk, bin, klo, sub_shift);
  bool lds = n <= (uint32_t)BIN_CAP;
  // the thread's entries, every load issued before the first use (a load -> LDS-atomic loop was one memory round
  // trip per 256 entries: 42 us for the kernel)
  constexpr int PER = BIN_CAP / SORT_BLOCK;
  unsigned long long mine[PER];
  if (lds) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const uint32_t e = (uint32_t)q * SORT_BLOCK + threadIdx.x, ec = e < n ? e : n - 1;
      mine[q] = (keys_b[base + ec] << 32) | (unsigned long long)vals_b[base + ec];
    }
    s_cur[threadIdx.x] = 0u;  // (SORT_BLOCK == SUB_BINS)
    if (threadIdx.x == 0) s_big = 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((uint32_t)q * SORT_BLOCK + threadIdx.x < n)
        atomicAdd(&s_cur[depth_bin((uint32_t)(mine[q] >> 32), klo, sub_shift, (uint32_t)SUB_BINS)], 1u);
    __syncthreads();
    {  // exclusive scan of the 256 sub-bin sizes (thread f owns sub-bin f)
      const uint32_t cnt = s_cur[threadIdx.x];
      if (cnt > (uint32_t)SUB_MAX) s_big = 1u;
      uint32_t inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) s_run[wave] = inc;
      __syncthreads();
      uint32_t off = 0;
      for (int w = 0; w < wave; ++w) off += s_run[w];
      s_start[threadIdx.x] = off + inc - cnt;
      if (threadIdx.x == SORT_BLOCK - 1) s_start[SUB_BINS] = off + inc;
      __syncthreads();
      s_cur[threadIdx.x] = s_start[threadIdx.x];
    }
    __syncthreads();
    lds = s_big == 0u;  // (workgroup-uniform)
  }
  if (lds) {
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((uint32_t)q * SORT_BLOCK + threadIdx.x < n)
        s_k[atomicAdd(&s_cur[depth_bin((uint32_t)(mine[q] >> 32), klo, sub_shift, (uint32_t)SUB_BINS)], 1u)] = mine[q];
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < n; e += SORT_BLOCK) {
      const unsigned long long c = s_k[e];
      const uint32_t f = depth_bin((uint32_t)(c >> 32), klo, sub_shift, (uint32_t)SUB_BINS);
      const uint32_t lo = s_start[f], hi = s_start[f + 1];
      uint32_t r = 0;
      for (uint32_t t = lo; t < hi; t += 8) {  // eight LDS reads in flight (clamped, masked)
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s_k[min(t + u, hi - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) r += (t + u < hi && v[u] < c) ? 1u : 0u;
      }
      keys_a[base + lo + r] = c >> 32;
      vals_a[base + lo + r] = (uint32_t)c;
    }
    return;
  }
  // nine stable byte passes over (id, key), b -> a -> b ... -> a
  for (int byte = 0; byte < 9; ++byte) {
    const bool b2a = (byte & 1) == 0;
    wg_radix_pass((b2a ? keys_b : keys_a) + base, (b2a ? vals_b : vals_a) + base, (b2a ? keys_a : keys_b) + base,
                  (b2a ? vals_a : vals_b) + base, n, byte, s_run, s_cnt);
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ placement
// From the depth-ordered Gaussian list to the per-tile lists with ORDERED FILTERS, two levels deep, so that every
// output is written as long contiguous runs (a first version placed each instance with an 8-byte scattered write:
// 15 M partial-line L2 misses per step, 220 us):
//   level 1: the image is cut into <= 256 supertiles of SS x SS tiles; the depth-ordered list is cut into segments
//            of 256; a workgroup per segment (a wave per 64 entries) tests its Gaussians against the supertiles (ballot / popcount give the
//            order-preserving ranks) -- a count pass, a scan over the segments, a fill pass -> per-supertile lists
//            of (Gaussian id, depth bits), still in depth order; list starts are aligned to 256 entries so that
//   level 2: every 256-entry window of the level-1 array belongs to one supertile; a workgroup per window filters it
//            against the <= 64 tiles of that supertile -- count, scan over the supertile's windows, tile starts,
//            fill -> the final (key, value) lists.
struct BinGrid {
  int tiles_x, tiles_y, ss_shift, stx, sty, NS;  // supertile edge = 1 << ss_shift tiles, stx * sty = NS supertiles
};
constexpr int SEG = 256;        // list entries per workgroup (both levels): 4 waves x 64
constexpr int MAX_SUPER = 256;  // supertiles (one level-1 thread each)
static_assert(SEG == MAX_SUPER, "level 1: a workgroup has one thread per entry AND per supertile");

// level-1 metadata in the bin workspace (uint32): [0, 256) list length, [256, 512) list start (multiple of SEG),
// [512] number of level-2 windows, [1024, ...) supertile of every window
constexpr int META_LEN = 0, META_START = MAX_SUPER, META_NWIN = 2 * MAX_SUPER, META_WIN = 4 * MAX_SUPER;

// One workgroup (4 waves) per segment of 256 list entries, one wave per 64 of them: the entries of a segment reach a
// supertile list in (wave, lane) order, so every wave counts its own entries per supertile, an exclusive prefix over
// the four waves (one thread per supertile) orders them, and the fill pass adds the wave-local rank.  (A first
// version walked the segment with ONE wave, four rounds of 64 entries chained through the running counts: with only
// N / 256 waves per render the kernel is pure latency, and that chain was four times longer.)
template <bool FILL>
__device__ __forceinline__ void level1_body(int N_all, const uint32_t *__restrict__ n_sorted, BinGrid gi,
                                            const uint32_t *__restrict__ perm,
                                            const uint64_t *__restrict__ nkeys, const uint16_t *__restrict__ rect,
                                            uint32_t *__restrict__ cnt1, const uint32_t *__restrict__ meta,
                                            uint2 *__restrict__ l1list, size_t l1cap) {
  __shared__ uint32_t s_cnt[SEG / 64][MAX_SUPER];  // count pass: per-wave counts; fill pass: per-wave first slots
  const int seg = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t *row = cnt1 + (size_t)seg * MAX_SUPER;
  // this thread's supertile: the segment's first slot in its list (fill pass), requested before anything else
  const int my_s = threadIdx.x;  // SEG == MAX_SUPER == blockDim.x
  const uint32_t my_base = (FILL && my_s < gi.NS) ? meta[META_START + my_s] + row[my_s] : 0u;
#pragma unroll
  for (int w = 0; w < SEG / 64; ++w) s_cnt[w][threadIdx.x] = 0u;
  int nbits = 0;
  while ((1 << nbits) < gi.NS) ++nbits;
  const int N = min(N_all, (int)n_sorted[0]);  // the depth sort holds only the Gaussians that touch a tile
  const int k = seg * SEG + wave * 64 + lane;
  const bool valid = k < N;
  const uint32_t g = valid ? perm[k] : 0u;
  const uint32_t dbits = (FILL && valid) ? (uint32_t)nkeys[k] : 0u;
  const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)g);
  const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
  const bool some = valid && x1 > x0 && y1 > y0;
  const int sx0 = x0 >> gi.ss_shift, sx1 = (x1 - 1) >> gi.ss_shift;
  const int sy0 = y0 >> gi.ss_shift, sy1 = (y1 - 1) >> gi.ss_shift;
  // Fast path (every rectangle of the wave spans at most 2 x 2 supertiles): a Gaussian then has at most ONE
  // supertile of each (column parity, row parity) class, and a supertile belongs to one class -- so four rounds, in
  // each of which the lanes naming the same supertile are grouped by ballots over the id bits and ranked by lane,
  // keep the list order per supertile.  ~50 instructions per round instead of one ballot per supertile (64+).
  const bool fast = __ballot(some && (sx1 - sx0 >= 2 || sy1 - sy0 >= 2)) == 0;
  __syncthreads();  // counters cleared
  int c_sidx[4];
  uint32_t c_rank[4];
  bool c_has[4];
  if (fast) {
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int sx = sx0 + (((cls & 1) ^ sx0) & 1), sy = sy0 + (((cls >> 1) ^ sy0) & 1);
      const bool has = some && sx <= sx1 && sy <= sy1;
      const int sidx = has ? sy * gi.stx + sx : 0;
      unsigned long long peers = __ballot(has);
      for (int bit = 0; bit < nbits; ++bit) {
        const unsigned long long bal = __ballot((sidx >> bit) & 1);
        peers &= ((sidx >> bit) & 1) ? bal : ~bal;
      }
      c_sidx[cls] = sidx, c_has[cls] = has, c_rank[cls] = (uint32_t)__popcll(peers & lt);
      // one class per supertile: the group's first lane stores the group's size (nobody else writes that counter)
      if (has && (peers & lt) == 0) s_cnt[wave][sidx] = (uint32_t)__popcll(peers);
    }
  } else {  // general path: some rectangle of this wave is larger -- one ballot per supertile
    int sx = 0, sy = 0;
    for (int sidx = 0; sidx < gi.NS; ++sidx) {
      const bool cov = some && sx >= sx0 && sx <= sx1 && sy >= sy0 && sy <= sy1;
      const unsigned long long bal = __ballot(cov);
      if (++sx == gi.stx) sx = 0, ++sy;
      if (bal != 0 && lane == 0) s_cnt[wave][sidx] = (uint32_t)__popcll(bal);
    }
  }
  __syncthreads();
  {  // thread = supertile: the segment's count (count pass) / each wave's first slot (fill pass)
    uint32_t run = my_base;
#pragma unroll
    for (int w = 0; w < SEG / 64; ++w) {
      const uint32_t c = s_cnt[w][threadIdx.x];
      if (FILL) s_cnt[w][threadIdx.x] = run;
      run += c;
    }
    if (!FILL) row[threadIdx.x] = run;
  }
  if (!FILL) return;
  __syncthreads();
  if (fast) {
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const size_t pos = (size_t)s_cnt[wave][c_sidx[cls]] + c_rank[cls];
      if (c_has[cls] && pos < l1cap) l1list[pos] = make_uint2(g, dbits);
    }
  } else {
    int sx = 0, sy = 0;
    for (int sidx = 0; sidx < gi.NS; ++sidx) {
      const bool cov = some && sx >= sx0 && sx <= sx1 && sy >= sy0 && sy <= sy1;
      const unsigned long long bal = __ballot(cov);
      if (++sx == gi.stx) sx = 0, ++sy;
      if (bal == 0) continue;
      const size_t pos = (size_t)s_cnt[wave][sidx] + (uint32_t)__popcll(bal & lt);
      if (cov && pos < l1cap) l1list[pos] = make_uint2(g, dbits);
    }
  }
}

// one workgroup of 4 x MAX_SUPER threads: thread (part, s) scans its quarter of column s of cnt1 down the segments
// (a local sum first, then the exclusive values in place: four quarters and 16 loads in flight per thread instead
// of one 391-step dependent walk, which was 70 us of pure latency), then the list starts (aligned to SEG) and the
// window -> supertile table
constexpr int L1_PARTS = 4;  // (the workgroup has L1_PARTS * MAX_SUPER = 1024 threads)
__device__ __forceinline__ void level1_scan_body(int nseg, int NS, uint32_t *__restrict__ cnt1,
                                                 uint32_t *__restrict__ meta, size_t max_windows) {
  __shared__ uint32_t s_part[16][MAX_SUPER];
  __shared__ uint32_t s_scan[MAX_SUPER];
  // columns = the supertiles rounded up to a power of two (>= 64), the 1024 threads split every column's segments
  // into 1024 / columns parts: 16 parts at 512^2 (64 supertiles) -- four parts over 256 columns walked 98 segments
  // each, in seven dependent rounds of loads, twice
  int lg = 6;
  while ((1 << lg) < NS) ++lg;
  const int C = 1 << lg, parts = (L1_PARTS * MAX_SUPER) >> lg;
  const int sidx = threadIdx.x & (C - 1), part = threadIdx.x >> lg;
  const int per = (nseg + parts - 1) / parts;
  const int b_lo = min(nseg, part * per), b_hi = min(nseg, b_lo + per);
  uint32_t *p = cnt1 + sidx;
  uint32_t sum = 0;
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = b0 + j < b_hi ? p[(size_t)(b0 + j) * MAX_SUPER] : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += v[j];
  }
  s_part[part][sidx] = sum;
  __syncthreads();
  uint32_t run = 0, len = 0;
  for (int q = 0; q < parts; ++q) {
    run += q < part ? s_part[q][sidx] : 0u;
    len += s_part[q][sidx];
  }
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = b0 + j < b_hi ? p[(size_t)(b0 + j) * MAX_SUPER] : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (b0 + j < b_hi) p[(size_t)(b0 + j) * MAX_SUPER] = run;
      run += v[j];
    }
  }
  if (sidx >= NS) len = 0;
  const uint32_t padded = (len + SEG - 1) / SEG * SEG;
  if (part == 0) s_scan[sidx] = padded;
  __syncthreads();
  for (int o = 1; o < C; o <<= 1) {  // Hillis-Steele inclusive scan of the padded lengths
    const uint32_t add = (part == 0 && sidx >= o) ? s_scan[sidx - o] : 0u;
    __syncthreads();
    if (part == 0) s_scan[sidx] += add;
    __syncthreads();
  }
  if (part == 0) {
    const uint32_t start = s_scan[sidx] - padded;
    meta[META_LEN + sidx] = len;
    meta[META_START + sidx] = start;
    if (sidx == C - 1) meta[META_NWIN] = (uint32_t)min((size_t)(s_scan[sidx] / SEG), max_windows);
  }
  // window -> supertile table, by ALL threads: window w belongs to the first supertile whose inclusive end (in
  // windows) lies beyond it -- a binary search over the <= 256 ends in LDS.  (One thread per supertile walking its
  // own windows was a serial chain of a list's length / 256 stores: 6 us of this kernel at R = 1e6, 35 us at 3e6.)
  const uint32_t n_win = (uint32_t)min((size_t)(s_scan[C - 1] / SEG), max_windows);
  for (uint32_t w = threadIdx.x; w < n_win; w += L1_PARTS * MAX_SUPER) {
    int lo = 0, hi = C - 1;  // smallest s with s_scan[s] / SEG > w
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_scan[mid] / SEG > w) hi = mid; else lo = mid + 1;
    }
    meta[META_WIN + w] = (uint32_t)lo;
  }
}

// A workgroup (4 waves) per 256-entry window, one wave per 64 entries (same reasoning as level 1: few windows per
// render, so the per-window chain is the kernel's duration).  Lane j of every wave owns tile j of the supertile.
// Count pass: each wave's per-tile counts -> cnt2w[window][wave][tile], and the window's totals -> cnt2[window][tile]
// (level2_scan turns those into the window's first slot per tile).  Fill pass: a wave starts at the window's first
// slot plus the counts of the waves before it, and ranks its own entries with one ballot per tile.
template <bool FILL>
__device__ __forceinline__ void level2_body(BinGrid gi, uint32_t R_cap, const uint32_t *__restrict__ meta,
                                            const uint2 *__restrict__ l1list, size_t l1cap,
                                            const uint16_t *__restrict__ rect, uint32_t *__restrict__ cnt2,
                                            uint32_t *__restrict__ cnt2w, const uint32_t *__restrict__ tstart,
                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                            uint8_t *__restrict__ grad_flags) {
  __shared__ uint32_t s_c[SEG / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint32_t n_win = meta[META_NWIN];
  const int ss = 1 << gi.ss_shift, ntile = ss * ss;
  for (uint32_t w = blockIdx.x; w < n_win; w += gridDim.x) {
    const int sidx = (int)meta[META_WIN + w];
    const int tx0 = (sidx % gi.stx) << gi.ss_shift, ty0 = (sidx / gi.stx) << gi.ss_shift;
    const size_t pend = min((size_t)meta[META_START + sidx] + meta[META_LEN + sidx], l1cap);
    // lane j owns tile j of the supertile: its count (count pass) or its next free slot (fill pass)
    const int my_tx = tx0 + (lane & (ss - 1)), my_ty = ty0 + (lane >> gi.ss_shift);
    const bool my_in = lane < ntile && my_tx < gi.tiles_x && my_ty < gi.tiles_y;
    uint32_t c = 0;
    if (FILL && my_in) {
      c = tstart[my_ty * gi.tiles_x + my_tx] + cnt2[(size_t)w * 64 + lane];
      for (int v = 0; v < wave; ++v) c += cnt2w[((size_t)w * (SEG / 64) + v) * 64 + lane];
    }
    const size_t p = (size_t)w * SEG + wave * 64 + lane;
    const bool valid = p < pend;
    uint2 en = l1list[valid ? p : 0];
    if (!valid) en = make_uint2(0u, 0u);  // an unwritten slot may hold anything
    const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)en.x);
    const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
    if (__ballot(valid) != 0) {
      for (int j = 0; j < ntile; ++j) {
        const int tx = tx0 + (j & (ss - 1)), ty = ty0 + (j >> gi.ss_shift);
        if (tx >= gi.tiles_x || ty >= gi.tiles_y) continue;
        const bool cov = valid && tx >= x0 && tx < x1 && ty >= y0 && ty < y1;
        const unsigned long long bal = __ballot(cov);
        if (bal == 0) continue;
        if (FILL) {
          const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)c, j) + (uint32_t)__popcll(bal & lt);
          if (cov && pos < R_cap) {
            keys[pos] = ((uint64_t)(uint32_t)(ty * gi.tiles_x + tx) << 32) | en.y;
            vals[pos] = en.x;
            // the fill pass touches every instance slot [0, R) exactly once: it also clears the backward's "gradient
            // record written" flags (indexed by emission position, the same range) -- a launch of its own before
            if (grad_flags) grad_flags[pos] = 0;
          }
        } else {
          c += lane == j ? (uint32_t)__popcll(bal) : 0u;
        }
      }
    }
    if (!FILL) {
      cnt2w[((size_t)w * (SEG / 64) + wave) * 64 + lane] = c;
      __syncthreads();  // (the previous window's sums have been read: see the barrier below)
      s_c[wave][lane] = c;
      __syncthreads();
      if (wave == 0) cnt2[(size_t)w * 64 + lane] = s_c[0][lane] + s_c[1][lane] + s_c[2][lane] + s_c[3][lane];
    }
  }
}

// one wave per supertile, lane j = tile j of it: exclusive scan of cnt2[window][j] down the supertile's windows
// (in place), tile totals
__device__ __forceinline__ void level2_scan_body(BinGrid gi, const uint32_t *__restrict__ meta,
                                                 uint32_t *__restrict__ cnt2, uint32_t *__restrict__ totals) {
  const int sidx = blockIdx.x, lane = threadIdx.x;
  const int ss = 1 << gi.ss_shift;
  const int tx = ((sidx % gi.stx) << gi.ss_shift) + (lane & (ss - 1));
  const int ty = ((sidx / gi.stx) << gi.ss_shift) + (lane >> gi.ss_shift);
  const uint32_t n_win = meta[META_NWIN];
  const uint32_t w0 = meta[META_START + sidx] / SEG;
  const uint32_t w1 = min(w0 + (meta[META_LEN + sidx] + SEG - 1) / SEG, n_win);
  uint32_t run = 0;
  for (uint32_t w = w0; w < w1; w += 16) {  // sixteen windows' counts in flight (one at a time: 13 us of latency)
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = w + j < w1 ? cnt2[(size_t)(w + j) * 64 + lane] : 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (w + j < w1) cnt2[(size_t)(w + j) * 64 + lane] = run;
      run += v[j];
    }
  }
  if (lane < ss * ss && tx < gi.tiles_x && ty < gi.tiles_y) totals[ty * gi.tiles_x + tx] = run;
}

// one workgroup: tile starts = exclusive scan of the tile totals (left in totals[]), tile ranges clamped to the
// instance capacity, overflow flag, and the backward's three work counters cleared for the blend forward behind it
__device__ __forceinline__ void tile_starts_body(int T, uint32_t R_cap, uint32_t *__restrict__ totals,
                                                 uint32_t *__restrict__ ranges, uint32_t *__restrict__ total,
                                                 uint32_t *__restrict__ work_count) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0, work_count[0] = 0, work_count[1] = 0, work_count[2] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < T; base += 1024) {
    const int t = base + threadIdx.x;
    const uint32_t v = t < T ? totals[t] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (t < T) {
      const uint32_t start = off + inc - v;
      totals[t] = start;
      // an empty tile reports (0, 0) like the published identifyTileRanges leaves it
      ranges[2 * t] = v ? min(start, R_cap) : 0u, ranges[2 * t + 1] = v ? min(start + v, R_cap) : 0u;
      if (start + v > R_cap) total[1] = 1;  // capacity overflow: flagged, never written out of bounds
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + inc;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ kernel entry points
// Every stage exists as a single-render kernel (the C-ABI calls) and as a batched one whose blockIdx.y selects the
// render of a RenderBatch (the native step executor: one launch per stage for all renders of a range).
struct BinPtrs {  // byte offsets into the geometry (g_) and bin (b_) workspaces
  size_t g_total, g_rect, g_perm, g_nkeys, g_cnt1;
  size_t b_meta, b_l1, b_cnt2, b_totals, b_ranges, b_work, b_keys, b_vals;
  size_t l1cap, max_windows;
};

__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nb, int N, uint32_t *__restrict__ sums,
                                                               uint32_t *__restrict__ total,
                                                               uint32_t *__restrict__ bk) {
  scan_block_sums_body(nb, N, sums, total, bk);
}
__global__ void __launch_bounds__(PRE_BLOCK) write_offsets_kernel(int N, const uint32_t *__restrict__ tiles,
                                                                  const uint32_t *__restrict__ block_prefix,
                                                                  uint32_t *__restrict__ offsets,
                                                                  const Splat *__restrict__ splat,
                                                                  uint64_t *__restrict__ nkeys,
                                                                  uint32_t *__restrict__ nvals,
                                                                  uint32_t *__restrict__ bk) {
  write_offsets_body(N, tiles, block_prefix, offsets, splat, nkeys, nvals, bk);
}
__global__ void __launch_bounds__(1024) depth_bin_scan_kernel(int N, GeomLayout L, void *geom) {
  depth_bin_scan_body(N, at<uint32_t>(geom, L.bk), at<uint32_t>(geom, L.total));
}
__global__ void __launch_bounds__(PRE_BLOCK) depth_bin_scatter_kernel(int N, GeomLayout L, void *geom) {
  depth_bin_scatter_body(N, at<uint64_t>(geom, L.nkeys_a), at<uint64_t>(geom, L.nkeys_b), at<uint32_t>(geom, L.nvals_b),
                         at<uint32_t>(geom, L.bk));
}
__global__ void __launch_bounds__(SORT_BLOCK) depth_bin_sort_kernel(int N, GeomLayout L, void *geom) {
  depth_bin_sort_body(N, at<uint64_t>(geom, L.nkeys_b), at<uint32_t>(geom, L.nvals_b), at<uint64_t>(geom, L.nkeys_a),
                      at<uint32_t>(geom, L.nvals_a), at<uint32_t>(geom, L.bk));
}

// placement stages: `geom` / `bin` are the workspaces of the render (blockIdx.y picks it in the batched launches)
template <bool FILL>
__device__ __forceinline__ void level1_stage(int N, BinGrid gi, const BinPtrs &o, void *geom, void *bin) {
  level1_body<FILL>(N, at<uint32_t>(geom, o.g_total) + 3, gi, at<uint32_t>(geom, o.g_perm),
                    at<uint64_t>(geom, o.g_nkeys), at<uint16_t>(geom, o.g_rect),
                    at<uint32_t>(geom, o.g_cnt1), at<uint32_t>(bin, o.b_meta), at<uint2>(bin, o.b_l1), o.l1cap);
}
template <bool FILL>
__device__ __forceinline__ void level2_stage(BinGrid gi, uint32_t R_cap, const BinPtrs &o, void *geom, void *bin,
                                             uint8_t *grad_flags) {
  level2_body<FILL>(gi, R_cap, at<uint32_t>(bin, o.b_meta), at<uint2>(bin, o.b_l1), o.l1cap,
                    at<uint16_t>(geom, o.g_rect), at<uint32_t>(bin, o.b_cnt2),
                    at<uint32_t>(bin, o.b_cnt2) + o.max_windows * 64, at<uint32_t>(bin, o.b_totals),
                    at<uint64_t>(bin, o.b_keys), at<uint32_t>(bin, o.b_vals), grad_flags);
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level1_kernel(int N, BinGrid gi, BinPtrs o, void *geom, void *bin) {
  level1_stage<FILL>(N, gi, o, geom, bin);
}
__global__ void __launch_bounds__(L1_PARTS *MAX_SUPER) level1_scan_kernel(int nseg, int NS, BinPtrs o, void *geom, void *bin) {
  level1_scan_body(nseg, NS, at<uint32_t>(geom, o.g_cnt1), at<uint32_t>(bin, o.b_meta), o.max_windows);
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level2_kernel(BinGrid gi, uint32_t R_cap, BinPtrs o, void *geom, void *bin) {
  level2_stage<FILL>(gi, R_cap, o, geom, bin, nullptr);  // (the C-ABI backward clears its own scratch)
}
__global__ void __launch_bounds__(64) level2_scan_kernel(BinGrid gi, BinPtrs o, void *bin) {
  level2_scan_body(gi, at<uint32_t>(bin, o.b_meta), at<uint32_t>(bin, o.b_cnt2), at<uint32_t>(bin, o.b_totals));
}
__global__ void __launch_bounds__(1024) tile_starts_kernel(int T, uint32_t R_cap, BinPtrs o, void *geom, void *bin) {
  tile_starts_body(T, R_cap, at<uint32_t>(bin, o.b_totals), at<uint32_t>(bin, o.b_ranges),
                   at<uint32_t>(geom, o.g_total), at<uint32_t>(bin, o.b_work));
}

__global__ void __launch_bounds__(1024) scan_block_sums_batched_kernel(int nb, int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  scan_block_sums_body(nb, N, at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.total), at<uint32_t>(geom, L.bk));
}
__global__ void __launch_bounds__(PRE_BLOCK) write_offsets_batched_kernel(int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  write_offsets_body(N, at<uint32_t>(geom, L.tiles), at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.offsets),
                     at<Splat>(geom, L.splat), at<uint64_t>(geom, L.nkeys_a), at<uint32_t>(geom, L.nvals_a),
                     at<uint32_t>(geom, L.bk));
}
__global__ void __launch_bounds__(1024) depth_bin_scan_batched_kernel(int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  depth_bin_scan_body(N, at<uint32_t>(geom, L.bk), at<uint32_t>(geom, L.total));
}
__global__ void __launch_bounds__(PRE_BLOCK) depth_bin_scatter_batched_kernel(int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  depth_bin_scatter_body(N, at<uint64_t>(geom, L.nkeys_a), at<uint64_t>(geom, L.nkeys_b), at<uint32_t>(geom, L.nvals_b),
                         at<uint32_t>(geom, L.bk));
}
__global__ void __launch_bounds__(SORT_BLOCK) depth_bin_sort_batched_kernel(int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  depth_bin_sort_body(N, at<uint64_t>(geom, L.nkeys_b), at<uint32_t>(geom, L.nvals_b), at<uint64_t>(geom, L.nkeys_a),
                      at<uint32_t>(geom, L.nvals_a), at<uint32_t>(geom, L.bk));
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level1_batched_kernel(int N, BinGrid gi, BinPtrs o, RenderBatch b) {
  level1_stage<FILL>(N, gi, o, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
__global__ void __launch_bounds__(L1_PARTS *MAX_SUPER) level1_scan_batched_kernel(int nseg, int NS, BinPtrs o, RenderBatch b) {
  level1_scan_body(nseg, NS, at<uint32_t>(b.r[blockIdx.y].geom, o.g_cnt1), at<uint32_t>(b.r[blockIdx.y].bin, o.b_meta),
                   o.max_windows);
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level2_batched_kernel(BinGrid gi, uint32_t R_cap, BinPtrs o, size_t flag_off,
                                                             RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  level2_stage<FILL>(gi, R_cap, o, r.geom, r.bin, FILL ? at<uint8_t>(r.bwd_scratch, flag_off) : nullptr);
}
__global__ void __launch_bounds__(64) level2_scan_batched_kernel(BinGrid gi, BinPtrs o, RenderBatch b) {
  void *bin = b.r[blockIdx.y].bin;
  level2_scan_body(gi, at<uint32_t>(bin, o.b_meta), at<uint32_t>(bin, o.b_cnt2), at<uint32_t>(bin, o.b_totals));
}
__global__ void __launch_bounds__(1024) tile_starts_batched_kernel(int T, uint32_t R_cap, BinPtrs o, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  tile_starts_body(T, R_cap, at<uint32_t>(r.bin, o.b_totals), at<uint32_t>(r.bin, o.b_ranges),
                   at<uint32_t>(r.geom, o.g_total), at<uint32_t>(r.bin, o.b_work));
}

// ------------------------------------------------------------------------------------ host side
constexpr int L2_GRID = 2048;  // persistent waves over the level-2 windows (their number is only known on device)

// supertile edge: the smallest power of two that leaves <= 64 supertiles, else <= MAX_SUPER; edge <= 8 (64 tiles
// per supertile = one lane each at level 2)
static bool make_grid(const BinLayout &B, BinGrid &gi) {
  gi.tiles_x = B.tiles_x, gi.tiles_y = B.tiles_y;
  for (int limit : {64, MAX_SUPER})
    for (int sh = 0; sh <= 3; ++sh) {
      const int ss = 1 << sh;
      const int stx = (B.tiles_x + ss - 1) / ss, sty = (B.tiles_y + ss - 1) / ss;
      if (stx * sty <= limit) {
        gi.ss_shift = sh, gi.stx = stx, gi.sty = sty, gi.NS = stx * sty;
        return true;
      }
    }
  return false;
}

static BinPtrs make_ptrs(const GeomLayout &G, const BinLayout &B) {
  BinPtrs o;
  o.g_total = G.total, o.g_rect = G.rect, o.g_perm = G.nvals_a, o.g_nkeys = G.nkeys_a, o.g_cnt1 = G.cnt1;
  o.b_meta = B.meta, o.b_l1 = B.l1list, o.b_cnt2 = B.cnt2, o.b_totals = B.totals, o.b_ranges = B.ranges;
  o.b_work = B.work, o.b_keys = B.keys_b, o.b_vals = B.vals_b;
  o.l1cap = B.l1cap, o.max_windows = B.max_windows;
  return o;
}

int scan_block_sums(int nb, int N, uint32_t *block_sums, uint32_t *total, uint32_t *bk, hipStream_t stream) {
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, nb, N, block_sums, total, bk);
  return check_launch();
}

int write_offsets(int N, const void *geom_c, hipStream_t stream) {
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb == 0) return DIMO_OK;
  GeomLayout L(N);
  void *geom = const_cast<void *>(geom_c);
  hipLaunchKernelGGL(write_offsets_kernel, dim3(nb), dim3(PRE_BLOCK), 0, stream, N, at<uint32_t>(geom, L.tiles),
                     at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.offsets), at<Splat>(geom, L.splat),
                     at<uint64_t>(geom, L.nkeys_a), at<uint32_t>(geom, L.nvals_a), at<uint32_t>(geom, L.bk));
  return check_launch();
}

int bin_instances(int N, int H, int W, int64_t R_cap, const void *geom_c, void *bin, hipStream_t stream) {
  GeomLayout G(N);
  BinLayout B(R_cap, H, W);
  BinGrid gi;
  if (!make_grid(B, gi)) return DIMO_E_ARG;  // more than MAX_SUPER * 64 tiles
  void *geom = const_cast<void *>(geom_c);  // sort scratch and the overflow flag live in the geometry workspace
  const uint32_t cap = (uint32_t)B.cap;
  const BinPtrs o = make_ptrs(G, B);
  if (N > 0) {
    ScopedTimer tm(T_SORT, stream);
    const unsigned nb = (unsigned)((N + PRE_BLOCK - 1) / PRE_BLOCK), nc = 1u << depth_bins_log2(N);
    hipLaunchKernelGGL(depth_bin_scan_kernel, dim3(1), dim3(1024), 0, stream, N, G, geom);
    hipLaunchKernelGGL(depth_bin_scatter_kernel, dim3(nb), dim3(PRE_BLOCK), 0, stream, N, G, geom);
    hipLaunchKernelGGL(depth_bin_sort_kernel, dim3(nc + SLICE_GRID), dim3(SORT_BLOCK), 0, stream, N, G, geom);
  }
  const int nseg = (int)G.nseg1;
  {
    ScopedTimer tm(T_EMIT, stream);
    if (N > 0) hipLaunchKernelGGL(level1_kernel<false>, dim3(nseg), dim3(SEG), 0, stream, N, gi, o, geom, bin);
    hipLaunchKernelGGL(level1_scan_kernel, dim3(1), dim3(L1_PARTS * MAX_SUPER), 0, stream, N > 0 ? nseg : 0, gi.NS, o, geom, bin);
    if (N > 0) hipLaunchKernelGGL(level1_kernel<true>, dim3(nseg), dim3(SEG), 0, stream, N, gi, o, geom, bin);
  }
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(level2_kernel<false>, dim3(L2_GRID), dim3(SEG), 0, stream, gi, cap, o, geom, bin);
    hipLaunchKernelGGL(level2_scan_kernel, dim3(gi.NS), dim3(64), 0, stream, gi, o, bin);
    hipLaunchKernelGGL(tile_starts_kernel, dim3(1), dim3(1024), 0, stream, B.T, cap, o, geom, bin);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    hipLaunchKernelGGL(level2_kernel<true>, dim3(L2_GRID), dim3(SEG), 0, stream, gi, cap, o, geom, bin);
  }
  return check_launch();
}

int scan_offsets_batched(int N, const GeomLayout &L, const RenderBatch &b, int n, hipStream_t stream) {
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  hipLaunchKernelGGL(scan_block_sums_batched_kernel, dim3(1, n), dim3(1024), 0, stream, nb, N, L, b);
  if (nb > 0) hipLaunchKernelGGL(write_offsets_batched_kernel, dim3(nb, n), dim3(PRE_BLOCK), 0, stream, N, L, b);
  return check_launch();
}

int bin_instances_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W);
  BinGrid gi;
  if (!make_grid(B, gi)) return DIMO_E_ARG;
  if (c.bin_bytes < B.bytes || c.geom_bytes < G.bytes) return DIMO_E_WORKSPACE;
  if (c.bwd_scratch_bytes < align_up(B.cap * sizeof(SplatGrad)) + align_up(B.cap)) return DIMO_E_WORKSPACE;
  for (int i = 0; i < n; ++i)
    if (!b.r[i].bwd_scratch) return DIMO_E_ARG;  // the fill pass clears the backward's record flags
  const uint32_t cap = (uint32_t)B.cap;
  const BinPtrs o = make_ptrs(G, B);
  if (c.N > 0) {
    ScopedTimer tm(T_SORT, stream);
    const unsigned nb = (unsigned)((c.N + PRE_BLOCK - 1) / PRE_BLOCK), nc = 1u << depth_bins_log2(c.N);
    hipLaunchKernelGGL(depth_bin_scan_batched_kernel, dim3(1, n), dim3(1024), 0, stream, c.N, G, b);
    hipLaunchKernelGGL(depth_bin_scatter_batched_kernel, dim3(nb, n), dim3(PRE_BLOCK), 0, stream, c.N, G, b);
    hipLaunchKernelGGL(depth_bin_sort_batched_kernel, dim3(nc + SLICE_GRID, n), dim3(SORT_BLOCK), 0, stream, c.N, G, b);
  }
  const int nseg = (int)G.nseg1;
  {
    ScopedTimer tm(T_EMIT, stream);
    if (c.N > 0) hipLaunchKernelGGL(level1_batched_kernel<false>, dim3(nseg, n), dim3(SEG), 0, stream, c.N, gi, o, b);
    hipLaunchKernelGGL(level1_scan_batched_kernel, dim3(1, n), dim3(L1_PARTS * MAX_SUPER), 0, stream, c.N > 0 ? nseg : 0, gi.NS,
                       o, b);
    if (c.N > 0) hipLaunchKernelGGL(level1_batched_kernel<true>, dim3(nseg, n), dim3(SEG), 0, stream, c.N, gi, o, b);
  }
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(level2_batched_kernel<false>, dim3(L2_GRID, n), dim3(SEG), 0, stream, gi, cap, o, (size_t)0, b);
    hipLaunchKernelGGL(level2_scan_batched_kernel, dim3(gi.NS, n), dim3(64), 0, stream, gi, o, b);
    hipLaunchKernelGGL(tile_starts_batched_kernel, dim3(1, n), dim3(1024), 0, stream, B.T, cap, o, b);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    // (flags of the backward's scratch: [records: cap x 64 B][flags: cap x 1 B], see blend.hip)
    hipLaunchKernelGGL(level2_batched_kernel<true>, dim3(L2_GRID, n), dim3(SEG), 0, stream, gi, cap, o,
                       align_up(B.cap * sizeof(SplatGrad)), b);
  }
  return check_launch();
}

}  // namespace dimo
