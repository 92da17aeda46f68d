// Tile binning: scan of tiles_touched, key emission, sort of (tile | fp32 depth bits) keys with the Gaussian id
// as payload, and tile ranges.  The sorted arrays equal a stable LSD radix sort over the whole key (the published
// rasterizer's cub::DeviceRadixSort), but are produced in two steps that fit the machine better: stable radix
// passes over the TILE bits only (2 passes at 512^2 instead of 6 over 42 bits -- every pass is three launches
// whose cost is latency, not bandwidth: the instance arrays live in L2), then one workgroup per tile orders its
// segment by (depth bits, Gaussian id) with a bitonic network in LDS.  The radix passes keep the emission order
// (ascending Gaussian id) inside a tile, so (depth, id) is exactly the stable order.
//
// All kernels read the live instance count R from device memory (geom.total[0]) and are launched
// over the CAPACITY R_cap, so the whole chain is enqueued without a host round trip.
#include "common.hpp"

namespace dimo {

// ------------------------------------------------------------------------------------ scan
// Exclusive scan of the per-block sums (nb <= a few thousand) by one workgroup; writes the
// grand total R to total[0] and clears the overflow flag total[1].
__device__ __forceinline__ void scan_block_sums_body(int nb, uint32_t *__restrict__ sums,
                                                     uint32_t *__restrict__ total) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
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
    total[3] = 0;
  }
}

// offsets[i] = inclusive scan of tiles_touched (block prefix + in-block scan)
__device__ __forceinline__ void write_offsets_body(int N, const uint32_t *__restrict__ tiles,
                                                   const uint32_t *__restrict__ block_prefix,
                                                   uint32_t *__restrict__ offsets) {
  __shared__ uint32_t wave_tot[PRE_BLOCK / 64];
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
  if (i < N) offsets[i] = off + inc;
}

// ------------------------------------------------------------------------------------ emission
// One thread per Gaussian: writes its (key, id) run at [offsets[i-1], offsets[i]).
// key = (tile_id << 32) | depth bits; emission order = tile y, then tile x.
__device__ __forceinline__ void emit_keys_body(int N, int tiles_x, uint32_t R_cap, const Splat *__restrict__ splat,
                                               const uint16_t *__restrict__ rect,
                                               const uint32_t *__restrict__ offsets, uint32_t *__restrict__ total,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const uint32_t hi = offsets[i];
  uint32_t off = i == 0 ? 0u : offsets[i - 1];
  if (hi == off) return;
  if (hi > R_cap) {  // capacity overflow: flag it, never write out of bounds
    total[1] = 1;
    if (off >= R_cap) return;
  }
  const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i);
  const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
  const uint32_t dbits = __float_as_uint(splat[i].depth);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      if (off < R_cap) {
        keys[off] = ((uint64_t)(uint32_t)(y * tiles_x + x) << 32) | dbits;
        vals[off] = (uint32_t)i;
      }
      ++off;
    }
}

// ------------------------------------------------------------------------------------ radix sort
// LSD, 8 bits per pass, stable.  Per pass: (1) per-block digit histogram, (2) exclusive scan over
// (digit-major, block-minor) counts, (3) stable scatter using wave-level match ranking.
// Key i of a sort block lives at (wave w, item j, lane l) -> index ((w*ITEMS + j)*64 + l): order
// inside the block is (wave, item, lane), which the ranking below preserves.
constexpr int SORT_WAVES = SORT_BLOCK / 64;

__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift) { return (uint32_t)(k >> shift) & (RADIX - 1); }

__device__ __forceinline__ void radix_hist_body(const uint64_t *__restrict__ keys,
                                                const uint32_t *__restrict__ total, uint32_t R_cap, int shift,
                                                uint32_t num_blocks, uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[RADIX];
  const uint32_t R = min(total[0], R_cap);
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * SORT_TILE;
  if (base < R) {
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
      const uint32_t idx = base + j * SORT_BLOCK + threadIdx.x;
      if (idx < R) atomicAdd(&h[digit_of(keys[idx], shift)], 1u);
    }
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * num_blocks + blockIdx.x] = h[threadIdx.x];
}

// Row scan: one wave per digit turns hist[d][0..num_blocks) into its exclusive prefix and stores
// the digit total at hist[RADIX*num_blocks + d].  256 independent waves -> no single-block tail.
__device__ __forceinline__ void radix_rowscan_body(uint32_t num_blocks, uint32_t *__restrict__ hist) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t *row = hist + (size_t)d * num_blocks;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < num_blocks; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t v = i < num_blocks ? row[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (i < num_blocks) row[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) hist[(size_t)RADIX * num_blocks + d] = carry;
}

__device__ __forceinline__ void radix_scatter_body(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ total, uint32_t R_cap, int shift,
    uint32_t num_blocks, const uint32_t *__restrict__ hist) {
  __shared__ uint32_t cnt[SORT_WAVES][RADIX];  // per-wave running digit counts
  __shared__ uint32_t gbase[RADIX];            // global base of (digit, this block)
  const uint32_t R = min(total[0], R_cap);
  const uint32_t base = blockIdx.x * SORT_TILE;
  if (base >= R) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < SORT_WAVES; ++w) cnt[w][threadIdx.x] = 0;
  {
    // exclusive scan of the 256 digit totals (thread d owns digit d) + this block's row prefix
    __shared__ uint32_t wtot[SORT_WAVES];
    const uint32_t tot = hist[(size_t)RADIX * num_blocks + threadIdx.x];
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    gbase[threadIdx.x] = off + inc - tot + hist[(size_t)threadIdx.x * num_blocks + blockIdx.x];
  }
  __syncthreads();

  uint64_t k[SORT_ITEMS];
  uint32_t v[SORT_ITEMS], rank[SORT_ITEMS];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const uint32_t idx = base + (wave * SORT_ITEMS + j) * 64 + lane;
    const bool valid = idx < R;
    k[j] = valid ? keys_in[idx] : ~0ull;
    v[j] = valid ? vals_in[idx] : 0u;
    const uint32_t d = digit_of(k[j], shift);
    // lanes of this wave holding the same digit (invalid lanes excluded)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    uint32_t prev = 0;
    if (valid && before == 0) {  // group leader bumps the wave's running count
      prev = cnt[wave][d];
      cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    }
    const int leader = __ffsll((long long)peers) - 1;
    prev = __shfl(prev, leader < 0 ? 0 : leader, 64);
    rank[j] = prev + before;
  }
  __syncthreads();
  // exclusive prefix over waves per digit, folded into the global base
  {
    uint32_t run = gbase[threadIdx.x];
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const uint32_t c = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SORT_ITEMS; ++j) {
    const uint32_t idx = base + (wave * SORT_ITEMS + j) * 64 + lane;
    if (idx < R) {
      const uint32_t pos = cnt[wave][digit_of(k[j], shift)] + rank[j];
      keys_out[pos] = k[j];
      vals_out[pos] = v[j];
    }
  }
}

// ------------------------------------------------------------------------------------ ranges
__device__ __forceinline__ void clear_ranges_body(int T, uint32_t *__restrict__ ranges,
                                                  uint32_t *__restrict__ work_count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * T) ranges[i] = 0;
  if (i == 0) *work_count = 0;  // the blend forward queues the backward's (tile, bucket) items behind it
}
__device__ __forceinline__ void tile_ranges_body(const uint64_t *__restrict__ keys,
                                                 const uint32_t *__restrict__ total, uint32_t R_cap,
                                                 uint32_t *__restrict__ ranges) {
  const uint32_t R = min(total[0], R_cap);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t tile = (uint32_t)(keys[i] >> 32);
  if (i == 0)
    ranges[2 * tile] = 0;
  else {
    const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
    if (prev != tile) {
      ranges[2 * prev + 1] = i;
      ranges[2 * tile] = i;
    }
  }
  if (i == R - 1) ranges[2 * tile + 1] = R;
}

// ------------------------------------------------------------------------------------ per-tile depth sort
// One workgroup per tile orders the tile's segment by the 32 depth bits (stable, so ties keep the ascending
// Gaussian id the radix passes over the tile bits left them in).  Segments of up to TS_CAP instances -- all of
// them in practice -- are sorted in LDS with the same wave-ballot counting scatter as the global passes, 8 bits
// at a time, skipping digits every key of the tile agrees on (view-space depths of one tile share their exponent
// byte).  Linear in the segment length: the few crowded tiles do not leave a long tail behind.
constexpr int TS_CAP = 4096;
constexpr int TS_ITEMS = TS_CAP / 256;
constexpr size_t TS_LDS_BYTES = 2 * TS_CAP * sizeof(uint64_t) + 4 * RADIX * sizeof(uint32_t);

// Oversized segments fall back to a bitonic network on the global arrays (slow, exact).  "Mirror" form: every
// compare-exchange moves the smaller element down, so an arbitrary length n behaves as if padded with +inf --
// pairs whose upper index is >= n are skipped.
template <class CmpSwap>
__device__ __forceinline__ void bitonic_network(uint32_t n, CmpSwap cmp_swap) {
  uint32_t np2 = 1;
  while (np2 < n) np2 <<= 1;
  const uint32_t half = np2 >> 1;
  for (uint32_t k = 2; k <= np2; k <<= 1) {
    for (uint32_t p = threadIdx.x; p < half; p += blockDim.x) {  // mirror step
      const uint32_t blk = p / (k >> 1), l = p % (k >> 1);
      const uint32_t lo = blk * k + l, hi = blk * k + k - 1 - l;
      if (hi < n) cmp_swap(lo, hi);
    }
    __syncthreads();
    for (uint32_t j = k >> 2; j > 0; j >>= 1) {
      for (uint32_t p = threadIdx.x; p < half; p += blockDim.x) {
        const uint32_t lo = (p / j) * 2 * j + (p % j), hi = lo + j;
        if (hi < n) cmp_swap(lo, hi);
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ void tile_depth_sort_body(const uint32_t *__restrict__ ranges, uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ vals) {
  extern __shared__ uint64_t ts_smem[];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t diff_s;
  const uint32_t tile = blockIdx.x;
  const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
  if (end <= beg + 1) return;
  const uint32_t n = end - beg;
  if (n > TS_CAP) {
    uint64_t *k = keys + beg;
    uint32_t *v = vals + beg;
    bitonic_network(n, [&](uint32_t lo, uint32_t hi) {
      const uint64_t ka = k[lo], kb = k[hi];
      const uint32_t va = v[lo], vb = v[hi];
      if (kb < ka || (kb == ka && vb < va)) k[lo] = kb, k[hi] = ka, v[lo] = vb, v[hi] = va;
    });
    return;
  }
  uint64_t *src = ts_smem, *dst = ts_smem + TS_CAP;  // composites (depth bits << 32 | Gaussian id)
  uint32_t(*cnt)[RADIX] = reinterpret_cast<uint32_t(*)[RADIX]>(ts_smem + 2 * TS_CAP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) diff_s = 0;
  __syncthreads();
  {
    const uint32_t first = (uint32_t)keys[beg];
    uint32_t diff = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
      const uint32_t d = (uint32_t)keys[beg + i];
      src[i] = ((uint64_t)d << 32) | vals[beg + i];
      diff |= d ^ first;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o, 64);
    if (lane == 0 && diff) atomicOr(&diff_s, diff);
  }
  __syncthreads();
  const uint32_t diff = diff_s;
  const uint32_t chunk = (((n + 3) >> 2) + 63) & ~63u;  // contiguous elements per wave, a multiple of 64
  const int nitems = (int)(chunk >> 6);                 // <= TS_ITEMS
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int pass = 0; pass < 4; ++pass) {
    if (((diff >> (8 * pass)) & 0xffu) == 0) continue;  // every key of the tile has the same digit
    const int shift = 32 + 8 * pass;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    uint64_t k[TS_ITEMS];
    uint32_t rank[TS_ITEMS];
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j < nitems) {
        const uint32_t idx = wave * chunk + j * 64 + lane;
        const bool valid = idx < n;
        k[j] = valid ? src[idx] : ~0ull;
        const uint32_t d = digit_of(k[j], shift);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
          const unsigned long long bal = __ballot((d >> b) & 1u);
          peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        uint32_t prev = 0;
        if (valid && before == 0) {
          prev = cnt[wave][d];
          cnt[wave][d] = prev + (uint32_t)__popcll(peers);
        }
        const int leader = __ffsll((long long)peers) - 1;
        prev = __shfl(prev, leader < 0 ? 0 : leader, 64);
        rank[j] = prev + before;
      }
    }
    __syncthreads();
    {  // thread d: exclusive scan of the digit totals, then the per-wave bases of digit d
      const uint32_t c0 = cnt[0][threadIdx.x], c1 = cnt[1][threadIdx.x], c2 = cnt[2][threadIdx.x],
                     c3 = cnt[3][threadIdx.x];
      const uint32_t tot = c0 + c1 + c2 + c3;
      uint32_t inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      uint32_t run = inc - tot;
      for (int w = 0; w < wave; ++w) run += wtot[w];
      cnt[0][threadIdx.x] = run, run += c0;
      cnt[1][threadIdx.x] = run, run += c1;
      cnt[2][threadIdx.x] = run, run += c2;
      cnt[3][threadIdx.x] = run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TS_ITEMS; ++j) {
      if (j < nitems) {
        const uint32_t idx = wave * chunk + j * 64 + lane;
        if (idx < n) dst[cnt[wave][digit_of(k[j], shift)] + rank[j]] = k[j];
      }
    }
    __syncthreads();
    uint64_t *t = src;
    src = dst, dst = t;
  }
  const uint64_t tbits = (uint64_t)tile << 32;
  for (uint32_t i = threadIdx.x; i < n; i += 256) {
    const uint64_t c = src[i];
    keys[beg + i] = tbits | (c >> 32);
    vals[beg + i] = (uint32_t)c;
  }
}

// ------------------------------------------------------------------------------------ kernel entry points
// Every stage exists as a single-render kernel (the C-ABI calls) and as a batched one whose blockIdx.y selects the
// render of a RenderBatch (the native step executor: one launch per stage for all renders of a step).
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nb, uint32_t *__restrict__ sums,
                                                               uint32_t *__restrict__ total) {
  scan_block_sums_body(nb, sums, total);
}
__global__ void __launch_bounds__(PRE_BLOCK) write_offsets_kernel(int N, const uint32_t *__restrict__ tiles,
                                                                  const uint32_t *__restrict__ block_prefix,
                                                                  uint32_t *__restrict__ offsets) {
  write_offsets_body(N, tiles, block_prefix, offsets);
}
__global__ void __launch_bounds__(256) emit_keys_kernel(int N, int tiles_x, uint32_t R_cap,
                                                        const Splat *__restrict__ splat,
                                                        const uint16_t *__restrict__ rect,
                                                        const uint32_t *__restrict__ offsets,
                                                        uint32_t *__restrict__ total, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
  emit_keys_body(N, tiles_x, R_cap, splat, rect, offsets, total, keys, vals);
}
__global__ void __launch_bounds__(SORT_BLOCK) radix_hist_kernel(const uint64_t *__restrict__ keys,
                                                                const uint32_t *__restrict__ total, uint32_t R_cap,
                                                                int shift, uint32_t num_blocks,
                                                                uint32_t *__restrict__ hist) {
  radix_hist_body(keys, total, R_cap, shift, num_blocks, hist);
}
__global__ void __launch_bounds__(256) radix_rowscan_kernel(uint32_t num_blocks, uint32_t *__restrict__ hist) {
  radix_rowscan_body(num_blocks, hist);
}
__global__ void __launch_bounds__(SORT_BLOCK) radix_scatter_kernel(
    const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ total, uint32_t R_cap, int shift,
    uint32_t num_blocks, const uint32_t *__restrict__ hist) {
  radix_scatter_body(keys_in, vals_in, keys_out, vals_out, total, R_cap, shift, num_blocks, hist);
}
__global__ void __launch_bounds__(256) clear_ranges_kernel(int T, uint32_t *__restrict__ ranges,
                                                           uint32_t *__restrict__ work_count) {
  clear_ranges_body(T, ranges, work_count);
}
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint64_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ total, uint32_t R_cap,
                                                          uint32_t *__restrict__ ranges) {
  tile_ranges_body(keys, total, R_cap, ranges);
}
__global__ void __launch_bounds__(256) tile_depth_sort_kernel(const uint32_t *__restrict__ ranges,
                                                              uint64_t *__restrict__ keys,
                                                              uint32_t *__restrict__ vals) {
  tile_depth_sort_body(ranges, keys, vals);
}

__global__ void __launch_bounds__(1024) scan_block_sums_batched_kernel(int nb, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  scan_block_sums_body(nb, at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.total));
}
__global__ void __launch_bounds__(PRE_BLOCK) write_offsets_batched_kernel(int N, GeomLayout L, RenderBatch b) {
  void *geom = b.r[blockIdx.y].geom;
  write_offsets_body(N, at<uint32_t>(geom, L.tiles), at<uint32_t>(geom, L.block_sums), at<uint32_t>(geom, L.offsets));
}
__global__ void __launch_bounds__(256) emit_keys_batched_kernel(int N, int tiles_x, uint32_t R_cap, GeomLayout L,
                                                                size_t keys_off, size_t vals_off, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  emit_keys_body(N, tiles_x, R_cap, at<Splat>(r.geom, L.splat), at<uint16_t>(r.geom, L.rect),
                 at<uint32_t>(r.geom, L.offsets), at<uint32_t>(r.geom, L.total), at<uint64_t>(r.bin, keys_off),
                 at<uint32_t>(r.bin, vals_off));
}
__global__ void __launch_bounds__(SORT_BLOCK) radix_hist_batched_kernel(size_t keys_off, size_t total_off,
                                                                        uint32_t R_cap, int shift,
                                                                        uint32_t num_blocks, size_t hist_off,
                                                                        RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  radix_hist_body(at<uint64_t>(r.bin, keys_off), at<uint32_t>(r.geom, total_off), R_cap, shift, num_blocks,
                  at<uint32_t>(r.bin, hist_off));
}
__global__ void __launch_bounds__(256) radix_rowscan_batched_kernel(uint32_t num_blocks, size_t hist_off,
                                                                    RenderBatch b) {
  radix_rowscan_body(num_blocks, at<uint32_t>(b.r[blockIdx.y].bin, hist_off));
}
__global__ void __launch_bounds__(SORT_BLOCK) radix_scatter_batched_kernel(size_t kin, size_t vin, size_t kout,
                                                                           size_t vout, size_t total_off,
                                                                           uint32_t R_cap, int shift,
                                                                           uint32_t num_blocks, size_t hist_off,
                                                                           RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  radix_scatter_body(at<uint64_t>(r.bin, kin), at<uint32_t>(r.bin, vin), at<uint64_t>(r.bin, kout),
                     at<uint32_t>(r.bin, vout), at<uint32_t>(r.geom, total_off), R_cap, shift, num_blocks,
                     at<uint32_t>(r.bin, hist_off));
}
__global__ void __launch_bounds__(256) clear_ranges_batched_kernel(int T, size_t ranges_off, size_t work_off,
                                                                   RenderBatch b) {
  void *bin = b.r[blockIdx.y].bin;
  clear_ranges_body(T, at<uint32_t>(bin, ranges_off), at<uint32_t>(bin, work_off));
}
__global__ void __launch_bounds__(256) tile_ranges_batched_kernel(size_t keys_off, size_t total_off, uint32_t R_cap,
                                                                  size_t ranges_off, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  tile_ranges_body(at<uint64_t>(r.bin, keys_off), at<uint32_t>(r.geom, total_off), R_cap,
                   at<uint32_t>(r.bin, ranges_off));
}
__global__ void __launch_bounds__(256) tile_depth_sort_batched_kernel(size_t ranges_off, size_t keys_off,
                                                                      size_t vals_off, RenderBatch b) {
  void *bin = b.r[blockIdx.y].bin;
  tile_depth_sort_body(at<uint32_t>(bin, ranges_off), at<uint64_t>(bin, keys_off), at<uint32_t>(bin, vals_off));
}

int scan_block_sums(int nb, uint32_t *block_sums, uint32_t *total, hipStream_t stream) {
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, nb, block_sums, total);
  return check_launch();
}

int write_offsets(int N, const uint32_t *tiles, const uint32_t *block_sums, uint32_t *offsets, hipStream_t stream) {
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nb == 0) return DIMO_OK;
  hipLaunchKernelGGL(write_offsets_kernel, dim3(nb), dim3(PRE_BLOCK), 0, stream, N, tiles, block_sums, offsets);
  return check_launch();
}

// keys_a/vals_a receive the emission and stay intact; passes ping-pong between the scratch pair
// (keys_c/vals_c) and the sorted pair (keys_b/vals_b) so that the last pass always lands in keys_b.
int bin_instances(int N, int H, int W, int64_t R_cap, const void *geom, void *bin, hipStream_t stream) {
  GeomLayout G(N);
  BinLayout B(R_cap, H, W);
  const uint32_t cap = (uint32_t)B.cap;
  uint32_t *total = const_cast<uint32_t *>(at<uint32_t>(geom, G.total));
  uint64_t *keys_u = at<uint64_t>(bin, B.keys_a), *keys_s = at<uint64_t>(bin, B.keys_b);
  uint32_t *vals_u = at<uint32_t>(bin, B.vals_a), *vals_s = at<uint32_t>(bin, B.vals_b);
  uint64_t *keys_c = at<uint64_t>(bin, B.keys_c);
  uint32_t *vals_c = at<uint32_t>(bin, B.vals_c);
  uint32_t *ranges = at<uint32_t>(bin, B.ranges), *hist = at<uint32_t>(bin, B.hist);
  const int tile_bits = key_bits(B.T) - 32;  // the radix passes cover the tile id only
  const int passes = (tile_bits + RADIX_BITS - 1) / RADIX_BITS;

  if (N > 0) {
    ScopedTimer tm(T_EMIT, stream);
    hipLaunchKernelGGL(emit_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, B.tiles_x, cap,
                       at<Splat>(geom, G.splat), at<uint16_t>(geom, G.rect), at<uint32_t>(geom, G.offsets), total,
                       keys_u, vals_u);
  }
  const uint32_t nblk = (uint32_t)B.sort_blocks;
  const uint64_t *kin = keys_u;
  const uint32_t *vin = vals_u;
  ScopedTimer *sort_tm = new ScopedTimer(T_SORT, stream);
  for (int p = 0; p < passes; ++p) {
    const bool to_sorted = ((passes - 1 - p) & 1) == 0;
    uint64_t *kout = to_sorted ? keys_s : keys_c;
    uint32_t *vout = to_sorted ? vals_s : vals_c;
    const int shift = 32 + p * RADIX_BITS;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(SORT_BLOCK), 0, stream, kin, total, cap, shift, nblk, hist);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(RADIX / 4), dim3(256), 0, stream, nblk, hist);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(SORT_BLOCK), 0, stream, kin, vin, kout, vout, total, cap,
                       shift, nblk, hist);
    kin = kout, vin = vout;
  }
  delete sort_tm;
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(clear_ranges_kernel, dim3((2 * B.T + 255) / 256), dim3(256), 0, stream, B.T, ranges,
                       at<uint32_t>(bin, B.work));
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, keys_s, total, cap, ranges);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(tile_depth_sort_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)TS_LDS_BYTES);
    (void)attr;
    hipLaunchKernelGGL(tile_depth_sort_kernel, dim3(B.T), dim3(256), TS_LDS_BYTES, stream, ranges, keys_s, vals_s);
  }
  return check_launch();
}

int scan_offsets_batched(int N, const GeomLayout &L, const RenderBatch &b, int n, hipStream_t stream) {
  const int nb = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  hipLaunchKernelGGL(scan_block_sums_batched_kernel, dim3(1, n), dim3(1024), 0, stream, nb, L, b);
  if (nb > 0) hipLaunchKernelGGL(write_offsets_batched_kernel, dim3(nb, n), dim3(PRE_BLOCK), 0, stream, N, L, b);
  return check_launch();
}

int bin_instances_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W);
  if (c.bin_bytes < B.bytes) return DIMO_E_WORKSPACE;
  const uint32_t cap = (uint32_t)B.cap;
  const int tile_bits = key_bits(B.T) - 32;
  const int passes = (tile_bits + RADIX_BITS - 1) / RADIX_BITS;
  if (c.N > 0) {
    ScopedTimer tm(T_EMIT, stream);
    hipLaunchKernelGGL(emit_keys_batched_kernel, dim3((c.N + 255) / 256, n), dim3(256), 0, stream, c.N, B.tiles_x, cap,
                       G, B.keys_a, B.vals_a, b);
  }
  const uint32_t nblk = (uint32_t)B.sort_blocks;
  size_t kin = B.keys_a, vin = B.vals_a;
  {
    ScopedTimer tm(T_SORT, stream);
    for (int p = 0; p < passes; ++p) {
      const bool to_sorted = ((passes - 1 - p) & 1) == 0;
      const size_t kout = to_sorted ? B.keys_b : B.keys_c, vout = to_sorted ? B.vals_b : B.vals_c;
      const int shift = 32 + p * RADIX_BITS;
      hipLaunchKernelGGL(radix_hist_batched_kernel, dim3(nblk, n), dim3(SORT_BLOCK), 0, stream, kin, G.total, cap,
                         shift, nblk, B.hist, b);
      hipLaunchKernelGGL(radix_rowscan_batched_kernel, dim3(RADIX / 4, n), dim3(256), 0, stream, nblk, B.hist, b);
      hipLaunchKernelGGL(radix_scatter_batched_kernel, dim3(nblk, n), dim3(SORT_BLOCK), 0, stream, kin, vin, kout,
                         vout, G.total, cap, shift, nblk, B.hist, b);
      kin = kout, vin = vout;
    }
  }
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(clear_ranges_batched_kernel, dim3((2 * B.T + 255) / 256, n), dim3(256), 0, stream, B.T,
                       B.ranges, B.work, b);
    hipLaunchKernelGGL(tile_ranges_batched_kernel, dim3((cap + 255) / 256, n), dim3(256), 0, stream, B.keys_b, G.total,
                       cap, B.ranges, b);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void *>(tile_depth_sort_batched_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)TS_LDS_BYTES);
    (void)attr;
    hipLaunchKernelGGL(tile_depth_sort_batched_kernel, dim3(B.T, n), dim3(256), TS_LDS_BYTES, stream, B.ranges,
                       B.keys_b, B.vals_b, b);
  }
  return check_launch();
}

}  // namespace dimo
