// Tile binning: scan of tiles_touched, key emission, stable LSD radix sort of
// (tile | fp32 depth bits) keys with the Gaussian id as payload, and tile ranges.
//
// All kernels read the live instance count R from device memory (geom.total[0]) and are launched
// over the CAPACITY R_cap, so the whole chain is enqueued without a host round trip.
#include "common.hpp"

namespace dimo {

// ------------------------------------------------------------------------------------ scan
// Exclusive scan of the per-block sums (nb <= a few thousand) by one workgroup; writes the
// grand total R to total[0] and clears the overflow flag total[1].
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nb, uint32_t *__restrict__ sums,
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
__global__ void __launch_bounds__(PRE_BLOCK) write_offsets_kernel(int N, const uint32_t *__restrict__ tiles,
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
__global__ void __launch_bounds__(256) emit_keys_kernel(int N, int tiles_x, uint32_t R_cap,
                                                        const Splat *__restrict__ splat,
                                                        const uint16_t *__restrict__ rect,
                                                        const uint32_t *__restrict__ offsets,
                                                        uint32_t *__restrict__ total, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
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

__global__ void __launch_bounds__(SORT_BLOCK) radix_hist_kernel(const uint64_t *__restrict__ keys,
                                                                const uint32_t *__restrict__ total, uint32_t R_cap,
                                                                int shift, uint32_t num_blocks,
                                                                uint32_t *__restrict__ hist) {
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
__global__ void __launch_bounds__(256) radix_rowscan_kernel(uint32_t num_blocks, uint32_t *__restrict__ hist) {
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

__global__ void __launch_bounds__(SORT_BLOCK) radix_scatter_kernel(
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
__global__ void __launch_bounds__(256) clear_ranges_kernel(int T, uint32_t *__restrict__ ranges) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * T) ranges[i] = 0;
}
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint64_t *__restrict__ keys,
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
  const int bits = key_bits(B.T);
  const int passes = (bits + RADIX_BITS - 1) / RADIX_BITS;

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
    const int shift = p * RADIX_BITS;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(SORT_BLOCK), 0, stream, kin, total, cap, shift, nblk, hist);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(RADIX / 4), dim3(256), 0, stream, nblk, hist);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(SORT_BLOCK), 0, stream, kin, vin, kout, vout, total, cap,
                       shift, nblk, hist);
    kin = kout, vin = vout;
  }
  delete sort_tm;
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(clear_ranges_kernel, dim3((2 * B.T + 255) / 256), dim3(256), 0, stream, B.T, ranges);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, keys_s, total, cap, ranges);
  }
  return check_launch();
}

}  // namespace dimo
