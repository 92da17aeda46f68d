// Tile binning: produces, per tile, the list of Gaussians that touch it in (depth, id) order -- bit for bit the
// arrays a stable radix sort of the (tile | fp32 depth bits) keys of all (Gaussian, tile) instances yields (the
// published rasterizer's cub::DeviceRadixSort) -- without ever sorting the instances, in FIVE short launches:
//   1. level1_count   : a Gaussian has one LEVEL-1 ENTRY per supertile (SS x SS tiles) it touches; an entry belongs
//                       to the BUCKET (supertile, coarse depth bin).  A workgroup counts its 256 Gaussians' entries
//                       per bucket in LDS; one returning atomic per (workgroup, non-empty bucket) on the bucket total
//                       reserves the workgroup's share of the bucket.  Also: offsets (the inclusive scan of
//                       tiles_touched), R, and the depth-bin map -- every workgroup reduces the per-block words
//                       preprocess left (a few hundred) itself instead of waiting for a scan kernel.
//   2. level1_scatter : bucket starts (every workgroup scans the <= 2048 bucket totals itself: supertile lists one
//                       after the other, starts rounded up to a window of 256 entries, a supertile's buckets by depth
//                       bin), then every entry (depth bits, id, tile rectangle) goes to start + share + an LDS cursor.
//                       No ordering is kept inside a bucket.  Workgroup 0 also writes what the next kernels read:
//                       bucket starts, the level-2 window table, the slice list of oversized buckets.
//   3. bucket_sort    : ONE WORKGROUP PER BUCKET sorts its (contiguous) entries in LDS by the 64-bit word
//                       (depth bits << 32 | id): a counting pass over 256 sub-bins of the bucket's key range, then every
//                       entry ranks itself inside its sub-bin (all words are distinct because the ids are).  The
//                       per-supertile lists come out in (depth, id) order.
//   4. level2_count   : every 256-entry window of the level-1 array belongs to one supertile; a workgroup filters its
//                       windows against the <= 64 tiles of their supertile (ballot / popcount give order-preserving
//                       ranks), leaves per-tile counts per window and adds them to the tile totals.
//   5. level2_fill    : the same walk; a window's first slot per tile = tile start (every workgroup scans the tile
//                       totals itself) + the counts of the supertile's earlier windows (summed by the workgroup).
//                       Writes (depth bits, id) per instance as long contiguous runs, the tile ranges, the overflow
//                       flag, and clears the backward's record flags.
// An ORDERED filter keeps the input order, so every tile list comes out in (depth, id) order.  A frame has ~10x fewer
// Gaussians than instances (1e5 vs 1e6 at the benchmark configuration).  Earlier versions: 6 radix passes over the
// instances; 2 passes + a per-tile LDS sort; then (round 2) a depth sort of the Gaussians + two levels of ordered
// filters in twelve launches, whose cost was their dependent single-workgroup scans and chains of dependent loads
// (183 us per 8 renders for ~25 us of memory traffic; a kernel boundary itself costs ~1.5 us on this chip).  The
// level-2 kernels are persistent over their windows and request the next window's entries before they work on the
// current one.  The 64-bit (tile | depth) key of an instance is not stored: its tile is the list the instance sits
// in (`ranges`), its depth bits are `dkeys`.
//
// Everything reads live counts from device memory and is sized by capacities, so the whole chain is enqueued without
// a host round trip.
#include <type_traits>

#include "common.hpp"
#include <cstdlib>

namespace dimo {

typedef unsigned long long u64;

// Depth-bin map: keys at or below `lo` fall into bin 0, keys past the last bin into the last one
__device__ __forceinline__ uint32_t depth_bin(uint32_t key, uint32_t lo, uint32_t shift, uint32_t nbins) {
  return key <= lo ? 0u : min((key - lo) >> shift, nbins - 1u);
}

// Sub-bins of a bucket (monotone in the key): SUB_BINS = 1024 over the 2^shift keys of its depth bin (an entry ranks
// itself against the other entries of its sub-bin: the cost is the sum of the squared sub-bin sizes, 16 of the
// kernel's 54 us per 8 renders with 256 sub-bins).  Depth bin 0 also holds every key below the binned range (outliers
// in front of the scene): those share sub-bin 0 and the bin's own keys get the upper half of the sub-bins -- with ONE
// linear map from the smallest key a single floater would squeeze the bin's bulk into a few fat sub-bins.  Keys past
// the last bin share the last sub-bin.
constexpr int SUB_BINS = 1024;
struct SubMap {
  uint32_t klo, off, sh;
};
__device__ __forceinline__ SubMap sub_bin_map(uint32_t map_lo, uint32_t map_shift, uint32_t dbin) {
  SubMap m;
  const uint32_t shift = map_shift < 32u ? map_shift : 31u;
  m.klo = map_lo + (dbin << shift);
  m.off = dbin == 0u ? (uint32_t)SUB_BINS / 2u : 0u;
  const uint32_t bits = dbin == 0u ? 9u : 10u;
  static_assert(SUB_BINS == 1024, "bits");
  m.sh = shift > bits ? shift - bits : 0u;
  return m;
}
__device__ __forceinline__ uint32_t sub_bin(uint32_t key, const SubMap &m) {
  return key < m.klo ? 0u : min(m.off + ((key - m.klo) >> m.sh), (uint32_t)SUB_BINS - 1u);
}

constexpr int SEG = 256;  // list entries per level-2 window: 4 waves x 64
// level-1 metadata in the bin workspace (uint32): [0, 256) list length, [256, 512) list start (multiple of SEG),
// [512] number of level-2 windows, [1024, ...) four words per window: supertile | valid entries << 16, first window
// of that supertile, first tile x | y << 16, 0
constexpr int META_LEN = 0, META_START = MAX_SUPER, META_NWIN = 2 * MAX_SUPER, META_WIN = 4 * MAX_SUPER;

struct BinArgs {
  int N, nb, per, nwg1, lg, T;  // Gaussians, preprocess blocks, blocks per level-1 workgroup, level-1 workgroups,
                                // log2 depth bins, tiles
  uint32_t R_cap;
  BinGrid gi;
  size_t l1cap, max_windows;
  // byte offsets into the geometry (g_) and bin (b_) workspaces
  size_t g_total, g_rect, g_tiles, g_offsets, g_sums, g_key32, g_bk, g_wgbase;
  size_t b_order;
  size_t b_meta, b_l1tmp, b_l1a, b_l1b, b_l1, b_cnt2, b_totals, b_ranges, b_work, b_dkeys, b_vals;
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// Optional per-workgroup phase trace (diagnostics; compiled in with -DDIMO_BIN_TRACE -- build.py does that when the
// environment has DIMO_BIN_TRACE=1 -- and then off unless dimo_debug_bin_trace set a buffer): 32 x u64 per workgroup
// and launch -- [0] = kernel << 56 | render << 48 | workgroup, [1 ...] = s_memrealtime (100 MHz) at the marks.
__device__ unsigned long long *g_bin_trace = nullptr;
__device__ unsigned int g_bin_trace_cap = 0;
__device__ unsigned int g_bin_trace_n = 0;
struct BinTrace {
#ifdef DIMO_BIN_TRACE
  unsigned long long t[32];
  int n = 1;
  __device__ __forceinline__ BinTrace(int kernel) {
    t[0] = ((unsigned long long)kernel << 56) | ((unsigned long long)blockIdx.y << 48) | blockIdx.x;
    mark();
  }
  __device__ __forceinline__ void mark() {
    if (n < 32) t[n++] = __builtin_amdgcn_s_memrealtime();
  }
  __device__ __forceinline__ void flush() {
    mark();
    unsigned long long *buf = g_bin_trace;
    if (!buf || threadIdx.x != 0) return;
    const unsigned int slot = atomicAdd(&g_bin_trace_n, 1u);
    if (slot >= g_bin_trace_cap) return;
    for (int i = 0; i < 32; ++i) buf[(size_t)slot * 32 + i] = i < n ? t[i] : 0ull;
  }
#else
  __device__ __forceinline__ BinTrace(int) {}
  __device__ __forceinline__ void mark() {}
  __device__ __forceinline__ void flush() {}
#endif
};


// The level-1 entries of a wave's 64 Gaussians, walked in GROUPS of lanes that name the same bucket: a wave's
// Gaussians are Morton neighbours, so most of its entries share a handful of buckets -- one LDS atomic per entry was
// a 64-way same-address conflict per instruction.  Lane `lane` has `cnt` entries:
// supertiles (sx0 + e % nx, sy0 + e / nx), depth bin db.  f(bucket, lanes of the group, leader lane, this lane is in it).
template <class F, class G>
__device__ __forceinline__ void for_each_group(bool has, const uint2 rc, uint32_t db, const BinGrid &gi, int lg, int lane,
                                               F f, G direct) {
  const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
  const int sx0 = x0 >> gi.ss_shift, sy0 = y0 >> gi.ss_shift;
  const int nx = has ? ((x1 - 1) >> gi.ss_shift) - sx0 + 1 : 0, ny = has ? ((y1 - 1) >> gi.ss_shift) - sy0 + 1 : 0;
  int cx = 0, cy = 0;
  for (;;) {
    const bool on = cy < ny && nx > 0;
    u64 todo = __ballot(on);
    if (todo == 0) break;
    const uint32_t bucket = ((uint32_t)((sy0 + cy) * gi.stx + sx0 + cx) << lg) + db;
    // up to four groups by their leaders; a wave whose lanes name many buckets (Gaussians in no spatial order: a model
    // that was never Morton-sorted) has no conflicts to avoid -- its remaining lanes go one by one
    for (int round = 0; round < 4 && todo; ++round) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)bucket, leader);
      const bool mine = on && bucket == bl;
      const u64 m = __ballot(mine);
      f(bl, m, leader, mine);
      todo &= ~m;
    }
    if (todo) direct(bucket, (todo >> lane) & 1ull);
    if (++cx >= nx) cx = 0, ++cy;
  }
}

// A Gaussian that touches MORE than BIG_ENTRIES supertiles (a splat blown up over a good part of the image) is not
// walked by its lane -- the whole wave would iterate with it -- but put on a list and walked by the workgroup, one
// thread per supertile, once the lanes are done (distinct supertiles: distinct buckets, plain LDS atomics).
constexpr int BIG_ENTRIES = 4;
__device__ __forceinline__ int rect_entries(const uint2 rc, const BinGrid &gi) {
  const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
  return (((x1 - 1) >> gi.ss_shift) - (x0 >> gi.ss_shift) + 1) * (((y1 - 1) >> gi.ss_shift) - (y0 >> gi.ss_shift) + 1);
}
// (The list keeps the Gaussian's key and rectangle beside its index, and every WAVE walks a Gaussian of its own, a
// lane per supertile: with the words re-read from global memory inside a workgroup-wide loop, a stage-s1 model --
// 512 Gaussians, each over most of the image -- spent 75 + 93 us per launch in 256 dependent round trips per workgroup.)
struct BigList {
  uint32_t n;
  uint32_t i[PRE_BLOCK], key[PRE_BLOCK];
  uint2 rc[PRE_BLOCK];
  __device__ __forceinline__ void push(uint32_t gi_, uint32_t k, uint2 r) {
    const uint32_t slot = atomicAdd(&n, 1u);
    i[slot] = gi_, key[slot] = k, rc[slot] = r;
  }
};
template <class F>
__device__ __forceinline__ void for_each_big(const BigList &L, uint32_t lo, uint32_t shift, uint32_t nbins,
                                             const BinGrid &gi, int lg, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_big = L.n;
  for (uint32_t b = (uint32_t)wave; b < n_big; b += SORT_BLOCK / 64) {
    const uint32_t i = L.i[b], key = L.key[b];
    const uint2 rc = L.rc[b];
    const uint32_t db = depth_bin(key, lo, shift, nbins);
    const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff;
    const int sx0 = x0 >> gi.ss_shift, sy0 = y0 >> gi.ss_shift, nx = ((x1 - 1) >> gi.ss_shift) - sx0 + 1;
    const int cnt = rect_entries(rc, gi);
    for (int e = lane; e < cnt; e += 64)
      f((((uint32_t)((sy0 + e / nx) * gi.stx + sx0 + e % nx)) << lg) + db, i, key, rc);
  }
}

// ------------------------------------------------------------------------------------ 1. level-1 count
__device__ __forceinline__ void level1_count_body(const BinArgs &a, void *geom, void *bin) {
  __shared__ uint32_t s_hist[MAX_BUCKETS];
  __shared__ uint32_t s_w[4][8];
  __shared__ uint32_t s_wt[4];
  __shared__ BigList s_big;
  const uint32_t *__restrict__ tiles = at<uint32_t>(geom, a.g_tiles);
  const uint32_t *__restrict__ key32 = at<uint32_t>(geom, a.g_key32);
  const uint16_t *__restrict__ rect = at<uint16_t>(geom, a.g_rect);
  const uint32_t *__restrict__ sums = at<uint32_t>(geom, a.g_sums);
  uint32_t *__restrict__ offsets = at<uint32_t>(geom, a.g_offsets);
  uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  uint32_t *__restrict__ wgbase = at<uint32_t>(geom, a.g_wgbase) + (size_t)blockIdx.x * MAX_BUCKETS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = (int)blockIdx.x * a.per, c1 = min(a.nb, c0 + a.per);  // this workgroup's preprocess blocks
  const int stride = a.nb + 1;
  const int nbuckets = a.gi.NS << a.lg;
  BinTrace tr(1);
  // the first block's Gaussians, requested before anything else (one memory round trip with the reductions below)
  const int i0 = c0 * PRE_BLOCK + tid;
  uint32_t v0 = i0 < a.N ? tiles[i0] : 0u;
  uint32_t k0 = i0 < a.N ? key32[i0] : 0u;
  uint2 r0 = i0 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i0) : make_uint2(0u, 0u);
  for (int j = tid; j < nbuckets; j += SORT_BLOCK) s_hist[j] = 0u;
  // ---- the per-block words of preprocess, reduced by every workgroup itself: instances before this workgroup's
  // blocks and in total, and the depth-bin map.  The binned range is NOT [min, max] of the keys: a few floaters far behind
  // (or in front of) the scene would stretch it until the scene's bulk shares a handful of bins.  A block's MIN ignores
  // a far outlier inside it and its MAX a near one, so A = the largest block minimum and B = the smallest block maximum
  // bracket the bulk whatever the order of the Gaussians (Morton order: A ~ far end, B ~ near end; random order: the
  // other way round); the range is [min(A, B), max(A, B)] widened by a quarter on both sides, inside [min, max].
  // Keys outside it fall into the first / last bin (whose sub-bins start at the smallest key).
  uint32_t pre_t = 0, tot_t = 0, tot_e = 0, mn = 0xffffffffu, mx = 0u, A_ = 0u, B_ = 0xffffffffu;
  for (int i = tid; i < a.nb; i += SORT_BLOCK) {
    const uint32_t t = sums[i], bmin = sums[stride + i], bmax = sums[2 * stride + i], e = sums[3 * stride + i];
    tot_t += t, tot_e += e;
    if (i < c0) pre_t += t;
    if (bmin != 0xffffffffu) mn = min(mn, bmin), mx = max(mx, bmax), A_ = max(A_, bmin), B_ = min(B_, bmax);
  }
  pre_t = wave_sum(pre_t), tot_t = wave_sum(tot_t), tot_e = wave_sum(tot_e);
  mn = wave_min(mn), mx = wave_max(mx), A_ = wave_max(A_), B_ = wave_min(B_);
  if (lane == 0) {
    s_w[wave][0] = pre_t, s_w[wave][1] = tot_t, s_w[wave][2] = tot_e;
    s_w[wave][4] = mn, s_w[wave][5] = mx, s_w[wave][6] = A_, s_w[wave][7] = B_;
  }
  __syncthreads();
  pre_t = s_w[0][0] + s_w[1][0] + s_w[2][0] + s_w[3][0];
  tot_t = s_w[0][1] + s_w[1][1] + s_w[2][1] + s_w[3][1];
  tot_e = s_w[0][2] + s_w[1][2] + s_w[2][2] + s_w[3][2];
  mn = min(min(s_w[0][4], s_w[1][4]), min(s_w[2][4], s_w[3][4]));
  mx = max(max(s_w[0][5], s_w[1][5]), max(s_w[2][5], s_w[3][5]));
  A_ = max(max(s_w[0][6], s_w[1][6]), max(s_w[2][6], s_w[3][6]));
  B_ = min(min(s_w[0][7], s_w[1][7]), min(s_w[2][7], s_w[3][7]));
  uint32_t lo = mn, span = 0u;
  if (mx >= mn) {  // (some Gaussian touches a tile)
    const uint32_t rl = min(A_, B_), rh = max(A_, B_), ext = (rh - rl) >> 2;
    lo = rl - mn > ext ? rl - ext : mn;
    const uint32_t hi = mx - rh > ext ? rh + ext : mx;
    span = hi - lo;
  }
  const int nbits = span ? 32 - __clz((int)span) : 0;
  const uint32_t shift = (uint32_t)(nbits > a.lg ? nbits - a.lg : 0), nbins = 1u << a.lg;
  if (blockIdx.x == 0) {
    uint32_t *total = at<uint32_t>(geom, a.g_total);
    if (tid == 0) {
      bk[BK_KMIN] = lo, bk[BK_SHIFT] = shift, bk[BK_KMIN0] = min(mn, lo), bk[BK_NBLOG] = (uint32_t)a.lg;
      total[0] = tot_t, total[1] = 0u, total[2] = tot_e, total[3] = 0u;
    }
    uint32_t *tile_tot = at<uint32_t>(bin, a.b_totals);  // summed atomically by level2_count
    for (int t = tid; t < a.T; t += SORT_BLOCK) tile_tot[t] = 0u;
  }
  tr.mark();
  // ---- offsets, and this workgroup's entries per bucket
  uint32_t carry = pre_t;
  uint32_t vn = v0, kn = k0;
  uint2 rn = r0;
  for (int c = c0; c < c1; ++c) {
    const int i = c * PRE_BLOCK + tid;
    v0 = vn, k0 = kn, r0 = rn;
    if (c + 1 < c1) {  // the next block's Gaussians, requested before this block is worked on
      const int i2 = i + PRE_BLOCK;
      vn = i2 < a.N ? tiles[i2] : 0u;
      kn = i2 < a.N ? key32[i2] : 0u;
      rn = i2 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i2) : make_uint2(0u, 0u);
    }
    const uint32_t inc = wave_scan_incl(v0, lane);
    lds_barrier();  // (s_wt / the list of the previous block have been read; first round: s_hist cleared)
    if (lane == 63) s_wt[wave] = inc;
    if (tid == 0) s_big.n = 0u;
    lds_barrier();
    uint32_t off = carry;
    for (int w = 0; w < wave; ++w) off += s_wt[w];
    carry += s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
    if (i < a.N) offsets[i] = off + inc;
    const bool big = v0 != 0u && rect_entries(r0, a.gi) > BIG_ENTRIES;
    if (big) s_big.push((uint32_t)i, k0, r0);
    for_each_group(v0 != 0u && !big, r0, depth_bin(k0, lo, shift, nbins), a.gi, a.lg, lane,
                   [&](uint32_t bl, u64 m, int leader, bool) {
                     if (lane == leader) atomicAdd(&s_hist[bl], (uint32_t)__popcll(m));
                   },
                   [&](uint32_t bucket, bool mine) {
                     if (mine) atomicAdd(&s_hist[bucket], 1u);
                   });
    lds_barrier();
    for_each_big(s_big, lo, shift, nbins, a.gi, a.lg,
                 [&](uint32_t bucket, uint32_t, uint32_t, uint2) { atomicAdd(&s_hist[bucket], 1u); });
  }
  __syncthreads();
  tr.mark();
  // ---- every non-empty bucket of the workgroup: its share of the bucket (thread t owns counters [8 t, 8 t + 8))
  {
    constexpr int OWN = MAX_BUCKETS / SORT_BLOCK;
    uint32_t cnt[OWN], base[OWN];
#pragma unroll
    for (int u = 0; u < OWN; ++u) cnt[u] = tid * OWN + u < nbuckets ? s_hist[tid * OWN + u] : 0u;
#pragma unroll
    for (int u = 0; u < OWN; ++u) base[u] = cnt[u] ? atomicAdd(&bk[BK_TOT + tid * OWN + u], cnt[u]) : 0u;
#pragma unroll
    for (int u = 0; u < OWN; ++u)
      if (cnt[u]) wgbase[tid * OWN + u] = base[u];
  }
  tr.flush();
}

// ------------------------------------------------------------------------------------ 2. level-1 scatter
constexpr int BIN_CAP = 2048;      // entries a workgroup sorts in LDS (16 KB)
constexpr int SUB_MAX = 512;       // largest sub-bin ranked quadratically
// A bucket above BIN_CAP (thousands of Gaussians of one supertile in one depth bin) is cut into SLICES of
// ~SLICE_TARGET entries along its sub-bins, sorted by the extra workgroups of the bucket_sort launch; only a SUB-bin
// above SUB_MAX (hundreds of Gaussians at nearly ONE depth) sends its bucket to eight stable byte passes of a single
// workgroup through global memory: slow, correct, rare.  l1tmp is read-only in bucket_sort, so every slice of a
// bucket sees the same counts and takes the same decision.
constexpr int SLICE_TARGET = BIN_CAP - SUB_MAX;
constexpr int PER = BIN_CAP / SORT_BLOCK;
constexpr int MAXB = 8;            // buckets one bucket_sort workgroup walks
constexpr int SUB_OWN = SUB_BINS / SORT_BLOCK;  // sub-bins a thread owns in the scans

__device__ __forceinline__ void level1_scatter_body(const BinArgs &a, void *geom, void *bin) {
  __shared__ uint32_t s_cur[MAX_BUCKETS];       // bucket totals, then bucket starts, then this workgroup's cursors
  __shared__ uint32_t s_lstart[MAX_SUPER + 1];  // list start per supertile (multiples of SEG), [NS] = end
  __shared__ uint32_t s_len[MAX_SUPER];
  __shared__ uint32_t s_wt[4];
  __shared__ uint32_t s_nslice;
  __shared__ BigList s_big;
  const uint32_t *__restrict__ tiles = at<uint32_t>(geom, a.g_tiles);
  const uint32_t *__restrict__ key32 = at<uint32_t>(geom, a.g_key32);
  const uint16_t *__restrict__ rect = at<uint16_t>(geom, a.g_rect);
  uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  const uint32_t *__restrict__ wgbase = at<uint32_t>(geom, a.g_wgbase) + (size_t)blockIdx.x * MAX_BUCKETS;
  uint4 *__restrict__ l1tmp = at<uint4>(bin, a.b_l1tmp);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = (int)blockIdx.x * a.per, c1 = min(a.nb, c0 + a.per);
  const int NS = a.gi.NS, nbins = 1 << a.lg, nbuckets = NS << a.lg;
  constexpr int OWN = MAX_BUCKETS / SORT_BLOCK;
  BinTrace tr(2);
  // everything this workgroup reads, requested at once: its first block's Gaussians, the bucket totals, its shares
  // (the words of buckets it does not touch are stale and never used), the bin map
  const int i0 = c0 * PRE_BLOCK + tid;
  uint32_t v0 = i0 < a.N ? tiles[i0] : 0u;
  uint32_t k0 = i0 < a.N ? key32[i0] : 0u;
  uint2 r0 = i0 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i0) : make_uint2(0u, 0u);
  uint32_t share[OWN];
#pragma unroll
  for (int u = 0; u < OWN; ++u) share[u] = tid * OWN + u < nbuckets ? wgbase[tid * OWN + u] : 0u;
  for (int j = tid; j < nbuckets; j += SORT_BLOCK) s_cur[j] = bk[BK_TOT + j];
  const uint32_t lo = bk[BK_KMIN], shift = bk[BK_SHIFT];
  __syncthreads();
  tr.mark();
  // ---- where the buckets start
  {
    uint32_t len = 0;
    if (tid < NS)
      for (int j = 0; j < nbins; ++j) len += s_cur[(tid << a.lg) + j];
    const uint32_t padded = (len + SEG - 1) / SEG * SEG;
    const uint32_t inc = wave_scan_incl(padded, lane);
    if (lane == 63) s_wt[wave] = inc;
    if (tid == 0) s_nslice = 0u;
    __syncthreads();
    uint32_t base = inc - padded;
    for (int w = 0; w < wave; ++w) base += s_wt[w];
    if (tid < NS) {
      s_lstart[tid] = base, s_len[tid] = len;
      if (tid == NS - 1) s_lstart[NS] = base + padded;
      uint32_t run = base;
      for (int j = 0; j < nbins; ++j) {  // totals -> starts, in place
        const uint32_t n = s_cur[(tid << a.lg) + j];
        s_cur[(tid << a.lg) + j] = run;
        if (blockIdx.x == 0) {
          const uint32_t b = ((uint32_t)tid << a.lg) + j;
          bk[BK_START + b] = run;
          uint32_t J = 0;
          if (n > (uint32_t)BIN_CAP) {  // an oversized bucket: its slices go on the list (if they fit; else J stays 0)
            const uint32_t want = (n + SLICE_TARGET - 1) / SLICE_TARGET;
            const uint32_t pos = want <= 255u ? atomicAdd(&s_nslice, want) : (uint32_t)MAX_SLICES;
            if (pos + want <= (uint32_t)MAX_SLICES) {
              J = want;
              for (uint32_t q = 0; q < want; ++q) bk[BK_SLICE + pos + q] = (b << 16) | (want << 8) | q;
            } else if (want <= 255u) {  // (list full: the entries this bucket reserved stay unused; mark them so)
              for (uint32_t q = pos; q < min(pos + want, (uint32_t)MAX_SLICES); ++q) bk[BK_SLICE + q] = 0xffffffffu;
            }
          }
          bk[BK_BINJ + b] = J;
        }
        run += n;
      }
    }
    __syncthreads();
    if (blockIdx.x == 0) {  // the level-2 kernels' view of the lists
      uint32_t *meta = at<uint32_t>(bin, a.b_meta);
      if (tid == 0) bk[BK_NSLICE] = min(s_nslice, (uint32_t)MAX_SLICES);
      if (tid < NS) meta[META_LEN + tid] = s_len[tid], meta[META_START + tid] = s_lstart[tid];
      // window table, by ALL threads: window w belongs to the last supertile whose start (in windows) is <= w (an
      // empty supertile shares its start with its successor) -- a binary search over the <= 256 starts in LDS
      const uint32_t n_win = (uint32_t)min((size_t)(s_lstart[NS] / SEG), a.max_windows);
      if (tid == 0) meta[META_NWIN] = n_win;
      for (uint32_t w = tid; w < n_win; w += SORT_BLOCK) {
        int l = 0, h = NS;
        while (h - l > 1) {
          const int mid = (l + h) >> 1;
          if (s_lstart[mid] / SEG <= w) l = mid; else h = mid;
        }
        const uint32_t end = s_lstart[l] + s_len[l];
        const uint32_t nvalid = end > w * SEG ? min(end - w * SEG, (uint32_t)SEG) : 0u;
        uint4 *wintab = reinterpret_cast<uint4 *>(meta + META_WIN);
        wintab[w] = make_uint4((uint32_t)l | (nvalid << 16), s_lstart[l] / SEG,
                               (uint32_t)((l % a.gi.stx) << a.gi.ss_shift) | ((uint32_t)((l / a.gi.stx) << a.gi.ss_shift) << 16), 0u);
      }
    }
  }
  // ---- cursors of this workgroup, then every entry to its bucket's next slot (order inside a bucket: whatever the LDS
  // atomics make of it -- bucket_sort orders by the whole 64-bit word)
#pragma unroll
  for (int u = 0; u < OWN; ++u)
    if (tid * OWN + u < nbuckets) s_cur[tid * OWN + u] += share[u];
  __syncthreads();
  tr.mark();
  uint32_t vn = v0, kn = k0;
  uint2 rn = r0;
  for (int c = c0; c < c1; ++c) {
    const int i = c * PRE_BLOCK + tid;
    v0 = vn, k0 = kn, r0 = rn;
    if (c + 1 < c1) {  // the next block's Gaussians, requested before this block is worked on
      const int i2 = i + PRE_BLOCK;
      vn = i2 < a.N ? tiles[i2] : 0u;
      kn = i2 < a.N ? key32[i2] : 0u;
      rn = i2 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i2) : make_uint2(0u, 0u);
    }
    const uint4 en = make_uint4(k0, (uint32_t)i, r0.x, r0.y);
    const u64 lt = (1ull << lane) - 1ull;
    lds_barrier();  // (the list of the previous block has been read)
    if (tid == 0) s_big.n = 0u;
    lds_barrier();
    const bool big = v0 != 0u && rect_entries(r0, a.gi) > BIG_ENTRIES;
    if (big) s_big.push((uint32_t)i, k0, r0);
    for_each_group(v0 != 0u && !big, r0, depth_bin(k0, lo, shift, (uint32_t)nbins), a.gi, a.lg, lane,
                   [&](uint32_t bl, u64 m, int leader, bool mine) {
                     uint32_t first = 0;  // the group's slots: one returning LDS atomic by its leader
                     if (lane == leader) first = atomicAdd(&s_cur[bl], (uint32_t)__popcll(m));
                     first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                     const size_t pos = (size_t)first + (uint32_t)__popcll(m & lt);
                     if (mine && pos < a.l1cap) l1tmp[pos] = en;
                   },
                   [&](uint32_t bucket, bool mine) {
                     if (mine) {
                       const size_t pos = atomicAdd(&s_cur[bucket], 1u);
                       if (pos < a.l1cap) l1tmp[pos] = en;
                     }
                   });
    lds_barrier();
    for_each_big(s_big, lo, shift, (uint32_t)nbins, a.gi, a.lg,
                 [&](uint32_t bucket, uint32_t gi_, uint32_t key, uint2 rc) {
                   const size_t pos = atomicAdd(&s_cur[bucket], 1u);
                   if (pos < a.l1cap) l1tmp[pos] = make_uint4(key, gi_, rc.x, rc.y);
                 });
    tr.mark();
  }
  tr.flush();
}

// ------------------------------------------------------------------------------------ 3. bucket sort

// One stable counting pass of a single workgroup over n 64-bit words: digit = byte `byte` of the word.
__device__ __forceinline__ void wg_radix_pass(const u64 *kin, u64 *kout, uint32_t n, int byte, uint32_t *s_run,
                                              uint32_t (*s_cnt)[256]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 lt = (1ull << lane) - 1ull;
  s_run[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += SORT_BLOCK) atomicAdd(&s_run[(uint32_t)(kin[i] >> (8 * byte)) & 255u], 1u);
  __syncthreads();
  {  // exclusive scan of the 256 digit counts (thread d owns digit d)
    const uint32_t tot = s_run[threadIdx.x];
    const uint32_t inc = wave_scan_incl(tot, lane);
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
    const u64 k = valid ? kin[i] : 0;
    const uint32_t d = (uint32_t)(k >> (8 * byte)) & 255u;
    u64 peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const u64 bal = __ballot((d >> bit) & 1u);
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
    if (valid) kout[s_cnt[wave][d] + before] = k;
    __syncthreads();
  }
}

__device__ __forceinline__ uint4 entry_of(u64 word, const uint16_t *__restrict__ rect) {
  const uint32_t id = (uint32_t)word;
  const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)id);
  return make_uint4(id, (uint32_t)(word >> 32), rc.x, rc.y);
}

// the bucket's words go to la[0, n): eight byte passes a -> b -> ... -> a, then the entries
__device__ __forceinline__ void radix_fallback(const uint4 *in, u64 *la, u64 *lb, uint4 *out, const uint16_t *rect,
                                               uint32_t n, uint32_t *s_run, uint32_t (*s_cnt)[256]) {
  for (uint32_t e = threadIdx.x; e < n; e += SORT_BLOCK) la[e] = ((u64)in[e].x << 32) | (u64)in[e].y;
  for (int byte = 0; byte < 8; ++byte) {
    __threadfence_block();
    __syncthreads();
    wg_radix_pass((byte & 1) ? lb : la, (byte & 1) ? la : lb, n, byte, s_run, s_cnt);
  }
  __threadfence_block();
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < n; e += SORT_BLOCK) out[e] = entry_of(la[e], rect);
}

// exclusive scan of the SUB_BINS counts in s_cur (thread t owns sub-bins [4 t, 4 t + 4)) -> s_start (and, CURSORS,
// back into s_cur); returns true if some sub-bin exceeds SUB_MAX.  Ends with a barrier.
template <bool CURSORS>
__device__ __forceinline__ bool scan_sub_bins(uint32_t *s_cur, uint32_t *s_start, uint32_t *s_run, uint32_t *s_big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t cnt[SUB_OWN], sum = 0;
  bool big = false;
#pragma unroll
  for (int u = 0; u < SUB_OWN; ++u) cnt[u] = s_cur[tid * SUB_OWN + u], sum += cnt[u], big |= cnt[u] > (uint32_t)SUB_MAX;
  if (big) *s_big = 1u;
  const uint32_t inc = wave_scan_incl(sum, lane);
  if (lane == 63) s_run[wave] = inc;
  lds_barrier();
  uint32_t run = inc - sum;
  for (int w = 0; w < wave; ++w) run += s_run[w];
#pragma unroll
  for (int u = 0; u < SUB_OWN; ++u) {
    s_start[tid * SUB_OWN + u] = run;
    if (CURSORS) s_cur[tid * SUB_OWN + u] = run;
    run += cnt[u];
  }
  if (tid == SORT_BLOCK - 1) s_start[SUB_BINS] = run;
  lds_barrier();
  return *s_big != 0u;
}

// every entry of s_k[0, m) ranks itself inside its sub-bin and goes to out[...].  Entries are taken in the order they
// sit in LDS (grouped by sub-bin): the 64 lanes of a wave walk the same one or two sub-bins, so a wave's trip count is
// ITS sub-bins' size, not the largest sub-bin's of the bucket.
__device__ __forceinline__ void rank_and_store(const u64 *s_k, const uint2 *s_r, const uint32_t *s_start, uint32_t origin,
                                               uint32_t m, const SubMap &sm, uint4 *__restrict__ out) {
  for (uint32_t e = threadIdx.x; e < m; e += SORT_BLOCK) {
    const u64 c = s_k[e];
    const uint2 rc = s_r[e];
    const uint32_t f = sub_bin((uint32_t)(c >> 32), sm);
    const uint32_t lo = s_start[f] - origin, hi = min(s_start[f + 1] - origin, m);
    uint32_t r = 0;
    for (uint32_t t = lo; t < hi; t += 8) {  // eight LDS reads in flight (clamped, masked)
      u64 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s_k[min(t + u, hi - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) r += (t + u < hi && v[u] < c) ? 1u : 0u;
    }
    out[lo + r] = make_uint4((uint32_t)c, (uint32_t)(c >> 32), rc.x, rc.y);
  }
}

__device__ __forceinline__ void bucket_sort_body(const BinArgs &a, void *geom, void *bin) {
  __shared__ u64 s_k[BIN_CAP];
  __shared__ uint2 s_r[BIN_CAP];  // the entries' tile rectangles travel with them (a gather of rect[id] per sorted
                                  // entry cost 8 us per 8 renders: random 8-byte reads)
  __shared__ uint32_t s_start[SUB_BINS + 1], s_cur[SUB_BINS];
  __shared__ uint32_t s_run[256];
  __shared__ uint32_t s_big;
  uint32_t (*s_cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_cur);  // (the fallback sort's counters: s_cur is free then)
  static_assert(SUB_BINS >= (SORT_BLOCK / 64) * 256, "s_cnt overlay");
  const uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  const uint16_t *__restrict__ rect = at<uint16_t>(geom, a.g_rect);
  const uint4 *__restrict__ l1tmp = at<uint4>(bin, a.b_l1tmp);
  u64 *__restrict__ l1a = at<u64>(bin, a.b_l1a);
  u64 *__restrict__ l1b = at<u64>(bin, a.b_l1b);
  uint4 *__restrict__ l1list = at<uint4>(bin, a.b_l1);
  const int tid = threadIdx.x;
  const int nbins = 1 << a.lg, nbuckets = a.gi.NS << a.lg;
  // ---- the workgroup's buckets: b = blockIdx.x, + gridDim.x, ... (at most MAXB of them: see bucket_grid).  Their
  // (size, start, slices) words are fetched at once.
  __shared__ uint32_t s_bn[MAXB], s_bbase[MAXB], s_bj[MAXB];
  BinTrace tr(3);
  if (tid < MAXB) {
    const uint32_t b = blockIdx.x + (uint32_t)tid * gridDim.x;
    s_bn[tid] = b < (uint32_t)nbuckets ? bk[BK_TOT + b] : 0u;
    s_bbase[tid] = b < (uint32_t)nbuckets ? bk[BK_START + b] : 0u;
    s_bj[tid] = b < (uint32_t)nbuckets ? bk[BK_BINJ + b] : 0u;
  }
  const uint32_t map_lo = bk[BK_KMIN], map_shift = bk[BK_SHIFT];
  __syncthreads();
  uint4 mine[PER];
  tr.mark();
  for (int k = 0; k < MAXB; ++k) {
    const uint32_t b = blockIdx.x + (uint32_t)k * gridDim.x;
    if (b >= (uint32_t)nbuckets) break;
    const uint32_t n = s_bn[k], base = s_bbase[k];
    if (n == 0u || (size_t)base + n > a.l1cap) continue;  // (the second cannot happen while the instance capacity holds)
    if (n > (uint32_t)BIN_CAP && s_bj[k] != 0u) continue;  // cut into slices: sorted below
    const SubMap sm = sub_bin_map(map_lo, map_shift, b & (uint32_t)(nbins - 1));
    bool lds = n <= (uint32_t)BIN_CAP;
    if (lds) {
      // the thread's entries, every load issued before the first use.  (Requesting bucket k + 1's entries before
      // bucket k is sorted was tried: the second register set cost a third of the occupancy and more than the overlap
      // returned.)
#pragma unroll
      for (int q = 0; q < PER; ++q)
        if ((uint32_t)q * SORT_BLOCK < n) mine[q] = l1tmp[base + min((uint32_t)q * SORT_BLOCK + tid, n - 1)];
      lds_barrier();  // (the previous bucket's LDS has been consumed)
#pragma unroll
      for (int u = 0; u < SUB_OWN; ++u) s_cur[tid * SUB_OWN + u] = 0u;
      if (tid == 0) s_big = 0u;
      lds_barrier();
#pragma unroll
      for (int q = 0; q < PER; ++q)
        if ((uint32_t)q * SORT_BLOCK + tid < n) atomicAdd(&s_cur[sub_bin(mine[q].x, sm)], 1u);
      lds_barrier();
      lds = !scan_sub_bins<true>(s_cur, s_start, s_run, &s_big);  // (workgroup-uniform)
    }
    if (!lds) {  // hundreds of entries at nearly one depth, or an oversized bucket the slice list had no room for
      __syncthreads();
      radix_fallback(l1tmp + base, l1a + base, l1b + base, l1list + base, rect, n, s_run, s_cnt);
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((uint32_t)q * SORT_BLOCK + tid < n) {
        const uint32_t p = atomicAdd(&s_cur[sub_bin(mine[q].x, sm)], 1u);
        s_k[p] = ((u64)mine[q].x << 32) | (u64)mine[q].y;
        s_r[p] = make_uint2(mine[q].z, mine[q].w);
      }
    lds_barrier();
    rank_and_store(s_k, s_r, s_start, 0u, n, sm, l1list + base);
    tr.mark();
  }
  tr.mark();
  {
    // ---- slices of oversized buckets: every slice counts the bucket's sub-bins itself, takes the run of sub-bins
    // whose first entry falls into its share of the bucket, and sorts those (<= SLICE_TARGET + SUB_MAX = BIN_CAP entries)
    __syncthreads();
    const uint32_t n_slices = bk[BK_NSLICE];
    for (uint32_t t = blockIdx.x; t < n_slices; t += gridDim.x) {
      const uint32_t code = bk[BK_SLICE + t];
      if (code == 0xffffffffu) continue;
      const uint32_t b = code >> 16, J = (code >> 8) & 255u, j = code & 255u;
      const uint32_t base = bk[BK_START + b], n = bk[BK_TOT + b];
      if ((size_t)base + n > a.l1cap) continue;  // (cannot happen while the instance capacity holds)
      const SubMap sm = sub_bin_map(map_lo, map_shift, b & (uint32_t)(nbins - 1));
      __syncthreads();  // (the previous slice's LDS has been consumed)
#pragma unroll
      for (int u = 0; u < SUB_OWN; ++u) s_cur[tid * SUB_OWN + u] = 0u;
      if (tid == 0) s_big = 0u;
      __syncthreads();
      for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {
        uint32_t kk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = l1tmp[base + min(e0 + q * SORT_BLOCK + tid, n - 1)].x;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (e0 + q * SORT_BLOCK + tid < n) atomicAdd(&s_cur[sub_bin(kk[q], sm)], 1u);
      }
      __syncthreads();
      if (scan_sub_bins<false>(s_cur, s_start, s_run, &s_big)) {  // a sub-bin too large to rank: slice 0 sorts the
        if (j == 0) {                                              // whole bucket the slow way
          __syncthreads();
          radix_fallback(l1tmp + base, l1a + base, l1b + base, l1list + base, rect, n, s_run, s_cnt);
        }
        continue;
      }
      // this slice's sub-bins [f0, f1): those whose first entry lies in [j Tn, (j + 1) Tn) and that are not empty
      const uint32_t Tn = (n + J - 1) / J;
      uint32_t mnf = 0xffffu, mxf = 0u;
#pragma unroll
      for (int u = 0; u < SUB_OWN; ++u) {
        const uint32_t f = (uint32_t)(tid * SUB_OWN + u), my0 = s_start[f];
        if (my0 >= j * Tn && my0 < (j + 1) * Tn && s_start[f + 1] > my0) mnf = min(mnf, f), mxf = max(mxf, f + 1u);
      }
      mnf = wave_min(mnf), mxf = wave_max(mxf);
      if ((tid & 63) == 0) s_run[tid >> 6] = mnf, s_run[4 + (tid >> 6)] = mxf;
      __syncthreads();
      const uint32_t f0 = min(min(s_run[0], s_run[1]), min(s_run[2], s_run[3]));
      const uint32_t f1 = max(max(s_run[4], s_run[5]), max(s_run[6], s_run[7]));
      __syncthreads();
      if (f0 >= f1) continue;  // (an empty share)
      const uint32_t first = s_start[f0];
#pragma unroll
      for (int u = 0; u < SUB_OWN; ++u) s_cur[tid * SUB_OWN + u] = s_start[tid * SUB_OWN + u] - first;  // cursors
      __syncthreads();
      for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {
        uint4 kv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kv[q] = l1tmp[base + min(e0 + q * SORT_BLOCK + tid, n - 1)];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t f = sub_bin(kv[q].x, sm);
          if (e0 + q * SORT_BLOCK + tid < n && f >= f0 && f < f1) {
            const uint32_t p = atomicAdd(&s_cur[f], 1u);
            if (p < (uint32_t)BIN_CAP) s_k[p] = ((u64)kv[q].x << 32) | (u64)kv[q].y, s_r[p] = make_uint2(kv[q].z, kv[q].w);
          }
        }
      }
      __syncthreads();
      const uint32_t m = min(s_start[f1] - first, (uint32_t)BIN_CAP);  // entries of the slice
      rank_and_store(s_k, s_r, s_start, first, m, sm, l1list + base + first);
      tr.mark();
    }
  }
  tr.flush();
}

// ------------------------------------------------------------------------------------ 4. / 5. level 2
// A workgroup (4 waves) works through windows w = blockIdx.x, + gridDim.x, ... of 256 entries each, one wave per 64
// entries; the NEXT window's table words and entries are requested before the current one is worked on.  These
// kernels are INSTRUCTION-ISSUE bound (a few thousand short waves: ~0.6 M wave-instructions per us chip-wide whatever
// the mix -- tools/latency_model.hip, profiles/r03_sq_binning.txt), so the walk is written for instruction count: the
// supertile edge is a template parameter, every entry forms the bit mask of the supertile's tiles its rectangle
// covers once (a row mask times a column pattern), and the per-tile loop is unrolled over the mask bits: one ballot,
// one popcount and one v_writelane (count) or one v_readlane + mbcnt + three stores (fill) per tile.
// Lane j of every wave owns tile j of the supertile.  Count pass: each wave's per-tile counts -> cnt2w[window][wave]
// [tile], the window's totals -> cnt2[window][tile] and, atomically, the tile totals.  Fill pass: a wave starts at the
// tile's start + the counts of the supertile's earlier windows + the counts of the waves before it.
template <int J>
__device__ __forceinline__ uint32_t writelane(uint32_t v, uint32_t sval) {  // v[lane J] = sval (one SGPR operand: J is an immediate)
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sval), "n"(J));
  return v;
}
template <int J, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, N>(f);
  }
}
constexpr int TS_LDS = 4096;  // tiles whose starts a fill workgroup keeps in LDS (1024^2 pixels); beyond: re-walked per window

template <int SSH>
__device__ __forceinline__ u64 tile_mask(bool valid, const uint4 en, int tx0, int ty0) {
  constexpr int ss = 1 << SSH;
  const int x0 = en.z & 0xffff, y0 = en.z >> 16, x1 = en.w & 0xffff, y1 = en.w >> 16;
  const int lx0 = max(x0 - tx0, 0), lx1 = min(x1 - tx0, ss), ly0 = max(y0 - ty0, 0), ly1 = min(y1 - ty0, ss);
  if (!(valid && lx1 > lx0 && ly1 > ly0)) return 0ull;
  const uint32_t rm = (1u << lx1) - (1u << lx0);  // the covered tiles of one row
  if (SSH == 3) {
    const u64 hi = ly1 >= 8 ? ~0ull : ((1ull << (8 * ly1)) - 1ull), lo = (1ull << (8 * ly0)) - 1ull;
    return (u64)rm * ((hi ^ lo) & 0x0101010101010101ull);
  }
  const uint32_t yr = ((1u << (ss * ly1)) - (1u << (ss * ly0))) & (SSH == 2 ? 0x1111u : (SSH == 1 ? 0x5u : 0x1u));
  return (u64)(rm * yr);
}

struct L2Shared {  // (declared once in level2_dispatch: the four supertile-edge instances share it)
  uint32_t c[SEG / 64][64];
  uint32_t tsw[64];  // fill, T > TS_LDS: first slot of the current supertile's tiles
  uint32_t wt[4];
};
template <bool FILL, int SSH>
__device__ __forceinline__ void level2_body(const BinArgs &a, void *geom, void *bin, uint8_t *__restrict__ grad_flags,
                                            uint32_t *__restrict__ totals_out, L2Shared &sh,
                                            uint32_t *s_ts /* fill: first slot of every tile (T <= TS_LDS) */) {
  constexpr int ss = 1 << SSH, ntile = ss * ss;
  uint32_t (&s_c)[SEG / 64][64] = sh.c;
  uint32_t (&s_tsw)[64] = sh.tsw;
  uint32_t (&s_wt)[4] = sh.wt;
  const uint32_t *__restrict__ meta = at<uint32_t>(bin, a.b_meta);
  const uint4 *__restrict__ wintab = reinterpret_cast<const uint4 *>(meta + META_WIN);
  const uint4 *__restrict__ l1list = at<uint4>(bin, a.b_l1);
  uint32_t *__restrict__ cnt2 = at<uint32_t>(bin, a.b_cnt2);
  uint32_t *__restrict__ cnt2w = cnt2 + a.max_windows * 64;
  uint32_t *__restrict__ tile_tot = at<uint32_t>(bin, a.b_totals);
  uint32_t *__restrict__ dkeys = at<uint32_t>(bin, a.b_dkeys);
  uint32_t *__restrict__ vals = at<uint32_t>(bin, a.b_vals);
  const int tiles_x = a.gi.tiles_x, tiles_y = a.gi.tiles_y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  BinTrace tr(FILL ? 5 : 4);
  // the first window's entries and table words do not depend on anything: requested together with the window count
  uint32_t w = blockIdx.x;
  uint4 en_n = (size_t)w * SEG + tid < a.l1cap ? l1list[(size_t)w * SEG + tid] : make_uint4(0u, 0u, 0u, 0u);
  uint4 rec_n = w < a.max_windows ? wintab[w] : make_uint4(0u, 0u, 0u, 0u);
  const uint32_t n_win = meta[META_NWIN];
  // fill: exclusive scan of the tile totals, by every workgroup itself -- thread t owns tiles [t K, (t + 1) K)
  const int K = (a.T + SEG - 1) / SEG;
  const bool ts_lds = a.T <= TS_LDS;
  uint32_t my_first = 0;  // instances before this thread's tiles
  if (FILL) {
    if (blockIdx.x >= n_win && blockIdx.x != 0) return;
    uint32_t sum = 0;
    for (int q = 0; q < K; ++q) sum += tid * K + q < a.T ? tile_tot[tid * K + q] : 0u;
    const uint32_t inc = wave_scan_incl(sum, lane);
    if (lane == 63) s_wt[wave] = inc;
    lds_barrier();
    my_first = inc - sum;
    for (int q = 0; q < wave; ++q) my_first += s_wt[q];
    {
      uint32_t run = my_first;
      for (int q = 0; q < K; ++q) {
        const int t = tid * K + q;
        if (t >= a.T) break;
        if (ts_lds) s_ts[t] = run;
        run += tile_tot[t];
      }
    }
    if (blockIdx.x == 0) {
      // tile ranges clamped to the instance capacity, overflow flag, and the backward's three work counters cleared
      // for the blend forward behind this kernel
      uint32_t *ranges = at<uint32_t>(bin, a.b_ranges), *total = at<uint32_t>(geom, a.g_total);
      uint32_t *work_count = at<uint32_t>(bin, a.b_work);
      if (tid == 0) work_count[0] = 0, work_count[1] = 0, work_count[2] = 0;
      // ... and the bucket totals, so that the chain can run again on the same projection (preprocess clears them too)
      uint32_t *bk = at<uint32_t>(geom, a.g_bk);
      for (int t = tid; t < MAX_BUCKETS; t += SEG) bk[BK_TOT + t] = 0u;
      uint32_t run = my_first;
      int ovf = 0;
      for (int q = 0; q < K; ++q) {
        const int t = tid * K + q;
        if (t >= a.T) break;
        const uint32_t v = tile_tot[t];
        // an empty tile reports (0, 0) like the published identifyTileRanges leaves it
        ranges[2 * t] = v ? min(run, a.R_cap) : 0u, ranges[2 * t + 1] = v ? min(run + v, a.R_cap) : 0u;
        if (run + v > a.R_cap) total[1] = 1, ovf = 1;  // capacity overflow: flagged, never written out of bounds
        run += v;
      }
      ovf = __syncthreads_or(ovf);
      // the render's (R, overflow) for the step-level array (dimo_render_desc.totals_out)
      if (totals_out && tid == 0) totals_out[0] = total[0], totals_out[1] = (uint32_t)(ovf != 0);
    }
  }
  tr.mark();
  for (; w < n_win; w += gridDim.x) {
    tr.mark();
    const uint4 en = en_n;
    const uint4 rec = rec_n;
    {  // the next window of this workgroup
      const uint32_t wn = w + gridDim.x;
      if (wn < n_win) {
        en_n = (size_t)wn * SEG + tid < a.l1cap ? l1list[(size_t)wn * SEG + tid] : make_uint4(0u, 0u, 0u, 0u);
        rec_n = wintab[wn];
      }
    }
    const uint32_t nvalid = rec.x >> 16, w0 = rec.y;
    const int tx0 = (int)(rec.z & 0xffffu), ty0 = (int)(rec.z >> 16);
    // lane j owns tile j of the supertile: its count (count pass) or its next free slot (fill pass)
    const int my_tx = tx0 + (lane & (ss - 1)), my_ty = ty0 + (lane >> SSH);
    const bool my_in = lane < ntile && my_tx < tiles_x && my_ty < tiles_y;
    const u64 mask = tile_mask<SSH>((uint32_t)tid < nvalid, en, tx0, ty0);  // (an unwritten slot may hold anything)
    uint32_t c = 0;
    if (FILL) {
      lds_barrier();  // (s_tsw / s_c of the previous window have been read)
      // the counts of the supertile's earlier windows: wave v sums windows w0 + v, w0 + v + 4, ...
      {
        uint32_t acc = 0;
        for (uint32_t q = w0 + wave; q < w; q += 4 * 8) {
          uint32_t v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = q + 4 * r < w ? cnt2[(size_t)(q + 4 * r) * 64 + lane] : 0u;
#pragma unroll
          for (int r = 0; r < 8; ++r) acc += v[r];
        }
        s_c[wave][lane] = acc;
      }
      if (!ts_lds) {  // the starts of this supertile's tiles, picked out of the scan by the threads that own them
        uint32_t run = my_first;
        for (int q = 0; q < K; ++q) {
          const int t = tid * K + q;
          if (t >= a.T) break;
          const int tx = t % tiles_x - tx0, ty = t / tiles_x - ty0;
          if (tx >= 0 && tx < ss && ty >= 0 && ty < ss) s_tsw[(ty << SSH) + tx] = run;
          run += tile_tot[t];
        }
      }
      lds_barrier();
      if (my_in) {
        c = (ts_lds ? s_ts[my_ty * tiles_x + my_tx] : s_tsw[lane]) + s_c[0][lane] + s_c[1][lane] + s_c[2][lane] + s_c[3][lane];
        for (int v = 0; v < wave; ++v) c += cnt2w[((size_t)w * (SEG / 64) + v) * 64 + lane];
      }
#pragma unroll
      for (int j = 0; j < ntile; ++j) {
        const bool cov = (mask >> j) & 1ull;
        const u64 bal = __ballot(cov);
        if (bal == 0) continue;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
        const uint32_t pos = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cov && pos < a.R_cap) {
          dkeys[pos] = en.y;
          vals[pos] = en.x;
          // the fill pass touches every instance slot [0, R) exactly once: it also clears the backward's "gradient
          // record written" flags (indexed by emission position, the same range) -- a launch of its own before
          if (grad_flags) grad_flags[pos] = 0;
        }
      }
    } else {
      static_for<0, ntile>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const u64 bal = __ballot((mask >> j) & 1ull);
        c = writelane<j>(c, (uint32_t)__popcll(bal));
      });
      cnt2w[((size_t)w * (SEG / 64) + wave) * 64 + lane] = c;
      lds_barrier();  // (the previous window's sums have been read: see the barrier below)
      s_c[wave][lane] = c;
      lds_barrier();
      if (wave == 0) {
        const uint32_t tot = s_c[0][lane] + s_c[1][lane] + s_c[2][lane] + s_c[3][lane];
        cnt2[(size_t)w * 64 + lane] = tot;
        if (my_in && tot) atomicAdd(&tile_tot[my_ty * tiles_x + my_tx], tot);
      }
    }
  }
  if (FILL && blockIdx.x == 0) {
    // ---- the blend forward's dispatch order: tiles by descending list length (a counting sort over 256 classes of
    // 16 entries; the order inside a class is whatever the LDS atomics make of it -- it only decides WHEN a tile's
    // workgroup starts).  The launch otherwise ends with a tail a quarter of its length: every workgroup has started
    // after 145 of 189 us (8 renders) and the last ones are as long as any.
    uint32_t *__restrict__ order = at<uint32_t>(bin, a.b_order);
    uint32_t *const s_hist = s_ts;  // (256 words; the tile starts are not needed any more)
    __syncthreads();
    s_hist[tid] = 0u;
    __syncthreads();
    for (int t = tid; t < a.T; t += SEG) atomicAdd(&s_hist[255u - min(tile_tot[t] >> 4, 255u)], 1u);
    __syncthreads();
    const uint32_t mine = s_hist[tid];
    const uint32_t inc = wave_scan_incl(mine, lane);
    if (lane == 63) s_wt[wave] = inc;
    __syncthreads();
    uint32_t start = inc - mine;
    for (int w2 = 0; w2 < wave; ++w2) start += s_wt[w2];
    s_hist[tid] = start;
    __syncthreads();
    for (int t = tid; t < a.T; t += SEG) order[atomicAdd(&s_hist[255u - min(tile_tot[t] >> 4, 255u)], 1u)] = (uint32_t)t;
  }
  tr.flush();
}

template <bool FILL>
__device__ __forceinline__ void level2_dispatch(const BinArgs &a, void *geom, void *bin, uint8_t *grad_flags,
                                                uint32_t *totals_out) {
  __shared__ L2Shared sh;
  __shared__ uint32_t s_ts[FILL ? TS_LDS : 1];
  switch (a.gi.ss_shift) {  // (uniform)
    case 0: level2_body<FILL, 0>(a, geom, bin, grad_flags, totals_out, sh, s_ts); break;
    case 1: level2_body<FILL, 1>(a, geom, bin, grad_flags, totals_out, sh, s_ts); break;
    case 2: level2_body<FILL, 2>(a, geom, bin, grad_flags, totals_out, sh, s_ts); break;
    default: level2_body<FILL, 3>(a, geom, bin, grad_flags, totals_out, sh, s_ts); break;
  }
}

// ------------------------------------------------------------------------------------ kernel entry points
// Every stage exists as a single-render kernel (the C-ABI calls) and as a batched one whose blockIdx.y selects the
// render of a RenderBatch (the native step executor: one launch per stage for all renders of a range).
__global__ void __launch_bounds__(SORT_BLOCK) level1_count_kernel(BinArgs a, void *geom, void *bin) {
  level1_count_body(a, geom, bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) level1_scatter_kernel(BinArgs a, void *geom, void *bin) {
  level1_scatter_body(a, geom, bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) bucket_sort_kernel(BinArgs a, void *geom, void *bin) {
  bucket_sort_body(a, geom, bin);
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level2_kernel(BinArgs a, void *geom, void *bin) {
  level2_dispatch<FILL>(a, geom, bin, nullptr, nullptr);  // (the C-ABI backward clears its own scratch)
}
__global__ void __launch_bounds__(SORT_BLOCK) level1_count_batched_kernel(BinArgs a, RenderBatch b) {
  level1_count_body(a, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) level1_scatter_batched_kernel(BinArgs a, RenderBatch b) {
  level1_scatter_body(a, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) bucket_sort_batched_kernel(BinArgs a, RenderBatch b) {
  bucket_sort_body(a, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
template <bool FILL>
__global__ void __launch_bounds__(SEG) level2_batched_kernel(BinArgs a, size_t flag_off, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  level2_dispatch<FILL>(a, r.geom, r.bin, FILL ? at<uint8_t>(r.bwd_scratch, flag_off) : nullptr, FILL ? r.totals_out : nullptr);
}

// ------------------------------------------------------------------------------------ host side
static bool make_args(int N, int H, int W, int64_t R_cap, int n_renders, const GeomLayout &G, const BinLayout &B,
                      BinArgs &a) {
  if (!make_bin_grid(H, W, a.gi)) return false;  // more than MAX_SUPER * 64 tiles
  // level-1 workgroups walk `per` preprocess blocks each: their fixed work (reducing the per-block words, the bucket
  // tables) is per workgroup, and the kernels are instruction bound -- about 1024 workgroups per launch
  int per = (int)(((size_t)G.nb * (size_t)(n_renders > 0 ? n_renders : 1) + 1023) / 1024);
  // ... and at most ~256 workgroups per render: every workgroup adds to each bucket it touches with one returning
  // atomic, and the workgroups of a model that was never Morton-sorted touch nearly all of them (391 workgroups x 400
  // buckets on 512 words: 50 us of serialised atomics for one render).  (~128 until the workgroups' lifetimes were
  // looked at: all of a launch's workgroups are resident at once, the launch lasts as long as its slowest one, and
  // with four blocks each the slowest took twice the mean -- two blocks: 7130 against 7040 frames/s, and no worse
  // for an unsorted model, 73.5 against 78.6 us per render for count + scatter + sort.)
  if (per < (G.nb + 255) / 256) per = (G.nb + 255) / 256;
  if (per < G.per) per = G.per;
  if (per > 8) per = 8 > G.per ? 8 : G.per;
  a.N = N, a.nb = G.nb, a.per = per, a.nwg1 = (G.nb + per - 1) / per, a.lg = depth_bins_log2(N, a.gi.NS), a.T = B.T;
  a.R_cap = (uint32_t)B.cap;
  a.l1cap = B.l1cap, a.max_windows = B.max_windows;
  a.g_total = G.total, a.g_rect = G.rect, a.g_tiles = G.tiles, a.g_offsets = G.offsets, a.g_sums = G.block_sums;
  a.g_key32 = G.key32, a.g_bk = G.bk, a.g_wgbase = G.wgbase;
  a.b_meta = B.meta, a.b_l1tmp = B.l1tmp, a.b_l1a = B.l1a, a.b_l1b = B.l1b, a.b_l1 = B.l1list, a.b_cnt2 = B.cnt2;
  a.b_totals = B.totals, a.b_ranges = B.ranges, a.b_work = B.work, a.b_dkeys = B.dkeys, a.b_vals = B.vals_b;
  a.b_order = B.order;
  return true;
}

// persistent workgroups over the level-2 windows (their number, ~ entries / 256 + supertiles, is only known on the
// device): sized so that all the renders of a launch fit the chip in one round (2048 workgroups of 256 threads),
// a workgroup then walks ~5 windows at the benchmark configuration
// bucket_sort: persistent workgroups, each walks <= MAXB buckets; sized so that the renders of a launch fit the chip
// in about one round
static unsigned bucket_grid(unsigned nbuckets, int n_renders) {
  const unsigned least = (nbuckets + MAXB - 1) / MAXB;  // (nbuckets <= 2048: at least 256 then)
  const unsigned room = (unsigned)(768 / (n_renders > 0 ? n_renders : 1));  // (42 KB of LDS: 3 workgroups per CU)
  unsigned g = room < nbuckets ? room : nbuckets;
  if (g < least) g = least;
  return g < 32u ? (nbuckets < 32u ? (nbuckets ? nbuckets : 1u) : 32u) : g;
}
static unsigned level2_grid(int N, int n_renders, int NS) {
  const unsigned want = (unsigned)((4 * (size_t)N) / SEG + NS + 1);  // ~ one window each, were there room
  const unsigned room = (unsigned)(2048 / (n_renders > 0 ? n_renders : 1));
  return want < room ? want : (room > 64u ? room : 64u);
}

int bin_instances(int N, int H, int W, int64_t R_cap, const void *geom_c, void *bin, hipStream_t stream) {
  GeomLayout G(N);
  BinLayout B(R_cap, H, W);
  BinArgs a;
  if (!make_args(N, H, W, R_cap, 1, G, B, a)) return DIMO_E_ARG;
  void *geom = const_cast<void *>(geom_c);  // offsets, bucket tables and the overflow flag live in the geometry workspace
  const unsigned nbuckets = (unsigned)(a.gi.NS << a.lg), g2 = level2_grid(N, 1, a.gi.NS);
  {
    ScopedTimer tm(T_SCAN, stream);
    hipLaunchKernelGGL(level1_count_kernel, dim3(a.nwg1), dim3(SORT_BLOCK), 0, stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_EMIT, stream);
    hipLaunchKernelGGL(level1_scatter_kernel, dim3(a.nwg1), dim3(SORT_BLOCK), 0, stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_SORT, stream);
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(bucket_grid(nbuckets, 1)), dim3(SORT_BLOCK), 0, stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(level2_kernel<false>, dim3(g2), dim3(SEG), 0, stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    hipLaunchKernelGGL(level2_kernel<true>, dim3(g2), dim3(SEG), 0, stream, a, geom, bin);
  }
  return check_launch();
}

int bin_instances_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W);
  BinArgs a;
  if (!make_args(c.N, c.H, c.W, c.R_cap, n, G, B, a)) return DIMO_E_ARG;
  if (c.bin_bytes < B.bytes || c.geom_bytes < G.bytes) return DIMO_E_WORKSPACE;
  if (c.bwd_scratch_bytes < align_up(B.cap * sizeof(SplatGrad)) + align_up(B.cap)) return DIMO_E_WORKSPACE;
  for (int i = 0; i < n; ++i)
    if (!b.r[i].bwd_scratch) return DIMO_E_ARG;  // the fill pass clears the backward's record flags
  const unsigned nbuckets = (unsigned)(a.gi.NS << a.lg), g2 = level2_grid(c.N, n, a.gi.NS);
  {
    ScopedTimer tm(T_SCAN, stream);
    hipLaunchKernelGGL(level1_count_batched_kernel, dim3(a.nwg1, n), dim3(SORT_BLOCK), 0, stream, a, b);
  }
  {
    ScopedTimer tm(T_EMIT, stream);
    hipLaunchKernelGGL(level1_scatter_batched_kernel, dim3(a.nwg1, n), dim3(SORT_BLOCK), 0, stream, a, b);
  }
  {
    ScopedTimer tm(T_SORT, stream);
    hipLaunchKernelGGL(bucket_sort_batched_kernel, dim3(bucket_grid(nbuckets, n), n), dim3(SORT_BLOCK), 0, stream, a, b);
  }
  {
    ScopedTimer tm(T_RANGES, stream);
    hipLaunchKernelGGL(level2_batched_kernel<false>, dim3(g2, n), dim3(SEG), 0, stream, a, (size_t)0, b);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    // (flags of the backward's scratch: [records: cap x 64 B][flags: cap x 1 B], see blend.hip)
    hipLaunchKernelGGL(level2_batched_kernel<true>, dim3(g2, n), dim3(SEG), 0, stream, a,
                       align_up(B.cap * sizeof(SplatGrad)), b);
  }
  return check_launch();
}

}  // namespace dimo

// Diagnostic: per-workgroup phase trace of the binning kernels (see BinTrace).  buffer = device memory for `capacity`
// records of 32 x u64, or null to switch the trace off; returns the number of records written since the last call.
extern "C" int64_t dimo_debug_bin_trace(void *buffer, int64_t capacity) {
#ifndef DIMO_BIN_TRACE
  if (buffer) return DIMO_E_ARG;  // not compiled in
#endif
  using namespace dimo;
  unsigned int n = 0, zero = 0, cap = (unsigned int)(capacity > 0 ? capacity : 0);
  unsigned long long *p = reinterpret_cast<unsigned long long *>(buffer);
  if (hipDeviceSynchronize() != hipSuccess) return DIMO_E_LAUNCH;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_bin_trace_n), sizeof(n)) != hipSuccess) return DIMO_E_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_bin_trace_n), &zero, sizeof(zero)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_bin_trace_cap), &cap, sizeof(cap)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_bin_trace), &p, sizeof(p)) != hipSuccess)
    return DIMO_E_LAUNCH;
  return (int64_t)n;
}
