// Tile binning: produces, per tile, the list of Gaussians that touch it in (depth, id) order -- bit for bit the ids a
// stable radix sort of the (tile | fp32 depth bits) keys of all (Gaussian, tile) instances yields (the published
// rasterizer's cub::DeviceRadixSort) -- without ever sorting the instances, in THREE launches (five until round 5):
//   1. level1      : a Gaussian has one LEVEL-1 ENTRY per supertile (SS x SS tiles) it touches; an entry belongs to
//                    the BUCKET (supertile, coarse depth bin).  A workgroup walks its 256 x `per` Gaussians ONCE: it
//                    counts its entries per bucket in LDS and every entry learns its rank inside the workgroup's share of
//                    its bucket (kept as a 32-bit code in LDS); ONE returning atomic per (workgroup, non-empty bucket)
//                    adds the share to the bucket total -- what it returns is the share's first PLACE in the bucket --
//                    and the 16-byte entries go to region(bucket) + place: every bucket owns a fixed region of
//                    BUCKET_REGION slots of the unsorted level-1 array, so no bucket start has to be known (which is
//                    what made count and scatter two launches) and a bucket is still one contiguous run for the sort.
//                    Places beyond a region (rare) go to an overflow area, with a record in the bucket's list.  Also:
//                    offsets (the inclusive scan of tiles_touched), R, and the depth-bin map -- every workgroup reduces
//                    the per-block words of the projection kernel (a few hundred) itself.  The render's LAST workgroup
//                    (a ticket) reads the bucket totals and lays the SORTED array out: bucket starts (supertile by
//                    supertile, a supertile's buckets by depth bin, every bucket rounded up to whole GROUPS of 64
//                    entries), the slices of oversized buckets, the rows of per-tile totals, bucket_sort's work list.
//   2. bucket_sort : persistent workgroups take the work list item by item (tickets); an item = a bucket, sorted in
//                    LDS by the 64-bit word (depth bits << 32 | id): a counting pass over 1024 sub-bins of the bucket's
//                    key range, then every entry ranks itself inside its sub-bin (all words are distinct because the
//                    ids are).  An entry that has its sorted place also has its group, and adds one to an LDS counter
//                    per tile of the supertile its rectangle covers: the first half of level 2 -- per group of 64 sorted
//                    entries and per tile, how many entries of the bucket's EARLIER groups cover the tile (a row per
//                    group), the bucket's totals per tile (one row per bucket: the UNIT row), and the tile totals
//                    (atomics).  The sorted entry is (id, tile mask).
//   3. level2_fill : one WAVE per group, no barriers: first slot of tile j for the group = tile start (every workgroup
//                    scans the tile totals itself) + the unit rows of the supertile's earlier buckets + the group's own
//                    row; an ORDERED filter (ballot / popcount ranks) then writes the Gaussian id per instance as long
//                    contiguous runs.  Also: the tile ranges, the overflow flag, the blend forward's dispatch order, and
//                    the backward's record flags cleared with wide stores.
// An ordered filter keeps the input order, so every tile list comes out in (depth, id) order.  A frame has ~10x fewer
// Gaussians than instances (1e5 vs 1e6 at the benchmark configuration).  Earlier versions: 6 radix passes over the
// instances; 2 passes + a per-tile LDS sort; (round 2) a depth sort of the Gaussians + two levels of ordered filters in
// twelve launches; (rounds 3-4) count / scatter / sort / level-2 count / level-2 fill in five, whose cost was their
// instruction count (a kernel of a few thousand short waves runs at ~0.6 M wave-instructions per us chip-wide whatever
// the mix: tools/latency_model.hip) -- the second walk over the Gaussians, the bucket-start scan of every scatter
// workgroup, the second read of the sorted entries, two barriers and a prefix loop per 256-entry window in the fill,
// and three stores per instance where one is needed are what this version removes.  The depth bits of an instance are
// NOT stored per instance (they are its Gaussian's: geom key32 / the splat record's depth); its tile is the list the
// instance sits in (`ranges`).
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
// itself against the other entries of its sub-bin: the cost is the sum of the squared sub-bin sizes).  Depth bin 0
// also holds every key below the binned range (outliers in front of the scene): those share sub-bin 0 and the bin's
// own keys get the upper half of the sub-bins -- with ONE linear map from the smallest key a single floater would
// squeeze the bin's bulk into a few fat sub-bins.  Keys past the last bin share the last sub-bin.
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

constexpr int GRP = 64;  // sorted level-1 entries per level-2 group: one wave
// (the small words of the bin workspace, `meta`: common.hpp)

struct BinArgs {
  int N, nb, per, nwg1, lg, T;  // Gaussians, preprocess blocks, blocks per level-1 workgroup, level-1 workgroups,
                                // log2 depth bins, tiles
  int sort_grid;                // workgroups per render of the bucket_sort launch
  uint32_t R_cap;
  BinGrid gi;
  size_t l1cap;
  // byte offsets into the geometry (g_) and bin (b_) workspaces
  size_t g_total, g_rect, g_tiles, g_offsets, g_sums, g_key32, g_bk, g_segs, g_work, g_wgob, g_hit;
  size_t b_order;
  size_t b_meta, b_l1tmp, b_l1a, b_l1b, b_l1, b_grpbase, b_grpinfo, b_cntu, b_totals, b_ranges, b_work, b_vals;
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
// exclusive prefix of v over the 256 threads of the workgroup, and the total; s4: four words of LDS (free to reuse
// after the call's second barrier has been passed by everybody, i.e. behind the caller's next barrier)
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *s4, uint32_t &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_scan_incl(v, lane);
  lds_barrier();  // (s4 of an earlier call has been read)
  if (lane == 63) s4[wave] = inc;
  lds_barrier();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += s4[w];
  total = s4[0] + s4[1] + s4[2] + s4[3];
  return base + inc - v;
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
  // (a mark that says what ended: the clock's two low bits -- 40 ns -- carry `kind`)
  __device__ __forceinline__ void mark(unsigned kind) {
    if (n < 32) t[n++] = (__builtin_amdgcn_s_memrealtime() & ~3ull) | (kind & 3u);
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
  __device__ __forceinline__ void mark(unsigned) {}
  __device__ __forceinline__ void flush() {}
#endif
};

// The level-1 entries of a wave's 64 Gaussians, walked in GROUPS of lanes that name the same bucket: a wave's
// Gaussians are Morton neighbours, so most of its entries share a handful of buckets -- one LDS atomic per entry was
// a 64-way same-address conflict per instruction.  Lane `lane` has its entries in the supertiles
// (sx0 + e % nx, sy0 + e / nx), depth bin db; iteration `it` of the walk is every lane's entry number `it`.
// f(bucket, lanes of the group, leader lane, this lane is in it, it); direct(bucket, this lane is left over, it).
template <class F, class G>
__device__ __forceinline__ void for_each_group(bool has, const uint2 rc, uint32_t db, const BinGrid &gi, int lg, int lane,
                                               F f, G direct) {
  const int x0 = rc.x & 0xffff, y0 = rc.x >> 16, x1 = rc.y & 0xffff, y1 = rc.y >> 16;
  const int sx0 = x0 >> gi.ss_shift, sy0 = y0 >> gi.ss_shift;
  const int nx = has ? ((x1 - 1) >> gi.ss_shift) - sx0 + 1 : 0, ny = has ? ((y1 - 1) >> gi.ss_shift) - sy0 + 1 : 0;
  int cx = 0, cy = 0;
  for (int it = 0;; ++it) {
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
      f(bl, m, leader, mine, it);
      todo &= ~m;
    }
    if (todo) direct(bucket, (todo >> lane) & 1ull, it);
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


// ------------------------------------------------------------------------------------ the sorted array's layout
constexpr int BIN_CAP = 2048;      // entries a workgroup sorts in LDS (16 KB)
constexpr int SUB_MAX = 512;       // largest sub-bin ranked quadratically
// A bucket above BIN_CAP (thousands of Gaussians of one supertile in one depth bin) is cut into SLICES of
// ~SLICE_TARGET entries along its sub-bins, sorted by different workgroups of the bucket_sort launch; only a SUB-bin
// above SUB_MAX (hundreds of Gaussians at nearly ONE depth) sends its bucket to eight stable byte passes of a single
// workgroup through global memory: slow, correct, rare.  The unsorted entries are read-only in bucket_sort, so every
// slice of a bucket sees the same counts and takes the same decisions.  Every slice's sorted entries start at a whole
// group: a bucket of n entries cut into J slices reserves n + 64 J slots (rounded up to a group), and its last slice
// fills what is left with empty entries.
constexpr int SLICE_TARGET = BIN_CAP - SUB_MAX;
constexpr int MAX_UNITS = MAX_BUCKETS + MAX_SLICES;  // rows of per-tile totals: one per bucket, or per slice of a cut bucket
static_assert(MAX_UNITS < 4096 && MAX_SUPER <= 256, "a group's word: unit and first unit in 12 bits each, supertile in 8");
__device__ __forceinline__ uint32_t pad_grp(uint32_t n) { return (n + GRP - 1) / GRP * GRP; }

// Run by ONE workgroup per render (level 1's last): where every bucket's sorted entries start (supertile by
// supertile, a supertile's buckets by depth bin, every bucket rounded up to whole groups), which buckets are cut into
// slices, and the rows (units) of their per-tile totals -- as bucket_sort's WORK LIST, which its workgroups take item
// by item (the first `sort_grid` items one each, the rest through a ticket counter): the slices first (the longest
// jobs), then the buckets that are sorted whole, largest size class first.  An item is two 16-byte words:
// (first slot, entries, overflow records | slices << 16, unit | first unit of the supertile << 12 | supertile << 24) of its
// bucket and (bucket | slice << 16 | (a slice) << 31, 0, 0, 0).
// s_tab: MAX_BUCKETS words of LDS, s4: four.
__device__ __forceinline__ void layout_buckets(const BinArgs &a, void *geom, void *bin, uint32_t *s_tab, uint32_t *s4) {
  constexpr int OWN = MAX_BUCKETS / SORT_BLOCK;
  uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  u64 *bk_tot = reinterpret_cast<u64 *>(bk + BK_TOT);
  uint4 *__restrict__ work = at<uint4>(geom, a.g_work);
  const int tid = threadIdx.x, nbuckets = a.gi.NS << a.lg;
  constexpr int CLASSES = 32;  // size classes of 64 entries, the last one open
  __shared__ uint32_t s_class[CLASSES];
  if (tid < CLASSES) s_class[tid] = 0u;
  uint32_t tot[OWN], nsg[OWN], want[OWN], wsum = 0;
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
    // The totals were summed by returning atomics of workgroups all over the chip, every one of them complete before
    // its workgroup drew the ticket that made this one the last.  They are read with an atomic read-modify-write
    // themselves (+ 0): performed where the adds were performed -- at the memory side, behind them in the word's own
    // modification order -- instead of a load that has to be ASSUMED to bypass this XCD's L2 (round-5 advice; a
    // device-scope release / acquire pair on the ticket would be the textbook form and writes the L2 back on this
    // chip: 92 us per launch measured).  Two returning atomics per thread of ONE workgroup per render.
    const u64 w = tid * OWN + u < nbuckets ? atomicAdd(&bk_tot[tid * OWN + u], (u64)0) : 0ull;
    tot[u] = (uint32_t)w, nsg[u] = (uint32_t)(w >> 32);
    const uint32_t J = tot[u] > (uint32_t)BIN_CAP ? (tot[u] + SLICE_TARGET - 1) / SLICE_TARGET : 0u;
    want[u] = J <= 255u ? J : 0u;  // (more: the byte passes)
    wsum += want[u];
  }
  uint32_t wall;
  uint32_t wrun = block_scan_excl(wsum, s4, wall);
  // a bucket gets its slices while the list has room (in bucket order)
  uint32_t J[OWN], reg[OWN], rsum = 0, usum = 0;
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
    J[u] = want[u] && wrun + want[u] <= (uint32_t)MAX_SLICES ? want[u] : 0u;
    wrun += want[u];
    reg[u] = tot[u] ? pad_grp(tot[u] + (uint32_t)GRP * J[u]) : 0u;
    rsum += reg[u], usum += tot[u] ? max(J[u], 1u) : 0u;
  }
  uint32_t rall, uall;
  uint32_t start = block_scan_excl(rsum, s4, rall);
  uint32_t unit = block_scan_excl(usum, s4, uall);
#pragma unroll
  for (int u = 0; u < OWN; ++u) s_tab[tid * OWN + u] = unit, unit += tot[u] ? max(J[u], 1u) : 0u;
  unit -= usum;
  // (the size classes of the buckets sorted whole: counted, largest first)
#pragma unroll
  for (int u = 0; u < OWN; ++u)
    if (tot[u] && !J[u]) atomicAdd(&s_class[CLASSES - 1 - min(tot[u] >> 6, (uint32_t)CLASSES - 1u)], 1u);
  lds_barrier();
  uint32_t n_sl_all;
  {
    uint32_t jsum = 0;
#pragma unroll
    for (int u = 0; u < OWN; ++u) jsum += J[u];
    const uint32_t jex = block_scan_excl(jsum, s4, n_sl_all);
    wrun = jex;  // (= the want prefix for every accepted bucket: the accepted ones are a prefix of the cut ones)
  }
  {  // class cursors: behind the slices
    const uint32_t c = tid < CLASSES ? s_class[tid] : 0u;
    uint32_t call;
    const uint32_t cex = block_scan_excl(c, s4, call);
    if (tid < CLASSES) s_class[tid] = n_sl_all + cex;
    if (tid == 0) bk[BK_NITEMS] = n_sl_all + call;
    // the ticket counters (item i belongs to queue i % WORK_QUEUES): behind the items the workgroups take untold
    if (tid < WORK_QUEUES) bk[BK_WORK + tid] = (uint32_t)(a.sort_grid > tid ? (a.sort_grid - tid + WORK_QUEUES - 1) / WORK_QUEUES : 0);
  }
  lds_barrier();
#pragma unroll
  for (int u = 0; u < OWN; ++u) {
    const uint32_t b = (uint32_t)(tid * OWN + u);
    if (tot[u]) {
      const uint4 bi = make_uint4(start, tot[u], nsg[u] | (J[u] << 16), unit | (s_tab[(b >> a.lg) << a.lg] << 12) | ((b >> a.lg) << 24));
      for (uint32_t j = 0; j < J[u]; ++j)
        work[2 * (wrun + j)] = bi, work[2 * (wrun + j) + 1] = make_uint4(b | (j << 16) | 0x80000000u, 0u, 0u, 0u);
      if (!J[u]) {
        const uint32_t i = atomicAdd(&s_class[CLASSES - 1 - min(tot[u] >> 6, (uint32_t)CLASSES - 1u)], 1u);
        work[2 * i] = bi, work[2 * i + 1] = make_uint4(b, 0u, 0u, 0u);
      }
    }
    wrun += J[u], start += reg[u], unit += tot[u] ? max(J[u], 1u) : 0u;
  }
  if (tid == 0) {
    // the groups the fill walks: those of the buckets that fit the sorted array whole (all of them unless the
    // instance capacity overflowed; the buckets are laid out in order, so the ones that fit are a prefix)
    uint32_t *meta = at<uint32_t>(bin, a.b_meta);
    meta[META_NGRP] = (uint32_t)(min((size_t)rall, a.l1cap) / GRP);
  }
}

// ------------------------------------------------------------------------------------ 1. level 1
// LDS: two counters per bucket (16 KB), the list of big Gaussians (4 KB) and, dynamically sized, one 32-bit code
// (bucket << 16 | rank in the workgroup's share of the bucket) per (block, entry number, thread): per x 4 KB.
__device__ __forceinline__ void level1_body(const BinArgs &a, void *geom, void *bin) {
  __shared__ uint32_t s_hist[MAX_BUCKETS];  // entries of the lanes' walk per bucket; then the place of the workgroup's first entry in the bucket
  __shared__ uint32_t s_bigc[MAX_BUCKETS];  // entries of the big Gaussians per bucket; then their cursor (a place in the bucket)
  __shared__ uint32_t s_w[4][9];
  __shared__ uint32_t s_wt[4];
  __shared__ uint32_t s_anybig;
  __shared__ BigList s_big;
  HIP_DYNAMIC_SHARED(uint32_t, s_code)
  const uint32_t *__restrict__ tiles = at<uint32_t>(geom, a.g_tiles);
  const uint32_t *__restrict__ key32 = at<uint32_t>(geom, a.g_key32);
  const uint16_t *__restrict__ rect = at<uint16_t>(geom, a.g_rect);
  const uint32_t *__restrict__ sums = at<uint32_t>(geom, a.g_sums);
  uint32_t *__restrict__ offsets = at<uint32_t>(geom, a.g_offsets);
  uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  u64 *__restrict__ bk_tot = reinterpret_cast<u64 *>(bk + BK_TOT);
  uint4 *__restrict__ segs = at<uint4>(geom, a.g_segs);
  uint4 *__restrict__ l1tmp = at<uint4>(bin, a.b_l1tmp);
  uint4 *__restrict__ l1ovf = l1tmp + (size_t)(a.gi.NS << a.lg) * BUCKET_REGION;  // (entries beyond a bucket's region)
  // (a bucket that outgrows its region, rare: overflow slot of place p = wgob[bucket] + p -- this workgroup's row of a
  // table in global memory; LDS is what decides how many of these workgroups a CU holds)
  uint32_t *__restrict__ wgob = at<uint32_t>(geom, a.g_wgob) + (size_t)blockIdx.x * MAX_BUCKETS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = (int)blockIdx.x * a.per, c1 = min(a.nb, c0 + a.per);  // this workgroup's preprocess blocks
  const int stride = a.nb + 1;
  const int nbuckets = a.gi.NS << a.lg;
  const u64 lt = (1ull << lane) - 1ull;
  BinTrace tr(1);
  // the first block's Gaussians, requested before anything else (one memory round trip with the reductions below)
  const int i0 = c0 * PRE_BLOCK + tid;
  uint32_t v0 = i0 < a.N ? tiles[i0] : 0u;
  uint32_t k0 = i0 < a.N ? key32[i0] : 0u;
  uint2 r0 = i0 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i0) : make_uint2(0u, 0u);
  for (int j = tid; j < nbuckets; j += SORT_BLOCK) s_hist[j] = 0u, s_bigc[j] = 0u;
  if (tid == 0) s_anybig = 0u;
  // ---- the per-block words of preprocess, reduced by every workgroup itself: instances and entries before this
  // workgroup's blocks and in total, and the depth-bin map.  The binned range is NOT [min, max] of the keys: a few
  // floaters far behind (or in front of) the scene would stretch it until the scene's bulk shares a handful of bins.  A
  // block's MIN ignores a far outlier inside it and its MAX a near one, so A = the largest block minimum and B = the
  // smallest block maximum bracket the bulk whatever the order of the Gaussians (Morton order: A ~ far end, B ~ near
  // end; random order: the other way round); the range is [min(A, B), max(A, B)] widened by a quarter on both sides,
  // inside [min, max].  Keys outside it fall into the first / last bin (whose sub-bins start at the smallest key).
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
      // (more level-1 entries than the unsorted array holds can only come with more instances than the capacity)
      total[0] = tot_t, total[1] = (size_t)tot_e > a.l1cap ? 1u : 0u, total[2] = tot_e, total[3] = 0u;
    }
    uint32_t *tile_tot = at<uint32_t>(bin, a.b_totals);  // summed atomically by bucket_sort
    for (int t = tid; t < a.T; t += SORT_BLOCK) tile_tot[t] = 0u;
  }
  tr.mark();
  // ---- one walk: offsets; this workgroup's entries per bucket; every entry's rank among them
  uint32_t carry = pre_t;
  uint32_t vn = v0, kn = k0;
  uint2 rn = r0;
  for (int c = c0; c < c1; ++c) {
    const int i = c * PRE_BLOCK + tid;
    uint32_t *code = s_code + (size_t)(c - c0) * BIG_ENTRIES * SORT_BLOCK + tid;
    v0 = vn, k0 = kn, r0 = rn;
    if (c + 1 < c1) {  // the next block's Gaussians, requested before this block is worked on
      const int i2 = i + PRE_BLOCK;
      vn = i2 < a.N ? tiles[i2] : 0u;
      kn = i2 < a.N ? key32[i2] : 0u;
      rn = i2 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i2) : make_uint2(0u, 0u);
    }
    const uint32_t inc = wave_scan_incl(v0, lane);
    lds_barrier();  // (s_wt / the list of the previous block have been read; first round: the counters are cleared)
    if (lane == 63) s_wt[wave] = inc;
    if (tid == 0) s_big.n = 0u;
    lds_barrier();
    uint32_t off = carry;
    for (int w = 0; w < wave; ++w) off += s_wt[w];
    carry += s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
    if (i < a.N) offsets[i] = off + inc;
    const bool big = v0 != 0u && rect_entries(r0, a.gi) > BIG_ENTRIES;
    if (big) s_big.push((uint32_t)i, k0, r0), atomicOr(&s_anybig, 1u << (c - c0));  // (which blocks have big Gaussians)
    for_each_group(v0 != 0u && !big, r0, depth_bin(k0, lo, shift, nbins), a.gi, a.lg, lane,
                   [&](uint32_t bl, u64 m, int leader, bool mine, int it) {
                     uint32_t first = 0;  // the group's ranks: one returning LDS atomic by its leader
                     if (lane == leader) first = atomicAdd(&s_hist[bl], (uint32_t)__popcll(m));
                     first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                     if (mine) code[it * SORT_BLOCK] = (bl << 16) | (first + (uint32_t)__popcll(m & lt));
                   },
                   [&](uint32_t bucket, bool mine, int it) {
                     if (mine) code[it * SORT_BLOCK] = (bucket << 16) | atomicAdd(&s_hist[bucket], 1u);
                   });
    lds_barrier();
    for_each_big(s_big, lo, shift, nbins, a.gi, a.lg,
                 [&](uint32_t bucket, uint32_t, uint32_t, uint2) { atomicAdd(&s_bigc[bucket], 1u); });
  }
  __syncthreads();
  tr.mark();
  // ---- every non-empty bucket of the workgroup (thread t owns buckets [8 t, 8 t + 8)): ONE returning atomic on the
  // bucket total reserves the places [share, share + entries) of the bucket for the workgroup.  A bucket's entries live
  // in ITS OWN fixed region of the unsorted array (BUCKET_REGION slots: bucket_sort reads a bucket as one contiguous
  // run and no bucket's start has to be known here).  Places beyond the region -- a bucket of more than 4096 entries:
  // thousands of Gaussians of one supertile in one depth bin -- go to a shared overflow area: a slot range from its
  // cursor, and a record (first overflow slot, first place, places) in the bucket's list.
  const uint32_t big_blocks = s_anybig;
  {
    // (thread t owns the buckets t, t + 256, ...: at the benchmark's 512 buckets every thread has two.  With eight
    // CONSECUTIVE buckets per thread one wave held them all, and its atomics -- each behind a branch of its own -- went
    // out one round trip after the other: the phase took 6.7 us on average and 30 in the slowest workgroup.  All of a
    // thread's atomics are issued before the first result is used.)
    constexpr int OWN = MAX_BUCKETS / SORT_BLOCK;
    uint32_t cn[OWN], tt[OWN];
    u64 old[OWN];
#pragma unroll
    for (int u = 0; u < OWN; ++u) {
      const uint32_t b = (uint32_t)(tid + u * SORT_BLOCK);
      const bool in = b < (uint32_t)nbuckets;
      cn[u] = in ? s_hist[b] : 0u;
      tt[u] = cn[u] + (in ? s_bigc[b] : 0u);
    }
#pragma unroll
    for (int u = 0; u < OWN; ++u) old[u] = tt[u] ? atomicAdd(&bk_tot[tid + u * SORT_BLOCK], (u64)tt[u]) : 0ull;
#pragma unroll
    for (int u = 0; u < OWN; ++u) {
      const uint32_t b = (uint32_t)(tid + u * SORT_BLOCK), t = tt[u];
      if (t) {
        const uint32_t share = (uint32_t)old[u];
        s_hist[b] = share, s_bigc[b] = share + cn[u];
        if (share + t > (uint32_t)BUCKET_REGION) {
          const uint32_t place = max(share, (uint32_t)BUCKET_REGION), cnt = share + t - place;
          const uint32_t osrc = atomicAdd(&bk[BK_OVF], cnt);
          const uint32_t slot = (uint32_t)(atomicAdd(&bk_tot[b], 1ull << 32) >> 32);
          if (slot < (uint32_t)a.nwg1) segs[(size_t)b * a.nwg1 + slot] = make_uint4(osrc, place, cnt, 0u);
          wgob[b] = osrc - place;
        }
      }
    }
  }
  // the LAST workgroup of the render to get here (every workgroup's entries are in the bucket totals then) lays the
  // sorted array out for bucket_sort, once its own entries are placed.  (Nobody waits for the ticket before that: the
  // tickets of a render are atomics on ONE word, which the memory side hands out at ~8 per us -- with a barrier
  // behind the draw the launch took 53 us instead of 41.)
  // (No device-wide fence: a release at device scope writes the whole L2 back on this chip -- 92 us per launch
  // measured with one per workgroup.  None is needed: the bucket atomics are performed at the memory side and have
  // RETURNED by the barrier below, the ticket is taken behind it, and the last workgroup reads the totals with
  // device-scope loads; everything else it and the others write is for the next launch.)
  __shared__ uint32_t s_last;
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&bk[BK_DONE], 1u) == gridDim.x - 1u ? 1u : 0u;
  tr.mark();
  // ---- the entries, to their places: the workgroup's first place in the bucket + the entry's rank
  {
    const int i = c0 * PRE_BLOCK + tid;
    vn = i < a.N ? tiles[i] : 0u, kn = i < a.N ? key32[i] : 0u;
    rn = i < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i) : make_uint2(0u, 0u);
  }
  for (int c = c0; c < c1; ++c) {
    const int i = c * PRE_BLOCK + tid;
    const uint32_t *code = s_code + (size_t)(c - c0) * BIG_ENTRIES * SORT_BLOCK + tid;
    const uint32_t v = vn, k = kn;
    const uint2 r = rn;
    if (c + 1 < c1) {  // the next block's Gaussians, requested before this block is worked on
      const int i2 = i + PRE_BLOCK;
      vn = i2 < a.N ? tiles[i2] : 0u, kn = i2 < a.N ? key32[i2] : 0u;
      rn = i2 < a.N ? *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)i2) : make_uint2(0u, 0u);
    }
    const int cnt = v != 0u ? rect_entries(r, a.gi) : 0;
    if (cnt <= BIG_ENTRIES) {
      const uint4 en = make_uint4(k, (uint32_t)i, r.x, r.y);
#pragma unroll
      for (int it = 0; it < BIG_ENTRIES; ++it)
        if (it < cnt) {
          const uint32_t cd = code[it * SORT_BLOCK], bu = cd >> 16, p = s_hist[bu] + (cd & 0xffffu);
          if (p < (uint32_t)BUCKET_REGION) {
            l1tmp[(size_t)bu * BUCKET_REGION + p] = en;
          } else {
            const uint32_t o = wgob[bu] + p;
            if ((size_t)o < a.l1cap) l1ovf[o] = en;
          }
        }
    }
    if ((big_blocks >> (c - c0)) & 1u) {  // (workgroup-uniform: this block has big Gaussians)
      lds_barrier();  // (the list of the previous block has been read)
      if (tid == 0) s_big.n = 0u;
      lds_barrier();
      if (cnt > BIG_ENTRIES) s_big.push((uint32_t)i, k, r);
      lds_barrier();
      for_each_big(s_big, lo, shift, nbins, a.gi, a.lg, [&](uint32_t bucket, uint32_t gi_, uint32_t key, uint2 rc) {
        const uint32_t p = atomicAdd(&s_bigc[bucket], 1u);
        if (p < (uint32_t)BUCKET_REGION) {
          l1tmp[(size_t)bucket * BUCKET_REGION + p] = make_uint4(key, gi_, rc.x, rc.y);
        } else {
          const uint32_t o = wgob[bucket] + p;
          if ((size_t)o < a.l1cap) l1ovf[o] = make_uint4(key, gi_, rc.x, rc.y);
        }
      });
    }
  }
  lds_barrier();  // (thread 0 has written s_last.  Not __syncthreads(): that would wait for this workgroup's stores)
  if (s_last) {   // (workgroup-uniform)
    __syncthreads();
    layout_buckets(a, geom, bin, s_hist, s_wt);
  }
  tr.flush();
}

// ------------------------------------------------------------------------------------ 2. bucket sort + level-2 counts
constexpr int PER = BIN_CAP / SORT_BLOCK;
constexpr int SUB_OWN = SUB_BINS / SORT_BLOCK;  // sub-bins a thread owns in the scans

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

// ---- level 2, first half: the tiles of a supertile that an entry's rectangle covers, as a bit mask (bit j = tile j of
// the supertile, row-major over its SS x SS tiles: a row mask times a column pattern), and per group of 64 entries one
// ballot + popcount + v_writelane per tile: lane j ends up with the number of the group's entries that cover tile j.
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
template <int SSH>
__device__ __forceinline__ u64 tile_mask(const uint32_t rx, const uint32_t ry, int tx0, int ty0) {
  constexpr int ss = 1 << SSH;
  const int x0 = rx & 0xffff, y0 = rx >> 16, x1 = ry & 0xffff, y1 = ry >> 16;
  const int lx0 = max(x0 - tx0, 0), lx1 = min(x1 - tx0, ss), ly0 = max(y0 - ty0, 0), ly1 = min(y1 - ty0, ss);
  if (!(lx1 > lx0 && ly1 > ly0)) return 0ull;  // (an EMPTY entry -- rectangle 0 -- covers nothing)
  const uint32_t rm = (1u << lx1) - (1u << lx0);  // the covered tiles of one row
  if (SSH == 3) {
    const u64 hi = ly1 >= 8 ? ~0ull : ((1ull << (8 * ly1)) - 1ull), lo = (1ull << (8 * ly0)) - 1ull;
    return (u64)rm * ((hi ^ lo) & 0x0101010101010101ull);
  }
  const uint32_t yr = ((1u << (ss * ly1)) - (1u << (ss * ly0))) & (SSH == 2 ? 0x1111u : (SSH == 1 ? 0x5u : 0x1u));
  return (u64)(rm * yr);
}
template <int SSH>
__device__ __forceinline__ uint32_t group_counts(u64 mask) {
  uint32_t c = 0;
  static_for<0, (1 << SSH) * (1 << SSH)>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const u64 bal = __ballot((mask >> j) & 1ull);
    c = writelane<j>(c, (uint32_t)__popcll(bal));
  });
  return c;
}

// A SORTED level-1 entry is what the fill needs of it: the Gaussian id and the mask of the supertile's tiles its
// rectangle covers -- 8 bytes up to 4 x 4 tiles per supertile (a 16-bit mask), 16 bytes for 8 x 8 (the sorted array is
// allocated for 16).  An empty entry (the padding of a run to whole groups) has mask 0.
template <int SSH>
struct Sorted {
  typedef typename std::conditional<(SSH <= 2), uint2, uint4>::type T;
  static __device__ __forceinline__ T make(uint32_t id, u64 mask) {
    if constexpr (SSH <= 2) return make_uint2(id, (uint32_t)mask);
    else return make_uint4(id, 0u, (uint32_t)mask, (uint32_t)(mask >> 32));
  }
  static __device__ __forceinline__ u64 mask(const T &e) {
    if constexpr (SSH <= 2) return (u64)e.y;
    else return (u64)e.z | ((u64)e.w << 32);
  }
  static __device__ __forceinline__ T *at(void *base, size_t i) { return reinterpret_cast<T *>(base) + i; }
  static __device__ __forceinline__ const T *at(const void *base, size_t i) { return reinterpret_cast<const T *>(base) + i; }
};

// what a run of sorted entries (a bucket, or a slice of one) tells level 2
struct Unit {
  uint32_t unit, info;  // row of per-tile totals; the word every group of the run carries: unit | first unit of the supertile << 12 | supertile << 24
  int tx0, ty0;         // first tile of the supertile
  uint32_t *grpbase, *grpinfo, *cntu, *tile_tot;
  int tiles_x, tiles_y;
};
struct UnitShared {
  uint32_t wtot[SORT_BLOCK / 64][64];
};
// The level-2 rows of a run whose sorted entries are in GLOBAL memory (`out`, `len` of them, a whole number of groups;
// the byte-pass fallback and the filler behind a cut bucket's last slice): wave w takes the groups [w gpw, (w + 1) gpw),
// two passes, the totals first.
template <int SSH>
__device__ __forceinline__ void count_run_global(const void *l1list, size_t first, uint32_t len, const Unit &U, UnitShared &sh,
                                                 bool with_unit_row) {
  constexpr int ss = 1 << SSH, ntile = ss * ss;
  const typename Sorted<SSH>::T *out = Sorted<SSH>::at(l1list, first);
  const uint32_t g0 = (uint32_t)(first / GRP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t ng = len / GRP, gpw = (ng + 3) / 4;
  const uint32_t ga = min((uint32_t)wave * gpw, ng), gb = min(ga + gpw, ng);
  uint32_t run = 0;
  for (uint32_t g = ga; g < gb; ++g) run += group_counts<SSH>(Sorted<SSH>::mask(out[(size_t)g * GRP + lane]));
  sh.wtot[wave][lane] = run;
  lds_barrier();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += sh.wtot[w][lane];
  uint32_t acc = base;
  for (uint32_t g = ga; g < gb; ++g) {
    U.grpbase[(size_t)(g0 + g) * GRP + lane] = acc;
    if (lane == 0) U.grpinfo[g0 + g] = U.info;
    acc += group_counts<SSH>(Sorted<SSH>::mask(out[(size_t)g * GRP + lane]));
  }
  if (with_unit_row && wave == SORT_BLOCK / 64 - 1) {
    const uint32_t tot = base + run;
    U.cntu[(size_t)U.unit * GRP + lane] = tot;
    const int my_tx = U.tx0 + (lane & (ss - 1)), my_ty = U.ty0 + (lane >> SSH);
    if (lane < ntile && my_tx < U.tiles_x && my_ty < U.tiles_y && tot) atomicAdd(&U.tile_tot[my_ty * U.tiles_x + my_tx], tot);
  }
  lds_barrier();
}
__device__ __forceinline__ void count_run_global_any(int ssh, const void *l1list, size_t first, uint32_t len, const Unit &U,
                                                     UnitShared &sh, bool with_unit_row) {
  switch (ssh) {
    case 0: count_run_global<0>(l1list, first, len, U, sh, with_unit_row); break;
    case 1: count_run_global<1>(l1list, first, len, U, sh, with_unit_row); break;
    case 2: count_run_global<2>(l1list, first, len, U, sh, with_unit_row); break;
    default: count_run_global<3>(l1list, first, len, U, sh, with_unit_row); break;
  }
}
// `len` empty entries from entry `first` on
__device__ __forceinline__ void clear_sorted(int ssh, void *l1list, size_t first, uint32_t len) {
  if (ssh <= 2) {
    for (uint32_t e = threadIdx.x; e < len; e += SORT_BLOCK) *Sorted<0>::at(l1list, first + e) = make_uint2(0u, 0u);
  } else {
    for (uint32_t e = threadIdx.x; e < len; e += SORT_BLOCK) *Sorted<3>::at(l1list, first + e) = make_uint4(0u, 0u, 0u, 0u);
  }
}
// a row of zeros for a unit without entries of its own (a slice with an empty share, the slices of a bucket that went
// to the byte passes as a whole)
__device__ __forceinline__ void zero_unit_row(const Unit &U) {
  if (threadIdx.x < GRP) U.cntu[(size_t)U.unit * GRP + threadIdx.x] = 0u;
}

// The unsorted entries of a bucket, as one array of n entries: the first BUCKET_REGION of them in the bucket's own
// region, the rest -- rare -- in the overflow area, described by the records (first overflow slot, first place, places)
// the level-1 workgroups left in the bucket's list (in no order: a linear search, uniform trip count).
struct BucketSrc {
  const uint4 *region;   // the bucket's region of the unsorted array
  const uint4 *ovf;      // the overflow area
  size_t ovf_cap;
  const uint32_t *rec;   // [3][nrec] in LDS: overflow slot, first place, places
  uint32_t nrec;
};
// the overflow records of a bucket, to LDS (all threads; the caller's next barrier publishes them).  nrec <= nwg1.
__device__ __forceinline__ void load_overflow_records(uint32_t *s_rec, const uint4 *__restrict__ segs, uint32_t nrec) {
  for (uint32_t q = threadIdx.x; q < nrec; q += SORT_BLOCK) {
    const uint4 r = segs[q];
    s_rec[q] = r.x, s_rec[nrec + q] = r.y, s_rec[2 * nrec + q] = r.z;
  }
}
__device__ __forceinline__ const uint4 *bucket_entry_ptr(const BucketSrc &B, uint32_t e) {
  if (e < (uint32_t)BUCKET_REGION) return B.region + e;
  uint32_t slot = 0xffffffffu;
  for (uint32_t s = 0; s < B.nrec; ++s) {
    const uint32_t d = e - B.rec[B.nrec + s];
    if (d < B.rec[2 * B.nrec + s]) slot = B.rec[s] + d;
  }
  // (no record, or past the overflow area: only when the instance capacity overflowed -- the render is flagged and
  // never used; any entry will do)
  return (size_t)slot < B.ovf_cap ? B.ovf + slot : B.region;
}
__device__ __forceinline__ uint4 bucket_entry(const BucketSrc &B, uint32_t e) { return *bucket_entry_ptr(B, e); }
__device__ __forceinline__ uint32_t bucket_key(const BucketSrc &B, uint32_t e) { return bucket_entry_ptr(B, e)->x; }

// the bucket's words go to la[0, n): eight byte passes a -> b -> ... -> a, then the sorted entries (the rectangle by id:
// a gather, on this path only) and `len - n` empty ones
template <int SSH>
__device__ __forceinline__ void emit_sorted_words(const u64 *la, void *l1list, size_t first, const uint16_t *rect, uint32_t n,
                                                  uint32_t len, const Unit &U) {
  for (uint32_t e = threadIdx.x; e < len; e += SORT_BLOCK) {
    u64 mask = 0;
    uint32_t id = 0;
    if (e < n) {
      id = (uint32_t)la[e];
      const uint2 rc = *reinterpret_cast<const uint2 *>(rect + 4 * (size_t)id);
      mask = tile_mask<SSH>(rc.x, rc.y, U.tx0, U.ty0);
    }
    *Sorted<SSH>::at(l1list, first + e) = Sorted<SSH>::make(id, mask);
  }
}
__device__ __forceinline__ void radix_fallback(const BucketSrc &S, u64 *la, u64 *lb, void *l1list, size_t first, int ssh,
                                               const Unit &U, const uint16_t *rect, uint32_t n, uint32_t len, uint32_t *s_run,
                                               uint32_t (*s_cnt)[256]) {
  for (uint32_t e = threadIdx.x; e < n; e += SORT_BLOCK) {
    const uint4 en = bucket_entry(S, e);
    la[e] = ((u64)en.x << 32) | (u64)en.y;
  }
  for (int byte = 0; byte < 8; ++byte) {
    __threadfence_block();
    __syncthreads();
    wg_radix_pass((byte & 1) ? lb : la, (byte & 1) ? la : lb, n, byte, s_run, s_cnt);
  }
  __threadfence_block();
  __syncthreads();
  switch (ssh) {
    case 0: emit_sorted_words<0>(la, l1list, first, rect, n, len, U); break;
    case 1: emit_sorted_words<1>(la, l1list, first, rect, n, len, U); break;
    case 2: emit_sorted_words<2>(la, l1list, first, rect, n, len, U); break;
    default: emit_sorted_words<3>(la, l1list, first, rect, n, len, U); break;
  }
  __threadfence_block();
  __syncthreads();
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

// Every entry of s_k[0, m) ranks itself inside its sub-bin and goes to out[...]; entries are taken in the order they
// sit in LDS (grouped by sub-bin): the 64 lanes of a wave walk the same one or two sub-bins, so a wave's trip count is
// ITS sub-bins' size, not the largest sub-bin's of the bucket.  An entry knows its GROUP then (its place / 64), and
// adds one to the counter of (group, tile) for every tile of the supertile its rectangle covers -- LDS atomics on
// s_gcnt[group][tile], zeroed by the caller: the first half of level 2, without the sorted entries ever being read
// again.  `len - m` empty entries pad the run to whole groups.  Then, per tile, the prefix over the run's groups: the
// rows, the unit row and the tile totals.
template <int SSH>
__device__ __forceinline__ void rank_store_count(const u64 *s_k, const uint2 *s_r, const uint32_t *s_start, uint32_t origin,
                                                 uint32_t m, uint32_t len, const SubMap &sm, void *l1list, size_t first_out,
                                                 const Unit &U, uint32_t *s_gcnt) {
  constexpr int ss = 1 << SSH, ntile = ss * ss;
  typename Sorted<SSH>::T *__restrict__ out = Sorted<SSH>::at(l1list, first_out);
  const uint32_t g0 = (uint32_t)(first_out / GRP);
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const uint32_t e = (uint32_t)q * SORT_BLOCK + threadIdx.x;
    if (e < m) {
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
      const uint32_t p = lo + r;
      const u64 mask = tile_mask<SSH>(rc.x, rc.y, U.tx0, U.ty0);
      out[p] = Sorted<SSH>::make((uint32_t)c, mask);
      uint32_t *cnt = s_gcnt + (p / GRP) * ntile;
      if (ntile <= 32) {
        for (uint32_t mk = (uint32_t)mask; mk; mk &= mk - 1u) atomicAdd(&cnt[__ffs((int)mk) - 1], 1u);
      } else {
        for (u64 mk = mask; mk; mk &= mk - 1ull) atomicAdd(&cnt[__ffsll((long long)mk) - 1], 1u);
      }
    }
  }
  for (uint32_t e = m + threadIdx.x; e < len; e += SORT_BLOCK) out[e] = Sorted<SSH>::make(0u, 0ull);
  lds_barrier();
  const uint32_t ng = len / GRP;
  if (threadIdx.x < (uint32_t)ntile) {  // thread j: tile j of the supertile
    const int j = (int)threadIdx.x;
    uint32_t run = 0;
    for (uint32_t gb = 0; gb < ng; gb += 8) {  // (eight LDS reads in flight)
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s_gcnt[min(gb + u, ng - 1) * ntile + j];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (gb + u < ng) {
          U.grpbase[(size_t)(g0 + gb + u) * GRP + j] = run;  // (the fill reads lanes [0, ntile) of a row only)
          run += v[u];
        }
    }
    U.cntu[(size_t)U.unit * GRP + j] = run;
    const int my_tx = U.tx0 + (j & (ss - 1)), my_ty = U.ty0 + (j >> SSH);
    if (my_tx < U.tiles_x && my_ty < U.tiles_y && run) atomicAdd(&U.tile_tot[my_ty * U.tiles_x + my_tx], run);
  } else if (threadIdx.x >= 64 && threadIdx.x - 64 < ng) {
    U.grpinfo[g0 + threadIdx.x - 64] = U.info;
  }
}
__device__ __forceinline__ void rank_store_count_any(int ssh, const u64 *s_k, const uint2 *s_r, const uint32_t *s_start,
                                                     uint32_t origin, uint32_t m, uint32_t len, const SubMap &sm, void *l1list,
                                                     size_t first_out, const Unit &U, uint32_t *s_gcnt) {
  switch (ssh) {  // (uniform)
    case 0: rank_store_count<0>(s_k, s_r, s_start, origin, m, len, sm, l1list, first_out, U, s_gcnt); break;
    case 1: rank_store_count<1>(s_k, s_r, s_start, origin, m, len, sm, l1list, first_out, U, s_gcnt); break;
    case 2: rank_store_count<2>(s_k, s_r, s_start, origin, m, len, sm, l1list, first_out, U, s_gcnt); break;
    default: rank_store_count<3>(s_k, s_r, s_start, origin, m, len, sm, l1list, first_out, U, s_gcnt); break;
  }
}

// An item of the work list (its two words: layout_buckets).  Requested one item ahead.
struct WorkItem {
  uint4 bi;     // first slot, entries, overflow records | slices << 16, unit words
  uint32_t w;   // bucket | slice << 16 | (a slice) << 31
};

__device__ __forceinline__ void bucket_sort_body(const BinArgs &a, void *geom, void *bin) {
  __shared__ u64 s_k[BIN_CAP];
  __shared__ uint2 s_r[BIN_CAP];  // the entries' tile rectangles travel with them (a gather of rect[id] per sorted
                                  // entry cost 8 us per 8 renders: random 8-byte reads)
  __shared__ uint32_t s_sub[2 * SUB_BINS + 1];  // sub-bin starts [SUB_BINS + 1], then counters / cursors [SUB_BINS]
  uint32_t *const s_start = s_sub, *const s_cur = s_sub + SUB_BINS + 1;
  // the overflow records of the bucket at hand (three words each: a bucket beyond its region only), then the
  // (group, tile) counters of its run: sized by the host (sort_lds_bytes)
  HIP_DYNAMIC_SHARED(uint32_t, s_seg)
  __shared__ uint32_t s_run[256];
  __shared__ uint32_t s_big;
  __shared__ uint32_t s_wt[4];
  __shared__ uint32_t s_ticket[2];
  __shared__ UnitShared s_unit;
  uint32_t (*s_cnt)[256] = reinterpret_cast<uint32_t (*)[256]>(s_cur);  // (the fallback sort's counters: s_cur is free then)
  static_assert(SUB_BINS >= (SORT_BLOCK / 64) * 256, "s_cnt overlay");
  uint32_t *__restrict__ bk = at<uint32_t>(geom, a.g_bk);
  const uint4 *__restrict__ work = at<uint4>(geom, a.g_work);
  const uint16_t *__restrict__ rect = at<uint16_t>(geom, a.g_rect);
  const uint4 *__restrict__ segs = at<uint4>(geom, a.g_segs);
  const uint4 *__restrict__ l1tmp = at<uint4>(bin, a.b_l1tmp);
  u64 *__restrict__ l1a = at<u64>(bin, a.b_l1a);
  u64 *__restrict__ l1b = at<u64>(bin, a.b_l1b);
  void *l1list = at<char>(bin, a.b_l1);  // (sorted entries: Sorted<ss_shift>::T)
  const int tid = threadIdx.x;
  const int nbins = 1 << a.lg;
  BinTrace tr(3);
  // ---- The render's work list (level 1's last workgroup wrote it: slices first, then the buckets sorted whole by
  // size class) is taken item by item through a ticket counter -- the workgroups' lives differed by 2x with a fixed
  // share of buckets each (bucket sizes differ by 10x, a slice costs four average buckets).  A workgroup's first item
  // is its own number (no ticket: one memory round trip less before the first entry is read); the ticket for the next
  // one is drawn when an item is begun and its words are requested while the item is sorted.  The memory side hands
  // out ~8 tickets per us and word (a single counter per render capped the launch at its 540 items' 50 us), so the
  // list is dealt to WORK_QUEUES counters: item i belongs to queue i % WORK_QUEUES, a workgroup draws from the queue
  // of its own number -- the list is in size order, so every queue holds the same mix.
  uint32_t t = blockIdx.x;
  const uint32_t queue = blockIdx.x % WORK_QUEUES;
  WorkItem cur, nxt;
  cur.bi = work[2 * t], cur.w = work[2 * t + 1].x;  // (stale beyond the list: checked against n_items below)
  const uint32_t n_items = bk[BK_NITEMS];
  const uint32_t map_lo = bk[BK_KMIN], map_shift = bk[BK_SHIFT];
  nxt = cur;
  Unit U;
  U.grpbase = at<uint32_t>(bin, a.b_grpbase), U.grpinfo = at<uint32_t>(bin, a.b_grpinfo), U.cntu = at<uint32_t>(bin, a.b_cntu);
  U.tile_tot = at<uint32_t>(bin, a.b_totals), U.tiles_x = a.gi.tiles_x, U.tiles_y = a.gi.tiles_y;
  BucketSrc S;
  S.ovf = l1tmp + (size_t)(a.gi.NS << a.lg) * BUCKET_REGION, S.ovf_cap = a.l1cap, S.rec = s_seg;
  uint4 mine[PER];
  bool have_mine = false;  // `mine` holds this item's entries already (requested while the previous item was ranked)
  tr.mark();
  for (uint32_t par = 0; t < n_items; par ^= 1u) {
    // the next ticket: drawn now, published behind the first barrier that comes after the atomic has returned
    uint32_t t_next = 0;
    if (tid == 0) t_next = queue + (uint32_t)WORK_QUEUES * atomicAdd(&bk[BK_WORK + queue], 1u);
    const bool is_slice = (cur.w >> 31) != 0u;
    const uint32_t b = cur.w & 0xffffu, j = (cur.w >> 16) & 255u;
    const uint4 bi = cur.bi;
    const uint32_t n = bi.y, base = bi.x, J = (bi.z >> 16) & 255u;
    S.region = l1tmp + (size_t)b * BUCKET_REGION, S.nrec = n > (uint32_t)BUCKET_REGION ? min(bi.z & 0xffffu, (uint32_t)a.nwg1) : 0u;
    const uint32_t reserved = pad_grp(n + (uint32_t)GRP * J);  // (a bucket sorted whole: J = 0)
    const uint32_t sup = b >> a.lg;
    U.unit = (bi.w & 0xfffu) + (is_slice ? j : 0u), U.info = (bi.w & ~0xfffu) | U.unit;
    U.tx0 = (int)((sup % a.gi.stx) << a.gi.ss_shift), U.ty0 = (int)((sup / a.gi.stx) << a.gi.ss_shift);
    const SubMap sm = sub_bin_map(map_lo, map_shift, b & (uint32_t)(nbins - 1));
    // (what does not fit the sorted array -- only when the instance capacity overflowed -- is skipped, like an item that
    // is not what it says: cannot happen)
    const bool skip = n == 0u || (size_t)base + reserved > a.l1cap || (is_slice ? j >= J : (J != 0u));
    __syncthreads();  // (the previous item's LDS has been consumed)
    // (a bucket beyond its region: its overflow records;) the sub-bin counters
    if (S.nrec) load_overflow_records(s_seg, segs + (size_t)b * a.nwg1, S.nrec);
#pragma unroll
    for (int u = 0; u < SUB_OWN; ++u) s_cur[tid * SUB_OWN + u] = 0u;
    if (tid == 0) s_big = 0u;
    lds_barrier();
    bool in_lds = false;  // the run's entries are sorted in LDS (else: nothing to do, or the byte passes)
    uint32_t first = 0, m = n, len = pad_grp(n), run_off = 0, used = 0;
    if (!skip && !is_slice) {
      if (n <= (uint32_t)BIN_CAP) {
        // the thread's entries, every load issued before the first use (unless they are here already)
        if (!have_mine) {  // (uniform)
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if ((uint32_t)q * SORT_BLOCK < n) mine[q] = S.region[min((uint32_t)q * SORT_BLOCK + tid, n - 1)];  // (n <= BIN_CAP: all in the region)
        }
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if ((uint32_t)q * SORT_BLOCK + tid < n) atomicAdd(&s_cur[sub_bin(mine[q].x, sm)], 1u);
      }
    } else if (!skip) {
      // a slice counts the bucket's sub-bins itself, takes the run of sub-bins whose first entry falls into its share
      // of the bucket, and sorts those (<= SLICE_TARGET + SUB_MAX = BIN_CAP entries)
      for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {
        uint32_t kk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kk[q] = bucket_key(S, min(e0 + q * SORT_BLOCK + tid, n - 1));
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (e0 + q * SORT_BLOCK + tid < n) atomicAdd(&s_cur[sub_bin(kk[q], sm)], 1u);
      }
    }
    if (tid == 0) s_ticket[par ^ 1u] = t_next;
    lds_barrier();
    // the next item's words
    const uint32_t tn = s_ticket[par ^ 1u];
    if (tn < n_items) nxt.bi = work[2 * tn], nxt.w = work[2 * tn + 1].x;
    bool fat = false;  // some sub-bin too large to rank
    if (!skip && (is_slice || n <= (uint32_t)BIN_CAP)) {
      fat = is_slice ? scan_sub_bins<false>(s_cur, s_start, s_run, &s_big) : scan_sub_bins<true>(s_cur, s_start, s_run, &s_big);
      in_lds = !fat;
    }
    const uint32_t n_gcnt = (uint32_t)(BIN_CAP / GRP) << (2 * a.gi.ss_shift);  // (group, tile) counters of a run
    if (!skip && !is_slice && in_lds) {
      // (the (group, tile) counters of the run)
      for (uint32_t q = tid; q < n_gcnt; q += SORT_BLOCK) s_seg[q] = 0u;
#pragma unroll
      for (int q = 0; q < PER; ++q)
        if ((uint32_t)q * SORT_BLOCK + tid < n) {
          const uint32_t p = atomicAdd(&s_cur[sub_bin(mine[q].x, sm)], 1u);
          s_k[p] = ((u64)mine[q].x << 32) | (u64)mine[q].y;
          s_r[p] = make_uint2(mine[q].z, mine[q].w);
        }
      lds_barrier();
    } else if (!skip && is_slice && in_lds) {
      // The slices' boundaries in the sorted bucket: P[i] = first entry of the first non-empty sub-bin that starts at
      // or behind i Tn (P[0] = 0, P[J] = n).  A non-empty sub-bin [S, E) makes E the boundary of every i with
      // S < i Tn <= E.  (s_run has 256 words: J <= 255.)
      const uint32_t Tn = (n + J - 1) / J;
      s_run[tid] = tid == 0 ? 0u : n;
      lds_barrier();
#pragma unroll
      for (int u = 0; u < SUB_OWN; ++u) {
        const uint32_t f = (uint32_t)(tid * SUB_OWN + u), sb = s_start[f], se = s_start[f + 1];
        if (se > sb)
          for (uint32_t i = sb / Tn + 1; i * Tn <= se && i < J; ++i) s_run[i] = se;
      }
      lds_barrier();
      // this slice: sorted entries [P[j], P[j + 1]); its run starts behind the earlier slices' runs, each rounded up
      // to whole groups
      first = s_run[j], m = min(s_run[j + 1] - first, (uint32_t)BIN_CAP);
      const uint32_t oex = block_scan_excl((uint32_t)tid < J ? pad_grp(s_run[tid + 1] - s_run[tid]) : 0u, s_wt, used);
      if ((uint32_t)tid == j) s_big = oex;  // (s_big is free: the fat sub-bin test has been read)
      lds_barrier();
      run_off = s_big, len = pad_grp(m);
      if (m > 0u) {
        // cursors: the slice's sub-bins start at `first`
#pragma unroll
        for (int u = 0; u < SUB_OWN; ++u) s_cur[tid * SUB_OWN + u] = s_start[tid * SUB_OWN + u] - first;
        lds_barrier();
        for (uint32_t e0 = 0; e0 < n; e0 += 8 * SORT_BLOCK) {  // the bucket once more: the slice's entries stay
          uint4 kv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) kv[q] = bucket_entry(S, min(e0 + q * SORT_BLOCK + tid, n - 1));
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const uint32_t f = sub_bin(kv[q].x, sm), sbn = s_start[f];
            if (e0 + q * SORT_BLOCK + tid < n && sbn >= first && sbn < first + m) {
              const uint32_t p = atomicAdd(&s_cur[f], 1u);
              if (p < (uint32_t)BIN_CAP) s_k[p] = ((u64)kv[q].x << 32) | (u64)kv[q].y, s_r[p] = make_uint2(kv[q].z, kv[q].w);
            }
          }
        }
        lds_barrier();
        // (a slice reads the bucket -- through its overflow records, if any -- twice: the counters are cleared here)
        for (uint32_t q = tid; q < n_gcnt; q += SORT_BLOCK) s_seg[q] = 0u;
        lds_barrier();
      }
    }
    // The next item's entries, into the registers this item's have just left -- a bucket sorted whole is one contiguous
    // read, and its words arrived behind the barriers above: the entries arrive while this item is ranked, instead
    // of a memory round trip with nothing to do at the top of the next item.
    {
      const uint32_t nn = nxt.bi.y;
      have_mine = tn < n_items && (nxt.w >> 31) == 0u && nn != 0u && nn <= (uint32_t)BIN_CAP && ((nxt.bi.z >> 16) & 255u) == 0u;
      if (have_mine) {
        const uint4 *__restrict__ rn = l1tmp + (size_t)(nxt.w & 0xffffu) * BUCKET_REGION;
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if ((uint32_t)q * SORT_BLOCK < nn) mine[q] = rn[min((uint32_t)q * SORT_BLOCK + tid, nn - 1)];
      }
    }
    if (skip) {
      // nothing
    } else if (in_lds) {
      if (m > 0u)
        rank_store_count_any(a.gi.ss_shift, s_k, s_r, s_start, first, m, len, sm, l1list, (size_t)base + run_off, U, s_seg);
      else
        zero_unit_row(U);
      if (is_slice && j == J - 1 && used < reserved) {  // what the bucket reserved beyond its slices' runs: empty entries, empty rows
        clear_sorted(a.gi.ss_shift, l1list, (size_t)base + used, reserved - used);
        __threadfence_block();
        __syncthreads();
        count_run_global_any(a.gi.ss_shift, l1list, (size_t)base + used, reserved - used, U, s_unit, false);
      }
    } else if (!is_slice || j == 0u) {
      // hundreds of entries at nearly one depth (or an oversized bucket the slice list had no room for): the byte
      // passes, by ONE workgroup -- the bucket's, or its slice 0 -- over the whole bucket as one run
      __syncthreads();
      radix_fallback(S, l1a + base, l1b + base, l1list, (size_t)base, a.gi.ss_shift, U, rect, n, reserved, s_run, s_cnt);
      count_run_global_any(a.gi.ss_shift, l1list, (size_t)base, reserved, U, s_unit, true);
    } else {
      zero_unit_row(U);  // (a further slice of such a bucket)
    }
    tr.mark(skip ? 0u : (in_lds ? (is_slice ? 2u : 1u) : 3u));  // (what the item was: sorted in LDS whole / a slice / the byte passes or their idle slices)
    cur = nxt, t = tn;
  }
  tr.flush();
}

// ------------------------------------------------------------------------------------ 3. level 2, second half: fill
// A WAVE per group of 64 sorted entries (persistent: group g = 4 blockIdx.x + wave, + 4 gridDim.x, ...), the next
// group's entries and words requested before the current one is worked on.  No barrier inside the loop.
// These kernels are INSTRUCTION-ISSUE bound, so the walk is written for instruction count: the supertile edge is a
// template parameter, every entry forms the bit mask of the supertile's tiles its rectangle covers once, and the
// per-tile loop is unrolled over the mask bits: one ballot, one v_readlane, mbcnt and ONE store per tile.
// Lane j of every wave owns tile j of the supertile: its next free slot.
// One tile of the fill's ordered filter, by hand: the entries of the wave that cover tile J of the supertile (bit JB of
// `mhalf`, the 32-bit half of their masks that holds it) take consecutive slots from lane J's counter `c`, in lane
// order, and store their Gaussian id -- 11 instructions (the compiler's version of the same loop body: 23, it forms
// the predicate twice, builds a 64-bit address and compares every slot with the capacity; the caller takes this path
// only when the render's instances fit the capacity).  `vals`: SGPR pair; the byte offset of a slot fits 32 bits.
template <int J>
__device__ __forceinline__ void place_tile(uint32_t mhalf, uint32_t c, uint32_t id, uint32_t *vals) {
  constexpr int JB = J & 31;
  uint32_t t;
  u64 save;
  uint32_t first;
  asm volatile("v_bfe_u32 %[t], %[m], %[jb], 1\n\tv_cmp_ne_u32_e32 vcc, 0, %[t]\n\ts_cbranch_vccz L_place_tile_%=\n\ts_nop 0\n\ts_mov_b64 %[sv], exec\n\tv_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\tv_readlane_b32 %[sf], %[c], %[j]\n\tv_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\ts_mov_b64 exec, vcc\n\tv_add_lshl_u32 %[t], %[sf], %[t], 2\n\tglobal_store_dword %[t], %[id], %[base]\n\ts_mov_b64 exec, %[sv]\nL_place_tile_%=:" : [t] "=&v"(t), [sv] "=&s"(save), [sf] "=&s"(first) : [m] "v"(mhalf), [c] "v"(c), [id] "v"(id), [base] "s"(vals), [jb] "n"(JB), [j] "n"(J) : "vcc", "memory");
}

template <int SSH>
__device__ __forceinline__ void level2_fill_body(const BinArgs &a, void *geom, void *bin, uint8_t *__restrict__ grad_flags,
                                                 uint32_t *__restrict__ totals_out, uint32_t *s_ts, uint32_t *s_wt) {
  constexpr int ss = 1 << SSH, ntile = ss * ss;
  const uint32_t *__restrict__ meta = at<uint32_t>(bin, a.b_meta);
  const typename Sorted<SSH>::T *__restrict__ l1list = Sorted<SSH>::at(at<char>(bin, a.b_l1), 0);
  const uint32_t *__restrict__ grpbase = at<uint32_t>(bin, a.b_grpbase);
  const uint32_t *__restrict__ grpinfo = at<uint32_t>(bin, a.b_grpinfo);
  const uint32_t *__restrict__ cntu = at<uint32_t>(bin, a.b_cntu);
  uint32_t *__restrict__ tile_tot = at<uint32_t>(bin, a.b_totals);
  uint32_t *__restrict__ vals = at<uint32_t>(bin, a.b_vals);
  const uint32_t *__restrict__ total = at<uint32_t>(geom, a.g_total);
  const int tiles_x = a.gi.tiles_x, tiles_y = a.gi.tiles_y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  BinTrace tr(5);
  // The groups of a render lie supertile by supertile, and a tile's list is written by the groups of its supertile:
  // each of the chip's eight XCDs (a workgroup's XCD = its number % 8: the launch's x dimension is a multiple of 8)
  // takes ONE contiguous eighth of the groups, so that the pieces of a tile's list meet in one L2 and leave it as
  // whole cache lines.  Inside its XCD's eighth a wave takes every (waves of the XCD)-th group.
  const uint32_t n_grp = meta[META_NGRP];
  const uint32_t xcd = blockIdx.x & 7u, wg_in_xcd = blockIdx.x >> 3, gstride = (gridDim.x >> 3) * (SORT_BLOCK / 64);
  const uint32_t per_xcd = (n_grp + 7u) / 8u, g_end = min((xcd + 1u) * per_xcd, n_grp);
  uint32_t g = xcd * per_xcd + wg_in_xcd * (SORT_BLOCK / 64) + wave;
  typename Sorted<SSH>::T en_n = Sorted<SSH>::make(0u, 0ull);
  uint32_t row_n = 0u, info_n = 0u, info_nn = 0u;
  // (a row's first `ntile` words are all there is: the other lanes do not ask for theirs)
  if (g < g_end) en_n = l1list[(size_t)g * GRP + lane], row_n = lane < ntile ? grpbase[(size_t)g * GRP + lane] : 0u, info_n = grpinfo[g];
  if (g + gstride < g_end) info_nn = grpinfo[g + gstride];  // (a group's word is known TWO groups ahead: see the loop)
  // The unit rows of the supertile's earlier buckets, summed per tile (lanes [0, ntile)): requested one group ahead --
  // the first eight rows, what a supertile's depth bins come to; a supertile with more units (slices) fetches the rest
  // when it gets there.
  auto unit_rows_request = [&](uint32_t info_w, uint32_t (&v)[8]) {
    const uint32_t unit = info_w & 0xfffu, unit0 = (info_w >> 12) & 0xfffu;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = lane < ntile && unit0 + r < unit ? cntu[(size_t)(unit0 + r) * GRP + lane] : 0u;
  };
  auto unit_rows_sum = [&](uint32_t info_w, const uint32_t (&v)[8]) {
    const uint32_t unit = info_w & 0xfffu, unit0 = (info_w >> 12) & 0xfffu;
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) c += v[r];
    if (lane < ntile)
      for (uint32_t u = unit0 + 8; u < unit; ++u) c += cntu[(size_t)u * GRP + lane];
    return c;
  };
  const uint32_t R = total[0];
  const bool fits = R <= a.R_cap;
  // exclusive scan of the tile totals, by every workgroup itself -- thread t owns tiles [t K, (t + 1) K)
  const int K = (a.T + SORT_BLOCK - 1) / SORT_BLOCK;
  uint32_t my_first = 0;  // instances before this thread's tiles
  {
    uint32_t sum = 0;
    for (int q = 0; q < K; ++q) sum += tid * K + q < a.T ? tile_tot[tid * K + q] : 0u;
    uint32_t all;
    my_first = block_scan_excl(sum, s_wt, all);
    uint32_t run = my_first;
    for (int q = 0; q < K; ++q) {
      const int t = tid * K + q;
      if (t >= a.T) break;
      s_ts[t] = run;
      run += tile_tot[t];
    }
  }
  if (grad_flags) {
    // the backward's "gradient record written" flags, indexed by instance slot: cleared over [0, R) with wide stores
    const size_t nb16 = ((size_t)min(R, a.R_cap) + 15) / 16;
    uint4 *f4 = reinterpret_cast<uint4 *>(grad_flags);
    for (size_t q = (size_t)blockIdx.x * SORT_BLOCK + tid; q < nb16; q += (size_t)gridDim.x * SORT_BLOCK)
      f4[q] = make_uint4(0u, 0u, 0u, 0u);
    // ... and the per-Gaussian hit masks (one 64-bit word each: what the projection backward reads instead of the flags)
    uint4 *h4 = at<uint4>(geom, a.g_hit);
    const size_t nh16 = ((size_t)a.N + 1) / 2;
    for (size_t q = (size_t)blockIdx.x * SORT_BLOCK + tid; q < nh16; q += (size_t)gridDim.x * SORT_BLOCK)
      h4[q] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (blockIdx.x == 0) {
    // tile ranges clamped to the instance capacity, overflow flag, and the backward's three work counters cleared
    // for the blend forward behind this kernel
    uint32_t *ranges = at<uint32_t>(bin, a.b_ranges), *total_w = at<uint32_t>(geom, a.g_total);
    uint32_t *work_count = at<uint32_t>(bin, a.b_work);
    if (tid == 0) {
      work_count[0] = 0, work_count[1] = 0, work_count[2] = 0;
      uint32_t *meta_w = at<uint32_t>(bin, a.b_meta);
      meta_w[META_TICK] = 0u, meta_w[META_TICK + 1] = 0u;  // the blend forward's (tiles done | buckets reached) word
    }
    // ... and the bucket totals, so that the chain can run again on the same projection (preprocess clears them too)
    uint32_t *bk = at<uint32_t>(geom, a.g_bk);
    for (int t = tid; t < 2 * MAX_BUCKETS; t += SORT_BLOCK) bk[BK_TOT + t] = 0u;
    if (tid == 0) bk[BK_DONE] = 0u, bk[BK_OVF] = 0u;
    uint32_t run = my_first;
    int ovf = total_w[1] != 0u;
    for (int q = 0; q < K; ++q) {
      const int t = tid * K + q;
      if (t >= a.T) break;
      const uint32_t v = tile_tot[t];
      // an empty tile reports (0, 0) like the published identifyTileRanges leaves it
      ranges[2 * t] = v ? min(run, a.R_cap) : 0u, ranges[2 * t + 1] = v ? min(run + v, a.R_cap) : 0u;
      if (run + v > a.R_cap) total_w[1] = 1, ovf = 1;  // capacity overflow: flagged, never written out of bounds
      run += v;
    }
    ovf = __syncthreads_or(ovf);
    // the render's (R, overflow) for the step-level array (dimo_render_desc.totals_out)
    if (totals_out && tid == 0) totals_out[0] = R, totals_out[1] = (uint32_t)(ovf != 0);
  }
  lds_barrier();  // (the tile starts are complete)
  tr.mark();
  uint32_t csum_n = 0;  // the unit rows' sum of the group at hand
  {
    uint32_t v[8];
    unit_rows_request(info_n, v);
    csum_n = g < g_end ? unit_rows_sum(info_n, v) : 0u;
  }
  for (; g < g_end; g += gstride) {
    const typename Sorted<SSH>::T en = en_n;
    const uint32_t row = row_n, info = info_n, csum = csum_n;
    uint32_t vn[8];
    {  // the next group of this wave: its entries and row; its unit rows (its word arrived a group ago); the word of the one behind it
      const uint32_t gn = g + gstride;
      info_n = info_nn;
      if (gn < g_end) {
        en_n = l1list[(size_t)gn * GRP + lane];
        row_n = lane < ntile ? grpbase[(size_t)gn * GRP + lane] : 0u;
        unit_rows_request(info_n, vn);
        if (gn + gstride < g_end) info_nn = grpinfo[gn + gstride];
      }
    }
    const uint32_t sup = info >> 24;
    const int tx0 = (int)((sup % (uint32_t)a.gi.stx) << SSH), ty0 = (int)((sup / (uint32_t)a.gi.stx) << SSH);
    const int my_tx = tx0 + (lane & (ss - 1)), my_ty = ty0 + (lane >> SSH);
    const bool my_in = lane < ntile && my_tx < tiles_x && my_ty < tiles_y;
    const u64 mask = Sorted<SSH>::mask(en);
    // lane j: the next free slot of tile j = tile start + the supertile's earlier units + the unit's earlier groups
    uint32_t c = row + csum;
    if (my_in) c += s_ts[my_ty * tiles_x + my_tx];
    if (fits) {  // (uniform: every instance has a slot)
      const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
      static_for<0, ntile>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        place_tile<j>(j < 32 ? mlo : mhi, c, en.x, vals);
      });
    } else {
#pragma unroll
      for (int j = 0; j < ntile; ++j) {
        const bool cov = (mask >> j) & 1ull;
        const u64 bal = __ballot(cov);
        if (bal == 0) continue;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
        const uint32_t pos = first + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (cov && pos < a.R_cap) vals[pos] = en.x;
      }
    }
    if (g + gstride < g_end) csum_n = unit_rows_sum(info_n, vn);  // (requested above: arrived behind the tile steps)
  }
  if (blockIdx.x == 0) {
    // ---- the blend forward's dispatch order: tiles by descending list length (a counting sort over 256 classes of
    // 16 entries; the order inside a class is whatever the LDS atomics make of it -- it only decides WHEN a tile's
    // workgroup starts).  The launch otherwise ends with a tail a quarter of its length.
    uint32_t *__restrict__ order = at<uint32_t>(bin, a.b_order);
    __shared__ uint32_t s_hist[256];
    __syncthreads();
    s_hist[tid] = 0u;
    __syncthreads();
    for (int t = tid; t < a.T; t += SORT_BLOCK) atomicAdd(&s_hist[255u - min(tile_tot[t] >> 4, 255u)], 1u);
    __syncthreads();
    const uint32_t mine = s_hist[tid];
    uint32_t all;
    const uint32_t start = block_scan_excl(mine, s_wt, all);
    s_hist[tid] = start;
    __syncthreads();
    for (int t = tid; t < a.T; t += SORT_BLOCK) order[atomicAdd(&s_hist[255u - min(tile_tot[t] >> 4, 255u)], 1u)] = (uint32_t)t;
  }
  tr.flush();
}

__device__ __forceinline__ void level2_fill_dispatch(const BinArgs &a, void *geom, void *bin, uint8_t *grad_flags,
                                                     uint32_t *totals_out) {
  HIP_DYNAMIC_SHARED(uint32_t, s_ts)  // first slot of every tile: [T]
  __shared__ uint32_t s_wt[4];
  switch (a.gi.ss_shift) {  // (uniform)
    case 0: level2_fill_body<0>(a, geom, bin, grad_flags, totals_out, s_ts, s_wt); break;
    case 1: level2_fill_body<1>(a, geom, bin, grad_flags, totals_out, s_ts, s_wt); break;
    case 2: level2_fill_body<2>(a, geom, bin, grad_flags, totals_out, s_ts, s_wt); break;
    default: level2_fill_body<3>(a, geom, bin, grad_flags, totals_out, s_ts, s_wt); break;
  }
}

// ------------------------------------------------------------------------------------ kernel entry points
// Every stage exists as a single-render kernel (the C-ABI calls) and as a batched one whose blockIdx.y selects the
// render of a RenderBatch (the native step executor: one launch per stage for all renders of a range).
__global__ void __launch_bounds__(SORT_BLOCK) level1_kernel(BinArgs a, void *geom, void *bin) { level1_body(a, geom, bin); }
// (three waves per SIMD -- three workgroups per CU, what their 50 KB of LDS allow -- is 168 VGPRs)
__global__ void __launch_bounds__(SORT_BLOCK, 3) bucket_sort_kernel(BinArgs a, void *geom, void *bin) {
  bucket_sort_body(a, geom, bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) level2_fill_kernel(BinArgs a, void *geom, void *bin) {
  level2_fill_dispatch(a, geom, bin, nullptr, nullptr);  // (the C-ABI backward clears its own scratch)
}
__global__ void __launch_bounds__(SORT_BLOCK) level1_batched_kernel(BinArgs a, RenderBatch b) {
  level1_body(a, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
__global__ void __launch_bounds__(SORT_BLOCK, 3) bucket_sort_batched_kernel(BinArgs a, RenderBatch b) {
  bucket_sort_body(a, b.r[blockIdx.y].geom, b.r[blockIdx.y].bin);
}
__global__ void __launch_bounds__(SORT_BLOCK) level2_fill_batched_kernel(BinArgs a, size_t flag_off, RenderBatch b) {
  const dimo_render_desc &r = b.r[blockIdx.y];
  level2_fill_dispatch(a, r.geom, r.bin, at<uint8_t>(r.bwd_scratch, flag_off), r.totals_out);
}
// Diagnostic (tests' inspect_state): the depth bits of every instance, gathered from its Gaussian's
__global__ void __launch_bounds__(256) depth_keys_kernel(uint32_t R_cap, const uint32_t *__restrict__ total,
                                                         const uint32_t *__restrict__ key32, const uint32_t *__restrict__ vals,
                                                         uint32_t *__restrict__ out) {
  const uint32_t R = min(total[0], R_cap);
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < R; p += gridDim.x * 256u) out[p] = key32[vals[p]];
}

// ------------------------------------------------------------------------------------ host side
// bucket_sort: persistent workgroups that take the render's work list item by item.  768 of them fit the chip at once
// (43 KB of LDS: 3 per CU) and give the shortest launch when it has the chip to itself (42 us per 8 renders against 48
// with 640); in the step, where the other motion's kernels run beside it, 640 measured 0.4 % more frames/s in four
// interleaved pairs (the launch leaves them a sixth of the LDS).  At least 32 per render.
static unsigned bucket_grid(unsigned nbuckets, int n_renders) {
  const unsigned room = (unsigned)(640 / (n_renders > 0 ? n_renders : 1));
  unsigned g = room < nbuckets ? room : nbuckets;
  return g < 32u ? 32u : g;
}
static size_t sort_lds_bytes(const BinArgs &a) {
  const size_t table = 3 * (size_t)a.nwg1, counters = (size_t)(BIN_CAP / GRP) << (2 * a.gi.ss_shift);
  return (table > counters ? table : counters) * sizeof(uint32_t);
}
static size_t level1_code_bytes(int per) { return (size_t)per * BIG_ENTRIES * SORT_BLOCK * sizeof(uint32_t); }
static bool make_args(int N, int H, int W, int64_t R_cap, int n_renders, const GeomLayout &G, const BinLayout &B,
                      BinArgs &a) {
  if (!make_bin_grid(H, W, a.gi)) return false;  // more than MAX_SUPER * 64 tiles
  // level-1 workgroups walk `per` preprocess blocks each: their fixed work (reducing the per-block words, the bucket
  // tables) is per workgroup, and the kernels are instruction bound -- about 1024 workgroups per launch
  int per = (int)(((size_t)G.nb * (size_t)(n_renders > 0 ? n_renders : 1) + 1023) / 1024);
  // ... and at most ~256 workgroups per render: every workgroup adds to each bucket it touches with one returning
  // atomic, and the workgroups of a model that was never Morton-sorted touch nearly all of them.  (~128 until the
  // workgroups' lifetimes were looked at: all of a launch's workgroups are resident at once, the launch lasts as long
  // as its slowest one, and with four blocks each the slowest took twice the mean.)
  if (per < (G.nb + 255) / 256) per = (G.nb + 255) / 256;
  if (per > MAX_L1_PER) per = MAX_L1_PER;
  // (a bucket's list of overflow records has a slot per level-1 workgroup, MAX_SEG of them at most; an entry's code
  // stays in LDS between the count and the placement: MAX_L1_PER blocks per workgroup at most)
  if (per < G.per) per = G.per;
  if (per > MAX_L1_PER) return false;  // more than MAX_L1_PER * MAX_SEG * 256 Gaussians (2 M)
  a.N = N, a.nb = G.nb, a.per = per, a.nwg1 = (G.nb + per - 1) / per, a.lg = depth_bins_log2(N, a.gi.NS), a.T = B.T;
  if ((a.gi.NS << a.lg) != B.nbuckets) return false;  // (the layout was built for another model size)
  a.R_cap = (uint32_t)B.cap;
  a.l1cap = B.l1cap;
  a.sort_grid = (int)bucket_grid((unsigned)(a.gi.NS << a.lg), n_renders);
  a.g_total = G.total, a.g_rect = G.rect, a.g_tiles = G.tiles, a.g_offsets = G.offsets, a.g_sums = G.block_sums;
  a.g_key32 = G.key32, a.g_bk = G.bk, a.g_segs = G.segs, a.g_work = G.work, a.g_wgob = G.wgob;
  a.g_hit = G.hitmask;
  a.b_meta = B.meta, a.b_l1tmp = B.l1tmp, a.b_l1a = B.l1a, a.b_l1b = B.l1b, a.b_l1 = B.l1list;
  a.b_grpbase = B.grpbase, a.b_grpinfo = B.grpinfo, a.b_cntu = B.cntu;
  a.b_totals = B.totals, a.b_ranges = B.ranges, a.b_work = B.work, a.b_vals = B.vals_b;
  a.b_order = B.order;
  return true;
}

// level2_fill: persistent waves over the groups (their number, ~ entries / 64 + half a group per bucket, is only known
// on the device): sized so that all the renders of a launch fit the chip in one round (2048 workgroups of 256
// threads); a wave then walks ~5 groups at the benchmark configuration
static unsigned level2_grid(int N, int n_renders, int nbuckets) {
  const unsigned want = (unsigned)((4 * (size_t)N) / SORT_BLOCK + nbuckets / 4 + 1);  // ~ one group per wave, were there room
  const unsigned room = (unsigned)(2048 / (n_renders > 0 ? n_renders : 1));
  const unsigned g = want < room ? want : (room > 64u ? room : 64u);
  return (g + 7u) / 8u * 8u;  // (a multiple of 8: the fill deals its groups to the XCDs by workgroup number % 8)
}

int bin_instances(int N, int H, int W, int64_t R_cap, const void *geom_c, void *bin, hipStream_t stream) {
  GeomLayout G(N);
  BinLayout B(R_cap, H, W, N);
  BinArgs a;
  if (!make_args(N, H, W, R_cap, 1, G, B, a)) return DIMO_E_ARG;
  void *geom = const_cast<void *>(geom_c);  // offsets, bucket tables and the overflow flag live in the geometry workspace
  const unsigned nbuckets = (unsigned)(a.gi.NS << a.lg);
  {
    ScopedTimer tm(T_SCAN, stream);
    hipLaunchKernelGGL(level1_kernel, dim3(a.nwg1), dim3(SORT_BLOCK), level1_code_bytes(a.per), stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_SORT, stream);
    hipLaunchKernelGGL(bucket_sort_kernel, dim3(a.sort_grid), dim3(SORT_BLOCK), sort_lds_bytes(a), stream, a, geom, bin);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    const size_t ts_bytes = (size_t)a.T * sizeof(uint32_t);  // (up to 64 KB at 16384 tiles)
    if (ts_bytes > 32768) (void)hipFuncSetAttribute((const void *)level2_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ts_bytes);
    hipLaunchKernelGGL(level2_fill_kernel, dim3(level2_grid(N, 1, (int)nbuckets)), dim3(SORT_BLOCK), ts_bytes, stream, a,
                       geom, bin);
  }
  return check_launch();
}

int bin_instances_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream) {
  if (n <= 0) return DIMO_OK;
  GeomLayout G(c.N);
  BinLayout B(c.R_cap, c.H, c.W, c.N);
  BinArgs a;
  if (!make_args(c.N, c.H, c.W, c.R_cap, n, G, B, a)) return DIMO_E_ARG;
  if (c.bin_bytes < B.bytes || c.geom_bytes < G.bytes) return DIMO_E_WORKSPACE;
  if (c.bwd_scratch_bytes < align_up(B.cap * sizeof(SplatGrad)) + align_up(B.cap)) return DIMO_E_WORKSPACE;
  for (int i = 0; i < n; ++i)
    if (!b.r[i].bwd_scratch) return DIMO_E_ARG;  // the fill pass clears the backward's record flags
  const unsigned nbuckets = (unsigned)(a.gi.NS << a.lg);
  {
    ScopedTimer tm(T_SCAN, stream);
    hipLaunchKernelGGL(level1_batched_kernel, dim3(a.nwg1, n), dim3(SORT_BLOCK), level1_code_bytes(a.per), stream, a, b);
  }
  {
    ScopedTimer tm(T_SORT, stream);
    hipLaunchKernelGGL(bucket_sort_batched_kernel, dim3(a.sort_grid, n), dim3(SORT_BLOCK), sort_lds_bytes(a), stream, a, b);
  }
  {
    ScopedTimer tm(T_TILE_SORT, stream);
    // (flags of the backward's scratch: [records: cap x 64 B][flags: cap x 1 B], see blend.hip)
    const size_t ts_bytes = (size_t)a.T * sizeof(uint32_t);
    if (ts_bytes > 32768) (void)hipFuncSetAttribute((const void *)level2_fill_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ts_bytes);
    hipLaunchKernelGGL(level2_fill_batched_kernel, dim3(level2_grid(c.N, n, (int)nbuckets), n), dim3(SORT_BLOCK), ts_bytes,
                       stream, a, align_up(B.cap * sizeof(SplatGrad)), b);
  }
  return check_launch();
}

int instance_depth_keys(int N, int H, int W, int64_t R_cap, const void *geom, const void *bin, uint32_t *out,
                        hipStream_t stream) {
  GeomLayout G(N);
  BinLayout B(R_cap, H, W, N);
  hipLaunchKernelGGL(depth_keys_kernel, dim3(1024), dim3(256), 0, stream, (uint32_t)B.cap, at<uint32_t>(geom, G.total),
                     at<uint32_t>(geom, G.key32), at<uint32_t>(bin, B.vals_b), out);
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
