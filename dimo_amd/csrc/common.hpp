// Shared declarations of libdimo_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dimo_hip.h"

namespace dimo {

// ---- constants of the published rasterizer (documented in DESIGN.md) -------------------------
constexpr int TILE = DIMO_TILE;
constexpr int NFEAT = DIMO_NFEAT;
constexpr float NEAR_CULL = 0.2f;
constexpr float W_EPS = 0.0000001f;
constexpr float FOV_CLAMP = 1.3f;
constexpr float LOWPASS = 0.3f;
constexpr float LAMBDA_FLOOR = 0.1f;
constexpr float RADIUS_SIGMA = 3.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

// One 64-byte record per Gaussian, written by preprocess and gathered by the blend kernels.
// 64 B = one aligned 4 x dwordx4 gather per tile-instance.
struct __attribute__((aligned(64))) Splat {
  float x, y;         // pixel-space mean
  float A, B, C;      // conic (inverse 2D covariance)
  float opacity;
  float r, g, b;      // colour
  float depth;        // view-space z
  float nx, ny, nz;   // view-space normal
  // the conic once more, pre-multiplied for the blend kernels' base-2 exponent (CONIC_HALF A, CONIC_FULL B, CONIC_HALF C:
  // a visit then computes power' = dx (A' dx + B' dy) + C' dy dy = log2(e) x the published exponent and feeds v_exp_f32
  // directly); they ride in what was padding, in the float4 that carries nz
  float As, Bs, Cs;
};
constexpr float CONIC_HALF = -0.5f * 1.44269504088896340736f, CONIC_FULL = -1.44269504088896340736f;
static_assert(sizeof(Splat) == 64, "Splat must be 64 bytes");

// Per-instance gradient record produced by the blend backward (one per (Gaussian, tile) instance,
// indexed by EMISSION position so that a Gaussian's instances are contiguous).
struct __attribute__((aligned(64))) SplatGrad {
  float m0, mx, my, mxx, mxy, myy;  // moments of g = G * dL/dG over the tile's pixels
  float dr, dg, db, ddepth, dnx, dny, dnz;
  float pad0, pad1, pad2;
};
static_assert(sizeof(SplatGrad) == 64, "SplatGrad must be 64 bytes");

constexpr size_t ALIGN = 256;
__host__ __device__ inline size_t align_up(size_t x, size_t a = ALIGN) { return (x + a - 1) / a * a; }

constexpr int PRE_BLOCK = 256;  // Gaussians per preprocess block (also the scan granule)

constexpr int SORT_BLOCK = 256;  // threads of a bucket-sort workgroup
// Tile binning (binning.hip).  Level 1 drops every visible Gaussian into BUCKETS = (supertile it touches, coarse depth
// bin): the image is cut into <= MAX_SUPER supertiles of SS x SS tiles (SS <= 8: at most 64 tiles, one lane each at
// level 2), the depth bits into 2^nb_log2 bins over a range that brackets the bulk of the keys.
constexpr int MAX_SUPER = 256;     // supertiles
constexpr int MAX_BUCKETS = 2048;  // supertiles x depth bins
constexpr int MAX_SEG = 1024;      // level-1 workgroups per render = slots of a bucket's list of overflow records
                                   // (beyond: several preprocess blocks per workgroup)
constexpr int MAX_L1_PER = 8;      // preprocess blocks a level-1 workgroup walks at most (N <= 8 * 1024 * 256 Gaussians)
struct BinGrid {
  int tiles_x, tiles_y, ss_shift, stx, sty, NS;  // supertile edge = 1 << ss_shift tiles, stx * sty = NS supertiles
};
// supertile edge: the smallest power of two that leaves <= 64 supertiles, else <= MAX_SUPER; edge <= 8
inline __host__ bool make_bin_grid(int H, int W, BinGrid &gi) {
  gi.tiles_x = (W + TILE - 1) / TILE, gi.tiles_y = (H + TILE - 1) / TILE;
  for (int limit : {64, MAX_SUPER})
    for (int sh = 0; sh <= 3; ++sh) {
      const int ss = 1 << sh;
      const int stx = (gi.tiles_x + ss - 1) / ss, sty = (gi.tiles_y + ss - 1) / ss;
      if (stx * sty <= limit) {
        gi.ss_shift = sh, gi.stx = stx, gi.sty = sty, gi.NS = stx * sty;
        return true;
      }
    }
  return false;
}
// depth bins per supertile: buckets of ~300-600 entries on average (a Gaussian touches ~2.4 supertiles; the busiest
// buckets of a frame hold ~6x the average: a few of them exceed the 2048 entries a workgroup sorts in LDS and are
// cut into slices), a power of two <= 32, NS * bins <= MAX_BUCKETS
inline __host__ __device__ int depth_bins_log2(int N, int NS) {
  int lg = 0;
  while (lg < 5 && ((size_t)NS << (lg + 1)) <= (size_t)MAX_BUCKETS && (12 * (size_t)N / 5) > ((size_t)NS << lg) * 600) ++lg;
  return lg;
}
// words of the `bk` area in the geometry workspace
constexpr int MAX_SLICES = 256;                                          // slices of oversized buckets (one workgroup each)
constexpr size_t BK_KMIN = 0, BK_SHIFT = 1, BK_KMIN0 = 2, BK_NBLOG = 3;  // bin map: origin, log2 bin width, smallest key, log2 bins
constexpr size_t BK_NITEMS = 4;                                          // items of bucket_sort's work list
constexpr size_t BK_DONE = 5;                                            // level-1 workgroups that have added their entries
constexpr size_t BK_OVF = 6;                                             // cursor of the unsorted level-1 array's overflow area
constexpr int BUCKET_REGION = 4096;  // slots of a bucket's own region in the unsorted level-1 array
constexpr int WORK_QUEUES = 16;
constexpr size_t BK_WORK = 8;                                            // [WORK_QUEUES] ticket counters of the work list
// [MAX_BUCKETS] 64-bit words: entries of the bucket (low half: summed by ONE returning atomic per (level-1 workgroup,
// bucket it touches) -- what it returns is the workgroup's first place in the bucket) | overflow records listed for it
// (high half)
constexpr size_t BK_TOT = 24;
constexpr size_t BK_WORDS = BK_TOT + 2 * MAX_BUCKETS;

struct GeomLayout {
  size_t splat, rect, tiles, offsets, flags, total, block_sums, key32, bk, segs, work, wgob, hitmask;
  size_t bytes;
  int nb, per, nwg1;  // preprocess blocks; blocks per level-1 workgroup; level-1 workgroups
  __host__ explicit GeomLayout(int N) {
    size_t n = (size_t)(N > 0 ? N : 1);
    size_t o = 0;
    splat = o, o = align_up(o + n * sizeof(Splat));
    rect = o, o = align_up(o + n * 4 * sizeof(uint16_t));
    tiles = o, o = align_up(o + n * sizeof(uint32_t));
    offsets = o, o = align_up(o + n * sizeof(uint32_t));
    flags = o, o = align_up(o + n);
    total = o, o = align_up(o + 4 * sizeof(uint32_t));  // R, overflow flag, level-1 entries, 0
    nb = (int)((n + PRE_BLOCK - 1) / PRE_BLOCK);
    per = (nb + MAX_SEG - 1) / MAX_SEG, nwg1 = (nb + per - 1) / per;
    // per preprocess block [nb + 1] each: tiles touched, min / max of the depth bits of its visible Gaussians,
    // level-1 entries (supertiles touched)
    block_sums = o, o = align_up(o + 4 * ((size_t)nb + 1) * sizeof(uint32_t));
    key32 = o, o = align_up(o + n * sizeof(uint32_t));  // depth bits, 0xffffffff for a Gaussian without tiles
    bk = o, o = align_up(o + BK_WORDS * sizeof(uint32_t));
    // segs[bucket][slot]: (first overflow slot, first place, places, 0) of every level-1 workgroup whose share of the
    // bucket reaches beyond the bucket's region -- `slot` from the bucket's atomic, row stride = the launch's level-1
    // workgroups (<= nwg1); only the listed slots are written and read
    segs = o, o = align_up(o + (size_t)MAX_BUCKETS * nwg1 * 4 * sizeof(uint32_t));
    // bucket_sort's work list = the sorted level-1 array's layout (binning.hip: layout_buckets): eight words per
    // slice and per bucket sorted whole
    work = o, o = align_up(o + (size_t)(MAX_BUCKETS + MAX_SLICES) * 8 * sizeof(uint32_t));
    // wgob[level-1 workgroup][bucket]: where the workgroup's places beyond the bucket's region go in the overflow area
    // (only the words of such buckets are written and read)
    wgob = o, o = align_up(o + (size_t)nwg1 * MAX_BUCKETS * sizeof(uint32_t));
    // per Gaussian one 64-bit word: bit k = its k-th tile instance got a gradient record from the blend backward
    // (cleared with the record flags, set by atomic OR; a Gaussian of more than 64 instances goes by the flags)
    hitmask = o, o = align_up(o + n * sizeof(uint64_t));
    bytes = o;
  }
};

// the small words of the bin workspace (`BinLayout::meta`, uint32[16]): [0] number of groups the level-2 fill walks;
// [2..3] one 64-bit word of the blend forward (tiles done | buckets reached << 32, reset by the fill); [4] the chain
// length -- buckets per backward work item -- the NEXT forward over this workspace uses (blend.hip: adaptive chains)
constexpr int META_NGRP = 0, META_TICK = 2, META_CHAIN = 4;
constexpr int BUCKET = 64;       // list entries per backward work item
constexpr int CKPT_FLOATS = 9;   // T, 7 accumulated features, accumulated weight
struct BinLayout {
  size_t vals_b, ranges, totals, meta, l1tmp, l1a, l1b, l1list, grpbase, grpinfo, cntu, ckpt, work, order, bytes;
  int tiles_x, tiles_y, T;
  size_t cap, ckpt_slots, l1cap;
  int nbuckets;  // (supertile, depth bin) buckets of a render of N Gaussians: each owns a region of l1tmp
  // N = the Gaussians of the model the workspace serves: the bucket count follows it (depth_bins_log2), and with it
  // the regions of the unsorted level-1 array -- 32 MB at 100 k Gaussians / 512^2 (512 buckets) where the layout's
  // limit of 2048 buckets would reserve 128 MB per workspace (round-5 advice: the drop-in path retains one bin workspace
  // per render until the backward).  Every user of a workspace must build its layout from the same (R_cap, H, W, N).
  __host__ BinLayout(int64_t R_cap, int H, int W, int N) {
    cap = (size_t)(R_cap > 0 ? R_cap : 1);
    tiles_x = (W + TILE - 1) / TILE, tiles_y = (H + TILE - 1) / TILE, T = tiles_x * tiles_y;
    BinGrid gi_;
    nbuckets = make_bin_grid(H, W, gi_) ? (gi_.NS << depth_bins_log2(N > 0 ? N : 1, gi_.NS)) : MAX_BUCKETS;
    size_t o = 0;
    // per instance, in the order of the per-tile lists: the Gaussian id.  (The instance's sort key is not stored: its
    // tile is the list it sits in -- `ranges` -- and its 32 depth bits are its Gaussian's, geom key32.)
    vals_b = o, o = align_up(o + cap * sizeof(uint32_t));
    ranges = o, o = align_up(o + (size_t)T * 2 * sizeof(uint32_t));
    totals = o, o = align_up(o + (size_t)T * sizeof(uint32_t));  // instances per tile (atomically summed)
    // level 1 (binning.hip): at most one entry per instance.  l1tmp: the 16-byte entries (depth bits, id, tile
    // rectangle) as the level-1 workgroups leave them: a region of BUCKET_REGION slots per bucket -- 64 KB of
    // HBM per bucket, of which a bucket touches what it holds --, then an overflow area; l1a / l1b: 64-bit scratch of
    // the byte-wise fallback sort; l1list: the sorted entries (id, depth bits, tile rectangle), bucket by bucket, every
    // bucket (and every slice of a cut bucket) rounded up to whole groups of 64 -- hence the slack
    l1cap = (cap + 255) / 256 * 256 + 64 * (size_t)(MAX_BUCKETS + 2 * MAX_SLICES);
    l1tmp = o, o = align_up(o + ((size_t)nbuckets * BUCKET_REGION + l1cap) * 4 * sizeof(uint32_t));
    l1a = o, o = align_up(o + l1cap * sizeof(uint64_t));
    l1b = o, o = align_up(o + l1cap * sizeof(uint64_t));
    l1list = o, o = align_up(o + l1cap * 4 * sizeof(uint32_t));
    // level 2: per group of 64 sorted entries a row of 64 words (tile j of the supertile: entries of the bucket's
    // earlier groups that cover it) and one word (unit | first unit of the supertile << 12 | supertile << 24); per
    // unit (a bucket, or a slice of a cut one) a row of per-tile totals
    grpbase = o, o = align_up(o + l1cap * sizeof(uint32_t));
    grpinfo = o, o = align_up(o + (l1cap / 64) * sizeof(uint32_t));
    cntu = o, o = align_up(o + (size_t)(MAX_BUCKETS + MAX_SLICES) * 64 * sizeof(uint32_t));
    meta = o, o = align_up(o + 16 * sizeof(uint32_t));
    // blend checkpoints: per (tile, bucket of BUCKET list entries) the 256 pixels' compositing state at the
    // bucket's first entry -- slot (lo_tile / BUCKET + tile + bucket), see blend.hip; work: [0] = item count,
    // then one word (tile << 12 | bucket) per bucket some pixel of the tile reaches
    ckpt_slots = cap / BUCKET + (size_t)T + 2;
    ckpt = o, o = align_up(o + ckpt_slots * CKPT_FLOATS * TILE * TILE * sizeof(float));
    // [0..2] = item counts of the three queues (head / second / deeper chains), then 16-byte items: heads at
    // [1, 1 + T), seconds at [1 + T, 1 + 2 T), the rest behind (blend.hip)
    work = o, o = align_up(o + (4 * (ckpt_slots + 2 * (size_t)T) + 8) * sizeof(uint32_t));
    // the tiles by descending list length (the blend forward's dispatch order; written by the level-2 fill)
    order = o, o = align_up(o + (size_t)T * sizeof(uint32_t));
    bytes = o;
  }
};

struct ImgLayout {
  size_t final_T, n_contrib, final_acc, bytes;
  __host__ ImgLayout(int H, int W) {
    size_t p = (size_t)H * W;
    size_t o = 0;
    final_T = o, o = align_up(o + p * sizeof(float));
    n_contrib = o, o = align_up(o + p * sizeof(uint32_t));
    final_acc = o, o = align_up(o + 8 * p * sizeof(float));  // [7 features + weight][H*W], background not included
    bytes = o;
  }
};

template <class T>
__host__ __device__ inline T *at(void *base, size_t off) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off);
}
template <class T>
__host__ __device__ inline const T *at(const void *base, size_t off) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + off);
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS operations, NOT for its global
// loads (__syncthreads() carries a fence that drains vmcnt, i.e. it would wait for every prefetch in flight).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

void set_last_error(hipError_t e, const char *where);  // api.hip; text retrievable with dimo_last_error()
#define check_launch() ::dimo::check_launch_at(__FILE__ ":" DIMO_STR(__LINE__))
#define DIMO_STR2(x) #x
#define DIMO_STR(x) DIMO_STR2(x)
inline int check_launch_at(const char *where) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return DIMO_OK;
  set_last_error(e, where);
  return DIMO_E_LAUNCH;
}
// drop a stale (sticky) error left by another user of the HIP runtime in this thread, e.g. PyTorch
inline void clear_errors() { (void)hipGetLastError(); }

// number of key bits the sort must cover: 32 depth bits + bits of the tile id
inline int key_bits(int T) {
  int b = 0;
  while ((1 << b) < T) ++b;
  return 32 + (b > 0 ? b : 1);
}

// ---- optional HIP-event timing of kernel groups (off by default; see api.hip) ------------------
enum TimedKernel {
  T_PREPROCESS_FWD = 0, T_SCAN, T_EMIT, T_SORT, T_RANGES, T_BLEND_FWD, T_BLEND_BWD, T_PREPROCESS_BWD, T_KNN, T_DIST2,
  T_SSIM_FWD, T_SSIM_BWD, T_DEFORM_FWD, T_DEFORM_BWD, T_LOSS, T_ADAM, T_TIMENET_FWD, T_TIMENET_BWD, T_TILE_SORT, TIMED_COUNT
};
class ScopedTimer {
 public:
  ScopedTimer(int id, hipStream_t s);
  ~ScopedTimer();
  ScopedTimer(const ScopedTimer &) = delete;
  ScopedTimer &operator=(const ScopedTimer &) = delete;

 private:
  int id_;
  hipStream_t stream_;
  hipEvent_t a_, b_;
};

// ---- batched execution of a step's renders (native step executor) -----------------------------
// The executor's descriptors travel BY VALUE in the kernel arguments (8 x 256 B), blockIdx.y selects the render:
// one launch per stage for all renders of a step instead of one per render and stage.
constexpr int MAX_BATCH = 8;
// Renders of a batch that share their TimeNet rows (same (motion, frame) pair, different cameras) share the skinned
// Gaussians: they form a deformation GROUP.  The skinning forward runs once per group (the members' pts / rot /
// scales / opac pointers are redirected to the leader's buffers when the batch is filled) and the skinning backward,
// linear in the rasterizer gradients, runs once on their sum.
struct RenderBatch {
  dimo_render_desc r[MAX_BATCH];
  int n_groups;
  unsigned char leader[MAX_BATCH];   // group -> its first render
  unsigned char members[MAX_BATCH];  // group -> bitmask of its renders (leader included)
};
void group_deformations(RenderBatch &b, int n);
int lbs_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream);
size_t lbs_backward_batched_scratch_bytes(int N, int M, int n);
int lbs_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream, int first_abs = 0,
                         int phase = 0);
int preprocess_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream);
int preprocess_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream);
int bin_instances_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream);
int blend_forward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream);
int blend_backward_batched(const dimo_step_common &c, const RenderBatch &b, int n, hipStream_t stream,
                           bool joint = false);

// ---- internal (C++ linkage) entry points shared between translation units ---------------------
int bin_instances(int N, int H, int W, int64_t R_cap, const void *geom, void *bin, hipStream_t stream);
int instance_depth_keys(int N, int H, int W, int64_t R_cap, const void *geom, const void *bin, uint32_t *out,
                        hipStream_t stream);
int preprocess_backward_launch(int N, int sh_degree, int M, int H, int W, int64_t R_cap, const float *means3D,
                               const float *shs, const float *colors_precomp, const float *scales,
                               const float *rotations, const float *cov3D_precomp, float scale_modifier,
                               const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                               float tanfovy, const int32_t *radii, const void *geom, const void *inst_grad,
                               float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dshs, float *dL_dcolors,
                               float *dL_dopacity, float *dL_dscales, float *dL_drot, float *dL_dcov3D,
                               hipStream_t stream);

}  // namespace dimo
