// Native step executor: runs the per-render kernel chains of one training step (skinning -> projection ->
// binning/sort -> blend, and the mirrored backward) from ONE host call each, fanning the independent renders
// out over a few private HIP streams and joining them with events against the caller's stream.
//
// Why native: after the kernels were fused, a step at 100k Gaussians / 512^2 spent ~70 % of its wall time in
// the Python interpreter (ctypes marshalling, allocator calls, stream context managers: ~0.4 ms per render),
// with the GPU idle.  All buffers are caller-owned and persistent (288 GB of HBM: eight render slots cost
// < 2 GB), so a step is a fixed sequence of launches that needs no host logic in between.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

#ifndef DIMO_EXECUTOR_TYPES  // normally provided by include/dimo_hip.h (kept here as documentation of the layout)
extern "C" {
typedef struct {
  int N, M, H, W, with_normal, local_frame;
  int64_t R_cap;
  const float *xyz, *rotation, *scaling, *opacity, *f_dc;   // raw canonical parameters ([N,3] [N,4] [N,3] [N] [N,1,3])
  const float *c_xyz, *c_log_radius;                        // control points
  const float *nn_dist;
  const int64_t *nn_idx;
  const float *bg;
  float scale_modifier;
  // gradient accumulation targets (views of the flat gradient bucket)
  float *g_xyz, *g_rotation, *g_scaling, *g_opacity, *g_f_dc, *g_c_xyz, *g_c_log_radius;
  void *lbs_scratch;
  size_t lbs_scratch_bytes, geom_bytes, bin_bytes, img_bytes, bwd_scratch_bytes;
  int stage1;            // stage s1: direct deformation (see include/dimo_hip.h)
  const float *log_r;
  float *g_log_r;
} dimo_step_common;

typedef struct {
  const float *view, *proj, *campos;
  float tanfovx, tanfovy;
  const float *d_xyz, *d_rot;      // TimeNet outputs of this render [M,3] [M,4]
  float *g_d_xyz, *g_d_rot;        // their gradient rows (accumulated)
  float *out_color, *out_depth, *out_normal, *out_alpha;
  const float *g_color, *g_depth, *g_normal, *g_alpha;
  // per-slot persistent workspaces
  float *pts, *rot, *scales, *opac;
  int32_t *radii;
  void *geom, *bin, *img, *bwd_scratch;
  float *g_means3D, *g_means2D, *g_shs, *g_opac, *g_scales, *g_rot;
  const float *g_dot;              // optional per-pixel sum of gradient x rendered value (include/dimo_hip.h)
  uint32_t *totals_out;            // optional (R, overflow) copy (include/dimo_hip.h)
} dimo_render_desc;
}
#endif

namespace dimo {

// Three ways to run the renders of a step (dimo_executor_create(n_streams)):
//   n > 0  per-render chains: render i runs its ~20 launches on private stream i % n
//   n == 0 batched: every stage is ONE launch over all renders of the call (blockIdx.y = render), caller's stream
//   n < 0  batched ranges: each [first, first + count) range given to forward_range / backward_launch is one
//          batch, ranges go round-robin over |n| private streams (a motion's renders form a range: its losses run
//          on the caller's stream while the other motions' batches are still rendering)
struct Executor {
  bool batched = false;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> stream_done;
  std::vector<hipEvent_t> render_done;
  std::vector<hipEvent_t> fwd_done;
  std::vector<int> range_stream;  // batched ranges: stream of the range that starts at render i (-1: none)
  std::vector<int> range_count;   // ... and its number of renders
  int next_stream = 0;
  hipEvent_t main_ready = nullptr;
  // per range start: its rasterizer backward ran on the CALLER's stream (joint launch, or a range whose whole chain is
  // on the caller's stream): dimo_executor_backward_accumulate has no event to wait for
  std::vector<char> range_bwd_main;
  // per range start: its skinning backward (phase 1: skin; control-point sums added atomically) already ran,
  // in order behind its rasterizer backward: dimo_executor_backward_accumulate only folds (phase 2)
  std::vector<char> range_skinned;
  // Cross-stream dependencies through stream memory operations instead of events (DIMO_XSTREAM=value): the producer
  // stream writes a sequence number into a device word (hipStreamWriteValue32), the consumer stream waits for it
  // (hipStreamWaitValue32, >=).  Measured on this platform (tools/xstream_latency.hip): 4-5.5 us from the end of the
  // producer's last kernel to the start of the consumer's first against 10.4-15.6 us through an event.
  bool use_values = false;
  uint32_t *sig = nullptr;            // [0]: the caller's stream ("fork"); [1 + si]: private stream si ("done")
  uint32_t fork_seq = 0;
  std::vector<uint32_t> done_seq;     // per private stream: last value written
  std::vector<uint32_t> fwd_val, render_val;  // per range start: the value its fwd_done / render_done stands for
  // side work (dimo_executor_side_stream / _side_done): per private stream, what its last side work stands for
  std::vector<hipEvent_t> side_ev;
  std::vector<uint32_t> side_val;
  std::vector<char> side_pending;
};

// "record" on private stream si what `ev` stands for, and remember it for range start `i`
static int mark_done(Executor *ex, int si, hipEvent_t ev, std::vector<uint32_t> &vals, int i) {
  if (ex->use_values) {
    const uint32_t v = ++ex->done_seq[si];
    vals[i] = v;
    return hipStreamWriteValue32(ex->streams[si], ex->sig + 1 + si, v, 0) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
  }
  return hipEventRecord(ev, ex->streams[si]) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
}
// the caller's stream waits for it
static int wait_done(Executor *ex, hipStream_t main, int si, hipEvent_t ev, const std::vector<uint32_t> &vals, int i) {
  if (ex->use_values)
    return hipStreamWaitValue32(main, ex->sig + 1 + si, vals[i], hipStreamWaitValueGte, 0xffffffffu) == hipSuccess
               ? DIMO_OK : DIMO_E_LAUNCH;
  return hipStreamWaitEvent(main, ev, 0) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
}
// sequence numbers stay below 2^31 (the comparison's signedness is not documented): start over long before
static void renew_values(Executor *ex) {
  bool high = ex->fork_seq > 0x70000000u;
  for (uint32_t v : ex->done_seq) high |= v > 0x70000000u;
  if (!high) return;
  (void)hipDeviceSynchronize();
  (void)hipMemset(ex->sig, 0, 64 * sizeof(uint32_t));
  ex->fork_seq = 0;
  std::fill(ex->done_seq.begin(), ex->done_seq.end(), 0u);
}

__global__ void __launch_bounds__(256) accumulate_kernel(size_t n, float *__restrict__ dst,
                                                         const float *__restrict__ src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

}  // namespace dimo

using namespace dimo;

// Private streams live for the whole process (per device) and are handed to whichever executor exists: HIP maps streams
// onto a few hardware queues in creation order, and the streams of an executor created AFTER another one had been
// destroyed (the trainer makes a new executor whenever a prune changes the number of Gaussians) came to share a queue
// -- the two motions' chains then serialised behind each other's cross-stream waits, 2.8 ms per step instead of 1.4.
static std::vector<hipStream_t> &stream_pool() {
  static std::vector<hipStream_t> pool[64];
  int dev = 0;
  (void)hipGetDevice(&dev);  // (the caller has made the device that owns its buffers current: torch.cuda.set_device)
  return pool[dev >= 0 && dev < 64 ? dev : 0];
}

extern "C" void *dimo_executor_create(int n_streams) {
  if (n_streams < -16 || n_streams > 16) return nullptr;
  Executor *ex = new Executor();
  ex->batched = n_streams <= 0;
  const int S = n_streams < 0 ? -n_streams : n_streams;
  std::vector<hipStream_t> &pool = stream_pool();
  for (int i = 0; i < S; ++i) {
    hipEvent_t e;
    if ((int)pool.size() <= i) {
      hipStream_t s;
      // LOWEST priority: the caller's stream carries the step's critical path (losses, skinning backward, TimeNet);
      // its kernels should win the workgroup slots against the other motion's batch when both are runnable
      int prio_lo = 0, prio_hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_lo) != hipSuccess) {
        delete ex;
        return nullptr;
      }
      pool.push_back(s);
    }
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      delete ex;
      return nullptr;
    }
    ex->streams.push_back(pool[i]);
    ex->stream_done.push_back(e);
    hipEvent_t e2;
    if (hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) {
      delete ex;
      return nullptr;
    }
    ex->side_ev.push_back(e2);
  }
  ex->side_val.assign(S, 0u);
  ex->side_pending.assign(S, 0);
  if (hipEventCreateWithFlags(&ex->main_ready, hipEventDisableTiming) != hipSuccess) {
    delete ex;
    return nullptr;
  }
  ex->done_seq.assign(S, 0u);
  const char *mode = getenv("DIMO_XSTREAM");
  // default: stream write / wait values (7.62 -> 7.79 k frames/s on the benchmark step); DIMO_XSTREAM=event: events
  if (S > 0 && S < 60 && ex->batched && !(mode && std::string(mode) == "event")) {
    if (hipMalloc((void **)&ex->sig, 64 * sizeof(uint32_t)) == hipSuccess &&
        hipMemset(ex->sig, 0, 64 * sizeof(uint32_t)) == hipSuccess)
      ex->use_values = true;
    else
      ex->sig = nullptr;
  }
  return ex;
}

extern "C" void dimo_executor_destroy(void *h) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex) return;
  // The private streams stay in the process-wide pool and may be carrying another live executor's work: wait for THIS
  // executor's own events (what it last recorded on them), not for the streams.  (An event that was never recorded
  // reports complete.)
  for (auto e : ex->stream_done) (void)hipEventSynchronize(e), (void)hipEventDestroy(e);
  for (auto e : ex->side_ev) (void)hipEventSynchronize(e), (void)hipEventDestroy(e);
  for (auto e : ex->render_done) (void)hipEventSynchronize(e), (void)hipEventDestroy(e);
  for (auto e : ex->fwd_done) (void)hipEventSynchronize(e), (void)hipEventDestroy(e);
  if (ex->main_ready) (void)hipEventDestroy(ex->main_ready);
  if (ex->sig) {  // (value mode has no per-executor events to wait for: the streams may still reference the words)
    for (auto st : ex->streams) (void)hipStreamSynchronize(st);
    (void)hipFree(ex->sig);
  }
  delete ex;
}

static void fill_batch(RenderBatch &b, const dimo_render_desc *d, int n) {
  for (int i = 0; i < MAX_BATCH; ++i) b.r[i] = d[i < n ? i : n - 1];
  group_deformations(b, n);
}

static int ensure_events(Executor *ex, int n) {
  while ((int)ex->render_done.size() < n || (int)ex->fwd_done.size() < n) {
    hipEvent_t e, f;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess)
      return DIMO_E_LAUNCH;
    ex->render_done.push_back(e);
    ex->fwd_done.push_back(f);
  }
  if ((int)ex->range_stream.size() < n) ex->range_stream.resize(n, -1), ex->range_count.resize(n, 0);
  if ((int)ex->range_bwd_main.size() < n) ex->range_bwd_main.resize(n, 0);
  if ((int)ex->range_skinned.size() < n) ex->range_skinned.resize(n, 0);
  if ((int)ex->fwd_val.size() < n) ex->fwd_val.resize(n, 0u), ex->render_val.resize(n, 0u);
  return DIMO_OK;
}

static int fork_from_main(Executor *ex, hipStream_t main) {
  if (hipEventRecord(ex->main_ready, main) != hipSuccess) return DIMO_E_LAUNCH;
  for (auto s : ex->streams)
    if (hipStreamWaitEvent(s, ex->main_ready, 0) != hipSuccess) return DIMO_E_LAUNCH;
  return DIMO_OK;
}

static int fork_one(Executor *ex, hipStream_t main, hipStream_t s) {
  if (ex->use_values) {
    renew_values(ex);
    const uint32_t v = ++ex->fork_seq;
    if (hipStreamWriteValue32(main, ex->sig, v, 0) != hipSuccess) return DIMO_E_LAUNCH;
    return hipStreamWaitValue32(s, ex->sig, v, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
  }
  if (hipEventRecord(ex->main_ready, main) != hipSuccess) return DIMO_E_LAUNCH;
  return hipStreamWaitEvent(s, ex->main_ready, 0) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
}

// Stream `s` (private stream `self`, or the caller's stream: self = -1) waits for the side work pending on the OTHER
// private streams (in order behind it on its own).
static int wait_side_work(Executor *ex, hipStream_t s, int self) {
  for (int j = 0; j < (int)ex->streams.size(); ++j) {
    if (j == self || !ex->side_pending[j]) continue;
    if (ex->use_values) {
      if (hipStreamWaitValue32(s, ex->sig + 1 + j, ex->side_val[j], hipStreamWaitValueGte, 0xffffffffu) != hipSuccess)
        return DIMO_E_LAUNCH;
    } else if (hipStreamWaitEvent(s, ex->side_ev[j], 0) != hipSuccess) {
      return DIMO_E_LAUNCH;
    }
  }
  return DIMO_OK;
}

// Side work: something the caller enqueues ITSELF on one of the executor's private streams, concurrently with what it
// goes on enqueueing on its own stream -- the step's KNN next to the TimeNet forward (main_train_dimo.py:257-258 and
// latent_gs_renderer.py:1171-1174 are independent of each other; both feed the skinning).
// dimo_executor_side_stream: private stream `which`, made to wait for everything on main_stream so far; the caller
// launches on it, then calls dimo_executor_side_done.  Every forward chain started afterwards (on any stream) waits for
// it; a chain on stream `which` simply follows in order.
extern "C" void *dimo_executor_side_stream(void *h, int which, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !ex->batched || which < 0 || which >= (int)ex->streams.size()) return nullptr;
  if (fork_one(ex, (hipStream_t)main_stream, ex->streams[which]) != DIMO_OK) return nullptr;
  return (void *)ex->streams[which];
}
extern "C" int dimo_executor_side_done(void *h, int which) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !ex->batched || which < 0 || which >= (int)ex->streams.size()) return DIMO_E_ARG;
  if (ex->use_values) {
    const uint32_t v = ++ex->done_seq[which];
    ex->side_val[which] = v;
    if (hipStreamWriteValue32(ex->streams[which], ex->sig + 1 + which, v, 0) != hipSuccess) return DIMO_E_LAUNCH;
  } else if (hipEventRecord(ex->side_ev[which], ex->streams[which]) != hipSuccess) {
    return DIMO_E_LAUNCH;
  }
  ex->side_pending[which] = 1;
  return DIMO_OK;
}

// The caller's stream waits for the side work of private stream `which`.
extern "C" int dimo_executor_wait_side(void *h, int which, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !ex->batched || which < 0 || which >= (int)ex->streams.size()) return DIMO_E_ARG;
  if (!ex->side_pending[which]) return DIMO_OK;
  hipStream_t main = (hipStream_t)main_stream;
  if (ex->use_values)
    return hipStreamWaitValue32(main, ex->sig + 1 + which, ex->side_val[which], hipStreamWaitValueGte, 0xffffffffu) ==
                   hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
  return hipStreamWaitEvent(main, ex->side_ev[which], 0) == hipSuccess ? DIMO_OK : DIMO_E_LAUNCH;
}
// Private stream `which` as it is (no dependency added): for work that continues what the stream already holds.
extern "C" void *dimo_executor_private_stream(void *h, int which) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !ex->batched || which < 0 || which >= (int)ex->streams.size()) return nullptr;
  return (void *)ex->streams[which];
}

// Launch chunks (start, count <= MAX_BATCH) over the ranges inside [first, first + count).  A range longer than a
// batch is cut from its start exactly like its forward was (deformation groups are formed per launch: the backward
// must see the forward's groups); whole ranges are merged into one launch while they fit -- groups never straddle
// ranges (a range is one motion's renders).  Renders outside any range (plain batched mode) are cut from `first`.
static bool plan_chunks(const Executor *ex, int first, int count, std::vector<std::pair<int, int>> &chunks) {
  chunks.clear();
  const int end = first + count;
  const bool ranged = !ex->streams.empty() && ex->batched;
  int i = first;
  while (i < end) {
    if (!ranged || i >= (int)ex->range_stream.size() || ex->range_stream[i] < 0) {
      if (ranged) return false;  // not a range start
      const int m = end - i < MAX_BATCH ? end - i : MAX_BATCH;
      chunks.emplace_back(i, m), i += m;
      continue;
    }
    const int len = ex->range_count[i];
    if (len <= 0 || i + len > end) return false;
    if (len > MAX_BATCH) {
      for (int o = 0; o < len; o += MAX_BATCH) chunks.emplace_back(i + o, len - o < MAX_BATCH ? len - o : MAX_BATCH);
      i += len;
      continue;
    }
    int tot = len, j = i + len;
    while (j < end && ex->range_stream[j] >= 0 && ex->range_count[j] <= MAX_BATCH && tot + ex->range_count[j] <= MAX_BATCH &&
           j + ex->range_count[j] <= end)
      tot += ex->range_count[j], j += ex->range_count[j];
    chunks.emplace_back(i, tot);
    i = j;
  }
  return true;
}

static int batched_forward(const dimo_step_common *c, const dimo_render_desc *d, int first, int count,
                           hipStream_t s) {
  for (int i0 = first; i0 < first + count; i0 += MAX_BATCH) {
    const int m = first + count - i0 < MAX_BATCH ? first + count - i0 : MAX_BATCH;
    RenderBatch b;
    fill_batch(b, d + i0, m);
    int rc = lbs_forward_batched(*c, b, m, s);
    if (!rc) rc = preprocess_forward_batched(*c, b, m, s);
    if (!rc) rc = bin_instances_batched(*c, b, m, s);
    if (!rc) rc = blend_forward_batched(*c, b, m, s);
    if (rc) return rc;
  }
  return DIMO_OK;
}

static int batched_backward_raster(const dimo_step_common *c, const dimo_render_desc *d, int first, int count,
                                   hipStream_t s, bool joint = false) {
  for (int i0 = first; i0 < first + count; i0 += MAX_BATCH) {
    const int m = first + count - i0 < MAX_BATCH ? first + count - i0 : MAX_BATCH;
    RenderBatch b;
    fill_batch(b, d + i0, m);
    int rc = blend_backward_batched(*c, b, m, s, joint);
    if (!rc) rc = preprocess_backward_batched(*c, b, m, s);
    if (rc) return rc;
  }
  return DIMO_OK;
}

extern "C" int dimo_executor_forward_range(void *h, const dimo_step_common *c, int first, int count,
                                           const dimo_render_desc *d, void *main_stream) {
  // Forward chains of renders [first, first + count), after everything enqueued on the caller's stream so far.
  // With private streams the caller's stream does NOT wait: see dimo_executor_join.
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  clear_errors();
  const int S = (int)ex->streams.size();
  if (ex->batched && S == 0) return batched_forward(c, d, first, count, main);
  int rc = ensure_events(ex, first + count);
  if (rc) return rc;
  if (ex->batched) {
    const int si = ex->next_stream++ % S;
    hipStream_t s = ex->streams[si];
    if (first == 0) std::fill(ex->range_stream.begin(), ex->range_stream.end(), -1);  // a new step's ranges
    ex->range_stream[first] = si, ex->range_count[first] = count, ex->range_skinned[first] = 0;
    rc = fork_one(ex, main, s);
    if (!rc) rc = wait_side_work(ex, s, si);
    if (!rc) rc = batched_forward(c, d, first, count, s);
    if (rc) return rc;
    return mark_done(ex, si, ex->fwd_done[first], ex->fwd_val, first);
  }
  if (c->stage1) return DIMO_E_ARG;  // stage s1 runs in the batched modes only
  rc = fork_from_main(ex, main);
  if (rc) return rc;
  for (int i = first; i < first + count; ++i) {
    hipStream_t s = ex->streams[i % S];
    const dimo_render_desc &r = d[i];
    rc = dimo_deform_forward(c->N, c->M, c->local_frame, c->xyz, c->rotation, c->scaling, c->opacity, c->c_xyz,
                             c->c_log_radius, r.d_xyz, r.d_rot, c->nn_dist, c->nn_idx, r.pts, r.rot, r.scales,
                             r.opac, s);
    if (rc) return rc;
    rc = dimo_raster_preprocess_forward(c->N, 0, 1, c->H, c->W, r.pts, c->f_dc, nullptr, r.opac, r.scales, r.rot,
                                        nullptr, c->scale_modifier, r.view, r.proj, r.campos, r.tanfovx, r.tanfovy,
                                        r.radii, r.geom, c->geom_bytes, nullptr, s);
    if (rc) return rc;
    rc = dimo_raster_render_forward(c->N, c->H, c->W, c->R_cap, c->bg, r.geom, r.bin, c->bin_bytes, r.img,
                                    c->img_bytes, r.out_color, r.out_depth, c->with_normal ? r.out_normal : nullptr,
                                    r.out_alpha, s);
    if (rc) return rc;
    if (hipEventRecord(ex->fwd_done[i], s) != hipSuccess) return DIMO_E_LAUNCH;
  }
  return DIMO_OK;
}

extern "C" int dimo_executor_forward(void *h, const dimo_step_common *c, int n, const dimo_render_desc *d,
                                     void *main_stream) {
  return dimo_executor_forward_range(h, c, 0, n, d, main_stream);
}

// The caller's stream waits for the forward of renders [first, first + count).
extern "C" int dimo_executor_join(void *h, int first, int count, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || first < 0 || count < 0) return DIMO_E_ARG;
  if (ex->streams.empty()) return DIMO_OK;  // everything already runs on the caller's stream
  if (first + count > (int)ex->fwd_done.size()) return DIMO_E_ARG;
  for (int i = first; i < first + count; ++i) {
    if (ex->batched && ex->range_stream[i] < 0) continue;  // only range starts carry an event
    if (ex->batched && ex->range_stream[i] >= (int)ex->streams.size()) continue;  // a range on the caller's stream
    if (ex->batched) {
      if (wait_done(ex, main, ex->range_stream[i], ex->fwd_done[i], ex->fwd_val, i)) return DIMO_E_LAUNCH;
      continue;
    }
    if (hipStreamWaitEvent(main, ex->fwd_done[i], 0) != hipSuccess) return DIMO_E_LAUNCH;
  }
  return DIMO_OK;
}

// Rasterizer backward of renders [first, first + count) on their private streams (ordered after everything the
// caller's stream has enqueued so far, i.e. after the kernels that produced the gradient images).
extern "C" int dimo_executor_backward_launch(void *h, const dimo_step_common *c, int first, int count,
                                             const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  clear_errors();
  const int S = (int)ex->streams.size();
  if (ex->batched && S == 0) return batched_backward_raster(c, d, first, count, main);
  if (first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  for (int i = first; i < first + count && i < (int)ex->range_bwd_main.size(); ++i) ex->range_bwd_main[i] = 0;
  if (ex->batched) {
    const int si = ex->range_stream[first] >= 0 ? ex->range_stream[first] : 0;
    if (si >= S) return DIMO_E_ARG;  // a range on the caller's stream: use the joint launch
    hipStream_t s = ex->streams[si];
    int rc = fork_one(ex, main, s);
    if (!rc) rc = batched_backward_raster(c, d, first, count, s);
    if (rc) return rc;
    return mark_done(ex, si, ex->render_done[first], ex->render_val, first);
  }
  int rc = fork_from_main(ex, main);
  if (rc) return rc;
  for (int i = first + count - 1; i >= first; --i) {
    hipStream_t s = ex->streams[i % S];
    const dimo_render_desc &r = d[i];
    rc = dimo_raster_backward(c->N, 0, 1, c->H, c->W, c->R_cap, r.pts, c->f_dc, nullptr, r.opac, r.scales, r.rot,
                              nullptr, c->scale_modifier, r.view, r.proj, r.campos, c->bg, r.tanfovx, r.tanfovy,
                              r.radii, r.geom, r.bin, r.img, r.g_color, r.g_depth,
                              c->with_normal ? r.g_normal : nullptr, r.g_alpha, r.g_means3D, r.g_means2D, r.g_shs,
                              nullptr, r.g_opac, r.g_scales, r.g_rot, nullptr, r.bwd_scratch, c->bwd_scratch_bytes, s);
    if (rc) return rc;
    if (hipEventRecord(ex->render_done[i], s) != hipSuccess) return DIMO_E_LAUNCH;
  }
  return DIMO_OK;
}

// Batched ranges only.  The private stream the range starting at render `first` runs on (null without one): the
// caller may enqueue that range's loss kernels THERE, behind its forward, instead of joining the caller's stream --
// a motion's whole chain (forward, losses, rasterizer backward) then needs no cross-stream event until the skinning
// backward, and dimo_executor_backward_launch_in_order continues it.
extern "C" void *dimo_executor_range_stream(void *h, int first) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !ex->batched || ex->streams.empty() || first < 0 || first >= (int)ex->range_stream.size()) return nullptr;
  const int si = ex->range_stream[first];
  return si >= 0 && si < (int)ex->streams.size() ? (void *)ex->streams[si] : nullptr;
}
extern "C" int dimo_executor_backward_launch_in_order(void *h, const dimo_step_common *c, int first, int count,
                                                      const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  if (!ex->batched || ex->streams.empty() || first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  const int si = ex->range_stream[first];
  if (si < 0 || si > (int)ex->streams.size()) return DIMO_E_ARG;
  clear_errors();
  if (si == (int)ex->streams.size()) {  // the range's chain is on the caller's stream: so is its backward
    ex->range_bwd_main[first] = 1;
    return batched_backward_raster(c, d, first, count, (hipStream_t)main_stream);
  }
  ex->range_bwd_main[first] = 0;
  hipStream_t s = ex->streams[si];
  const int rc = batched_backward_raster(c, d, first, count, s);
  if (rc) return rc;
  return mark_done(ex, si, ex->render_done[first], ex->render_val, first);
}

// Batched ranges only, after dimo_executor_backward_launch_in_order of the same range: the range's SKINNING backward
// (per-Gaussian gradients in place in the group leaders' buffers, control-point sums as atomics on the shared
// gradient words) in order on the same stream.  dimo_executor_backward_accumulate over the step's renders
// then only folds -- one launch on the caller's stream instead of one skinning backward per motion there (57 + 46 us
// serial on the step's critical path at the benchmark size).
extern "C" int dimo_executor_backward_skinning_in_order(void *h, const dimo_step_common *c, int first, int count,
                                                        const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  if (!ex->batched || ex->streams.empty() || first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  const int si = ex->range_stream[first];
  if (si < 0 || si > (int)ex->streams.size() || ex->range_count[first] != count) return DIMO_E_ARG;
  clear_errors();
  const bool on_main = si == (int)ex->streams.size();
  hipStream_t s = on_main ? (hipStream_t)main_stream : ex->streams[si];
  std::vector<std::pair<int, int>> chunks;
  if (!plan_chunks(ex, first, count, chunks)) return DIMO_E_ARG;
  for (const auto &ch : chunks) {
    RenderBatch b;
    fill_batch(b, d + ch.first, ch.second);
    const int rc = lbs_backward_batched(*c, b, ch.second, s, ch.first, 1);
    if (rc) return rc;
  }
  ex->range_skinned[first] = 1;
  if (on_main) return DIMO_OK;
  return mark_done(ex, si, ex->render_done[first], ex->render_val, first);
}

// Batched ranges only.  The rasterizer backward of ALL the ranges inside [first, first + count) as launches over up to
// MAX_BATCH renders on the CALLER's stream, behind whatever their private streams hold now (the ranges' loss kernels):
// the blend backward of eight renders in one launch runs 38 us per render against 45-57 in two overlapping launches
// of four, and alone on the device.
extern "C" int dimo_executor_backward_launch_joint(void *h, const dimo_step_common *c, int first, int count,
                                                   const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  if (!ex->batched || ex->streams.empty() || first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  clear_errors();
  for (int i = first; i < first + count; ++i) {
    const int si = ex->range_stream[i];
    if (si < 0 || si >= (int)ex->streams.size()) continue;  // (a range on the caller's stream is already in order)
    if (mark_done(ex, si, ex->fwd_done[i], ex->fwd_val, i) || wait_done(ex, main, si, ex->fwd_done[i], ex->fwd_val, i))
      return DIMO_E_LAUNCH;
  }
  std::vector<std::pair<int, int>> chunks;
  if (!plan_chunks(ex, first, count, chunks)) return DIMO_E_ARG;
  for (const auto &ch : chunks) {
    const int rc = batched_backward_raster(c, d, ch.first, ch.second, main, true);
    if (rc) return rc;
  }
  // (dimo_executor_backward_accumulate on the same stream follows in order: a wait on an event recorded on that very
  // stream cost a 10 us bubble between the two calls' kernels)
  for (int i = first; i < first + count; ++i) ex->range_bwd_main[i] = 1;
  return DIMO_OK;
}

// Batched ranges only: `stream` waits for what every range inside [first, first + count) last recorded on its private
// stream (its rasterizer backward, or its skinning backward after dimo_executor_backward_skinning_in_order) -- the
// waits of dimo_executor_backward_accumulate without its kernels: when the fold runs on ANOTHER stream (next to the
// TimeNet backward), the caller's stream still has to see the motions' TimeNet-row gradients.
extern "C" int dimo_executor_join_ranges(void *h, int first, int count, void *stream_) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t st = (hipStream_t)stream_;
  if (!ex || first < 0 || count < 0 || !ex->batched) return DIMO_E_ARG;
  if (ex->streams.empty()) return DIMO_OK;
  if (first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  for (int i = first; i < first + count; ++i) {
    const int si = i < (int)ex->range_stream.size() ? ex->range_stream[i] : -1;
    if (si < 0 || si >= (int)ex->streams.size() || ex->range_bwd_main[i]) continue;
    if (ex->streams[si] == st) continue;  // (its own stream: in order)
    if (wait_done(ex, st, si, ex->render_done[i], ex->render_val, i)) return DIMO_E_LAUNCH;
  }
  return DIMO_OK;
}

// On the caller's stream: wait for the rasterizer backward of renders [first, first + count), then the skinning
// backward accumulating into the shared gradient views (g_f_dc += g_shs included).
extern "C" int dimo_executor_backward_accumulate(void *h, const dimo_step_common *c, int first, int count,
                                                 const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || (count > 0 && !d)) return DIMO_E_ARG;
  if (count == 0) return DIMO_OK;
  clear_errors();
  if (ex->batched) {
    if (!ex->streams.empty()) {
      if (first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
      // every range inside [first, first + count) whose rasterizer backward ran on a private stream (plain batched
      // launches record their event at `first`)
      for (int i = first; i < first + count; ++i) {
        const bool start = i == first || (i < (int)ex->range_stream.size() && ex->range_stream[i] >= 0);
        if (!start || ex->range_bwd_main[i]) continue;
        if (i != first && ex->range_stream[i] >= (int)ex->streams.size()) continue;
        int si = i < (int)ex->range_stream.size() ? ex->range_stream[i] : -1;
        if (si < 0) si = 0;  // (dimo_executor_backward_launch on a render that starts no range used stream 0)
        if (si < (int)ex->streams.size()) {
          if (wait_done(ex, main, si, ex->render_done[i], ex->render_val, i)) return DIMO_E_LAUNCH;
        } else if (hipStreamWaitEvent(main, ex->render_done[i], 0) != hipSuccess) {
          return DIMO_E_LAUNCH;
        }
      }
    }
    std::vector<std::pair<int, int>> chunks;
    if (!plan_chunks(ex, first, count, chunks)) return DIMO_E_ARG;
    for (const auto &ch : chunks) {
      // every range of the chunk skinned already (phase 1, on its own stream): only the fold is left; none: the whole
      // skinning backward here.  (A chunk is one range or several WHOLE ranges.)
      // (a chunk inside a long range belongs to the range that starts before it)
      int skinned = 0, starts = 0;
      const int nrs = (int)ex->range_stream.size();
      int rs = ch.first < nrs ? ch.first : nrs - 1;
      while (rs >= 0 && !(ex->range_stream[rs] >= 0 && rs + ex->range_count[rs] > ch.first)) --rs;
      if (rs >= 0 && rs < ch.first) ++starts, skinned += ex->range_skinned[rs] ? 1 : 0;
      for (int i = ch.first; i < ch.first + ch.second && i < nrs; ++i)
        if (ex->range_stream[i] >= 0) ++starts, skinned += ex->range_skinned[i] ? 1 : 0;
      if (skinned != 0 && skinned != starts) return DIMO_E_ARG;
      RenderBatch b;
      fill_batch(b, d + ch.first, ch.second);
      const int rc = lbs_backward_batched(*c, b, ch.second, main, ch.first, skinned ? 2 : 0);
      if (rc) return rc;
    }
    std::fill(ex->side_pending.begin(), ex->side_pending.end(), 0);  // (every chain that waited for it has been joined)
    return DIMO_OK;
  }
  if (first + count > (int)ex->render_done.size()) return DIMO_E_ARG;
  const size_t n_dc = (size_t)c->N * 3;
  for (int i = first + count - 1; i >= first; --i) {
    const dimo_render_desc &r = d[i];
    if (hipStreamWaitEvent(main, ex->render_done[i], 0) != hipSuccess) return DIMO_E_LAUNCH;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((n_dc + 255) / 256)), dim3(256), 0, main, n_dc, c->g_f_dc,
                       r.g_shs);
    int rc = dimo_deform_backward(c->N, c->M, c->local_frame, 1, c->xyz, c->rotation, c->scaling, c->opacity,
                                  c->c_xyz, c->c_log_radius, r.d_xyz, r.d_rot, c->nn_dist, c->nn_idx, r.g_means3D,
                                  r.g_rot, r.g_scales, r.g_opac, c->g_xyz, c->g_rotation, c->g_scaling, c->g_opacity,
                                  c->g_c_xyz, c->g_c_log_radius, r.g_d_xyz, r.g_d_rot, c->lbs_scratch,
                                  c->lbs_scratch_bytes, main);
    if (rc) return rc;
  }
  return check_launch();
}
