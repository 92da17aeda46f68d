// Native step executor: runs the per-render kernel chains of one training step (skinning -> projection ->
// binning/sort -> blend, and the mirrored backward) from ONE host call each, fanning the independent renders
// out over a few private HIP streams and joining them with events against the caller's stream.
//
// Why native: after the kernels were fused, a step at 100k Gaussians / 512^2 spent ~70 % of its wall time in
// the Python interpreter (ctypes marshalling, allocator calls, stream context managers: ~0.4 ms per render),
// with the GPU idle.  All buffers are caller-owned and persistent (288 GB of HBM: eight render slots cost
// < 2 GB), so a step is a fixed sequence of launches that needs no host logic in between.
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
} dimo_render_desc;
}
#endif

namespace dimo {

struct Executor {
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> stream_done;
  std::vector<hipEvent_t> render_done;
  std::vector<hipEvent_t> fwd_done;
  hipEvent_t main_ready = nullptr;
};

__global__ void __launch_bounds__(256) accumulate_kernel(size_t n, float *__restrict__ dst,
                                                         const float *__restrict__ src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

}  // namespace dimo

using namespace dimo;

extern "C" void *dimo_executor_create(int n_streams) {
  if (n_streams < 1 || n_streams > 16) return nullptr;
  Executor *ex = new Executor();
  for (int i = 0; i < n_streams; ++i) {
    hipStream_t s;
    hipEvent_t e;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      delete ex;
      return nullptr;
    }
    ex->streams.push_back(s);
    ex->stream_done.push_back(e);
  }
  if (hipEventCreateWithFlags(&ex->main_ready, hipEventDisableTiming) != hipSuccess) {
    delete ex;
    return nullptr;
  }
  return ex;
}

extern "C" void dimo_executor_destroy(void *h) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  if (!ex) return;
  for (auto s : ex->streams) (void)hipStreamDestroy(s);
  for (auto e : ex->stream_done) (void)hipEventDestroy(e);
  for (auto e : ex->render_done) (void)hipEventDestroy(e);
  for (auto e : ex->fwd_done) (void)hipEventDestroy(e);
  if (ex->main_ready) (void)hipEventDestroy(ex->main_ready);
  delete ex;
}

static int fork_from_main(Executor *ex, hipStream_t main) {
  if (hipEventRecord(ex->main_ready, main) != hipSuccess) return DIMO_E_LAUNCH;
  for (auto s : ex->streams)
    if (hipStreamWaitEvent(s, ex->main_ready, 0) != hipSuccess) return DIMO_E_LAUNCH;
  return DIMO_OK;
}

extern "C" int dimo_executor_forward(void *h, const dimo_step_common *c, int n, const dimo_render_desc *d,
                                     void *main_stream) {
  // Launches the forward chain of renders [0, n) on the private streams (render i on stream i % S) and records
  // one event per render.  Does NOT make the caller's stream wait: see dimo_executor_join.
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || n < 0 || (n > 0 && !d)) return DIMO_E_ARG;
  while ((int)ex->render_done.size() < n || (int)ex->fwd_done.size() < n) {
    hipEvent_t e, f;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess)
      return DIMO_E_LAUNCH;
    ex->render_done.push_back(e);
    ex->fwd_done.push_back(f);
  }
  int rc = fork_from_main(ex, main);
  if (rc) return rc;
  const int S = (int)ex->streams.size();
  for (int i = 0; i < n; ++i) {
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

// The caller's stream waits for the forward of renders [first, first + count).
extern "C" int dimo_executor_join(void *h, int first, int count, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || first < 0 || count < 0 || first + count > (int)ex->fwd_done.size()) return DIMO_E_ARG;
  for (int i = first; i < first + count; ++i)
    if (hipStreamWaitEvent(main, ex->fwd_done[i], 0) != hipSuccess) return DIMO_E_LAUNCH;
  return DIMO_OK;
}

// Rasterizer backward of renders [first, first + count) on their private streams (ordered after everything the
// caller's stream has enqueued so far, i.e. after the kernels that produced the gradient images).
extern "C" int dimo_executor_backward_launch(void *h, const dimo_step_common *c, int first, int count,
                                             const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || first + count > (int)ex->render_done.size() || (count > 0 && !d))
    return DIMO_E_ARG;
  int rc = fork_from_main(ex, main);
  if (rc) return rc;
  const int S = (int)ex->streams.size();
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

// On the caller's stream, in render order: wait for render i's rasterizer backward, then g_f_dc += g_shs and the
// skinning backward accumulating into the shared gradient views.
extern "C" int dimo_executor_backward_accumulate(void *h, const dimo_step_common *c, int first, int count,
                                                 const dimo_render_desc *d, void *main_stream) {
  Executor *ex = reinterpret_cast<Executor *>(h);
  hipStream_t main = (hipStream_t)main_stream;
  if (!ex || !c || first < 0 || count < 0 || first + count > (int)ex->render_done.size() || (count > 0 && !d))
    return DIMO_E_ARG;
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
