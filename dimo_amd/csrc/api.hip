// Workspace size / layout queries and library identification (host only).
#include "common.hpp"

using namespace dimo;

extern "C" const char *dimo_version(void) { return "dimo_hip gfx950 0.1"; }

extern "C" size_t dimo_raster_geom_bytes(int N) { return GeomLayout(N).bytes; }
extern "C" size_t dimo_raster_bin_bytes(int N, int64_t R_cap, int H, int W) { return BinLayout(R_cap, H, W, N).bytes; }
extern "C" size_t dimo_raster_img_bytes(int H, int W) { return ImgLayout(H, W).bytes; }

extern "C" int dimo_raster_geom_layout(int N, size_t out[6]) {
  if (!out || N < 0) return DIMO_E_ARG;
  GeomLayout L(N);
  out[0] = L.splat, out[1] = L.rect, out[2] = L.tiles, out[3] = L.offsets, out[4] = L.flags, out[5] = L.total;
  return DIMO_OK;
}
extern "C" int dimo_raster_bin_layout(int64_t R_cap, int H, int W, size_t out[3]) {
  if (!out || R_cap < 0 || H <= 0 || W <= 0) return DIMO_E_ARG;
  BinLayout L(R_cap, H, W, 1);  // (the three inspectable arrays lead the workspace: their offsets do not depend on N)
  out[0] = L.vals_b, out[1] = L.ranges, out[2] = L.totals;
  return DIMO_OK;
}
extern "C" int dimo_raster_depth_keys(int N, int H, int W, int64_t R_cap, const void *geom, const void *bin, uint32_t *out,
                                      void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || R_cap < 0 || !geom || !bin || !out) return DIMO_E_ARG;
  clear_errors();
  return instance_depth_keys(N, H, W, R_cap, geom, bin, out, (hipStream_t)stream);
}
extern "C" int dimo_debug_bin_geom_layout(int N, int H, int W, size_t out[10]) {
  if (!out || N < 0 || H <= 0 || W <= 0) return DIMO_E_ARG;
  GeomLayout G(N);
  BinGrid gi;
  if (!make_bin_grid(H, W, gi)) return DIMO_E_ARG;
  out[0] = G.rect, out[1] = G.tiles, out[2] = G.offsets, out[3] = G.total, out[4] = G.block_sums, out[5] = G.key32;
  out[6] = G.bk, out[7] = G.bytes, out[8] = (size_t)G.nb, out[9] = (size_t)gi.ss_shift;
  return DIMO_OK;
}
extern "C" int dimo_debug_bin_instances(int N, int H, int W, int64_t R_cap, void *geom, void *bin, void *stream) {
  if (N < 0 || H <= 0 || W <= 0 || R_cap < 0 || R_cap > 0xfffffff0LL || !geom || !bin) return DIMO_E_ARG;
  clear_errors();
  return bin_instances(N, H, W, R_cap, geom, bin, (hipStream_t)stream);
}
extern "C" int dimo_raster_img_layout(int H, int W, size_t out[2]) {
  if (!out || H <= 0 || W <= 0) return DIMO_E_ARG;
  ImgLayout L(H, W);
  out[0] = L.final_T, out[1] = L.n_contrib;
  return DIMO_OK;
}

// ------------------------------------------------------------------------------------------------
// Optional kernel timing with HIP events, recorded on the very stream each kernel is launched on.
// This is the one piece of process-global state in the library; it is OFF by default and is meant
// for bench.py's live roofline figure (the rocprofv3 summaries under profiles/ must agree with it).
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace dimo {
namespace {
struct Rec {
  int id;
  hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
uint32_t g_mask = ~0u;            // groups being timed while g_on
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;   // events are recycled: creating two per launch costs more than the record
bool take_event(hipEvent_t *e) {
  if (!g_free.empty()) {
    *e = g_free.back();
    g_free.pop_back();
    return true;
  }
  return hipEventCreate(e) == hipSuccess;
}
const char *const g_names[TIMED_COUNT] = {"preprocess_fwd", "scan", "emit",     "sort",     "ranges",  "blend_fwd",
                                          "blend_bwd",      "preprocess_bwd", "knn", "dist2", "ssim_fwd", "ssim_bwd",
                                          "deform_fwd",     "deform_bwd", "image_loss", "adam", "timenet_fwd", "timenet_bwd", "place"};
}  // namespace

ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on || !((g_mask >> id) & 1u)) return;
  if (!take_event(&a_)) {
    a_ = nullptr;
    return;
  }
  if (!take_event(&b_)) {
    g_free.push_back(a_);
    a_ = b_ = nullptr;
    return;
  }
  (void)hipEventRecord(a_, stream_);
}
ScopedTimer::~ScopedTimer() {
  if (!a_) return;
  (void)hipEventRecord(b_, stream_);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back({id_, a_, b_});
}
}  // namespace dimo

extern "C" int dimo_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int prev = g_on ? 1 : 0;
  if (on && !g_on) {
    for (auto &r : g_recs) g_free.push_back(r.a), g_free.push_back(r.b);
    g_recs.clear();
  }
  g_on = on != 0;
  return prev;
}

extern "C" int dimo_timing_select(const char *names) {
  uint32_t mask = 0;
  if (!names || !*names) {
    mask = ~0u;
  } else {
    std::string all(names);
    size_t pos = 0;
    while (pos <= all.size()) {
      const size_t end = std::min(all.find(',', pos), all.size());
      const std::string one = all.substr(pos, end - pos);
      int id = -1;
      for (int i = 0; i < TIMED_COUNT; ++i)
        if (one == g_names[i]) id = i;
      if (id < 0) return DIMO_E_ARG;
      mask |= 1u << id;
      pos = end + 1;
    }
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_mask = mask;
  return DIMO_OK;
}

extern "C" int dimo_timing_read(const char *name, double *total_ms, int64_t *launches) {
  if (!name || !total_ms || !launches) return DIMO_E_ARG;
  int id = -1;
  for (int i = 0; i < TIMED_COUNT; ++i)
    if (std::string(name) == g_names[i]) id = i;
  if (id < 0) return DIMO_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return DIMO_E_LAUNCH;
  std::lock_guard<std::mutex> lk(g_mu);
  double tot = 0.0;
  int64_t n = 0;
  for (auto &r : g_recs)
    if (r.id == id) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) tot += ms, ++n;
    }
  *total_ms = tot, *launches = n;
  return DIMO_OK;
}

// ------------------------------------------------------------------------------------------------
namespace dimo {
namespace {
thread_local char g_last_error[256] = "";
}
void set_last_error(hipError_t e, const char *where) {
  snprintf(g_last_error, sizeof(g_last_error), "%s (%d) at %s", hipGetErrorString(e), (int)e, where);
}
}  // namespace dimo

extern "C" const char *dimo_last_error(void) { return dimo::g_last_error; }
