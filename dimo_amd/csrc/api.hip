// Workspace size / layout queries and library identification (host only).
#include "common.hpp"

using namespace dimo;

extern "C" const char *dimo_version(void) { return "dimo_hip gfx950 0.1"; }

extern "C" size_t dimo_raster_geom_bytes(int N) { return GeomLayout(N).bytes; }
extern "C" size_t dimo_raster_bin_bytes(int64_t R_cap, int H, int W) { return BinLayout(R_cap, H, W).bytes; }
extern "C" size_t dimo_raster_img_bytes(int H, int W) { return ImgLayout(H, W).bytes; }

extern "C" int dimo_raster_geom_layout(int N, size_t out[6]) {
  if (!out || N < 0) return DIMO_E_ARG;
  GeomLayout L(N);
  out[0] = L.splat, out[1] = L.rect, out[2] = L.tiles, out[3] = L.offsets, out[4] = L.flags, out[5] = L.total;
  return DIMO_OK;
}
extern "C" int dimo_raster_bin_layout(int64_t R_cap, int H, int W, size_t out[5]) {
  if (!out || R_cap < 0 || H <= 0 || W <= 0) return DIMO_E_ARG;
  BinLayout L(R_cap, H, W);
  out[0] = L.keys_a, out[1] = L.vals_a, out[2] = L.keys_b, out[3] = L.vals_b, out[4] = L.ranges;
  return DIMO_OK;
}
extern "C" int dimo_raster_img_layout(int H, int W, size_t out[2]) {
  if (!out || H <= 0 || W <= 0) return DIMO_E_ARG;
  ImgLayout L(H, W);
  out[0] = L.final_T, out[1] = L.n_contrib;
  return DIMO_OK;
}
