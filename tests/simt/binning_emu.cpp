// Host build of dimo_amd/csrc/binning.hip on the SIMT emulation shim (tests/simt/build.py substitutes the two gfx950
// inline-asm statements).  TEST INFRASTRUCTURE ONLY: exports the binning chain to tests/test_binning_emulated.py.
#include "binning_src.inc"

namespace dimo {
void set_last_error(hipError_t, const char *) {}
ScopedTimer::ScopedTimer(int id, hipStream_t s) : id_(id), stream_(s), a_(nullptr), b_(nullptr) {}
ScopedTimer::~ScopedTimer() {}
}  // namespace dimo

extern "C" {
// out: rect, tiles, offsets, total, block_sums, key32, bk, bytes, nb, byte offset of the bucket totals
void simt_geom_layout(int N, size_t out[10]) {
  dimo::GeomLayout G(N);
  out[0] = G.rect, out[1] = G.tiles, out[2] = G.offsets, out[3] = G.total, out[4] = G.block_sums, out[5] = G.key32;
  out[6] = G.bk, out[7] = G.bytes, out[8] = (size_t)G.nb, out[9] = G.bk + dimo::BK_TOT * sizeof(uint32_t);
}
// out: vals, ranges, totals, order, bytes, T, cap, l1tmp, l1list, meta, grpbase, grpinfo, cntu, l1cap
void simt_bin_layout(int N, int64_t R_cap, int H, int W, size_t out[14]) {
  dimo::BinLayout B(R_cap, H, W, N);
  out[0] = B.vals_b, out[1] = B.ranges, out[2] = B.totals, out[3] = B.order, out[4] = B.bytes;
  out[5] = (size_t)B.T, out[6] = B.cap;
  out[7] = B.l1tmp, out[8] = B.l1list, out[9] = B.meta, out[10] = B.grpbase, out[11] = B.grpinfo, out[12] = B.cntu;
  out[13] = B.l1cap;
}
int simt_depth_keys(int N, int H, int W, int64_t R_cap, const void *geom, const void *bin, uint32_t *out) {
  return dimo::instance_depth_keys(N, H, W, R_cap, geom, bin, out, nullptr);
}
int simt_supertile_shift(int H, int W) {
  dimo::BinGrid gi;
  return dimo::make_bin_grid(H, W, gi) ? gi.ss_shift : -1;
}
int simt_bin_instances(int N, int H, int W, int64_t R_cap, void *geom, void *bin) {
  return dimo::bin_instances(N, H, W, R_cap, geom, bin, nullptr);
}
// the batched kernels (blockIdx.y = render): n renders with workspaces geom[i], bin[i], bwd_scratch[i]
int simt_bin_instances_batched(int N, int H, int W, int64_t R_cap, int n, void **geom, void **bin, void **scratch,
                               size_t geom_bytes, size_t bin_bytes, size_t scratch_bytes, uint32_t *totals_out) {
  dimo_step_common c;
  memset(&c, 0, sizeof(c));
  c.N = N, c.H = H, c.W = W, c.R_cap = R_cap, c.geom_bytes = geom_bytes, c.bin_bytes = bin_bytes;
  c.bwd_scratch_bytes = scratch_bytes;
  dimo::RenderBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < n; ++i) {
    b.r[i].geom = geom[i], b.r[i].bin = bin[i], b.r[i].bwd_scratch = scratch[i];
    b.r[i].totals_out = totals_out ? totals_out + 2 * i : nullptr;
  }
  return dimo::bin_instances_batched(c, b, n, nullptr);
}
size_t simt_bwd_scratch_bytes(int64_t R_cap, int H, int W) {
  dimo::BinLayout B(R_cap, H, W, 1);
  return dimo::align_up(B.cap * sizeof(dimo::SplatGrad)) + dimo::align_up(B.cap);
}
}
