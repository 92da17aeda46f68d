// Wave64 cross-lane reductions shared by the blend and skinning backward kernels (gfx950).
#pragma once
#include "common.hpp"

namespace dimo {

// DPP helpers.  update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lane l reads src of the lane the
// control selects; lanes whose source is out of range keep `old` (= 0 here).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;


// Wave reduction of 16 values per lane in halving steps: at step s the lanes whose bit s differs exchange the half
// of their values the partner is going to keep, so every step costs half of the previous one
//   lane ^ 1 (quad_perm)  16 -> 8 values per lane      lane ^ 2 (quad_perm)  8 -> 4
//   lane ^ 4 (row_shl/shr:4 on alternating banks) 4 -> 2    lane ^ 8 (row_ror:8)  2 -> 1
// and the last value is folded over the four 16-lane rows with gfx950's v_permlane16_swap / v_permlane32_swap.
// Every lane ends up with the WAVE total of value  8 b0 + 4 b1 + 2 b2 + b3  (b_i = bit i of its lane number):
// lanes 0..15 hold the 16 totals, so ONE 16-lane conflict-free LDS add stores them.  ~60 VALU instead of the
// 16 x 6 of a plain butterfly (a first version stopped halving after two steps and finished four values per
// lane: 76 VALU, eight permlane swaps with their s_nop padding, four LDS adds).
constexpr int DPP_ROW_SHL4 = 0x104;
constexpr int DPP_ROW_ROR8 = 0x128;
__device__ __forceinline__ float dpp_xor4(float v) {
  const int x = __builtin_bit_cast(int, v);
  int t = __builtin_amdgcn_update_dpp(0, x, DPP_ROW_SHL4, 0xf, 0x5, false);  // banks 0, 2 read lane + 4
  t = __builtin_amdgcn_update_dpp(t, x, DPP_ROW_SHR4, 0xf, 0xa, false);      // banks 1, 3 read lane - 4
  return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ int butterfly16_slot(int lane) {
  return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
}
__device__ __forceinline__ float butterfly16(const float (&v)[16], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float h[8], q[4], p[2];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float keep = b0 ? v[8 + s] : v[s];
    const float send = b0 ? v[s] : v[8 + s];
    h[s] = keep + dpp_mov<DPP_QUAD_XOR1>(send);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float keep = b1 ? h[4 + t] : h[t];
    const float send = b1 ? h[t] : h[4 + t];
    q[t] = keep + dpp_mov<DPP_QUAD_XOR2>(send);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float keep = b2 ? q[2 + u] : q[u];
    const float send = b2 ? q[u] : q[2 + u];
    p[u] = keep + dpp_xor4(send);
  }
  float r = (b3 ? p[1] : p[0]) + dpp_mov<DPP_ROW_ROR8>(b3 ? p[0] : p[1]);  // row total of this lane's value
  float a = r, b = r;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  r = a + b;  // rows (0,1) and (2,3) summed, replicated
  a = r, b = r;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;  // all four rows
}


// Scatter-add of NV per-lane values into row `idx` of an LDS table (row stride `stride` floats).
// LDS float atomics retire ~one LANE every three cycles whatever the addresses are (SQ_LDS_IDX_ACTIVE / lane-atomics
// measured 3.1 on gfx950), so the cost of a scatter is the number of lanes that issue an add.  Gaussians listed next
// to each other share their control points: inside every 16-lane DPP row, RUNS of consecutive lanes naming the same
// table row are summed with a segmented scan (4 row_shr steps, masks shared by all NV values) and only the last
// lane of each run adds.  Any index pattern is handled (a run may be a single lane); runs do not cross DPP rows.
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHL1 = 0x101;
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v, int oob) {
  return __builtin_amdgcn_update_dpp(oob, v, CTRL, 0xf, 0xf, false);
}
template <int NV>
__device__ __forceinline__ void wave_scatter_add(float *table, int stride, int idx, const float (&v)[NV], bool valid,
                                                 int lane) {
  const int pos = lane & 15;
  const int key = valid ? idx : -1 - lane;  // an invalid lane is a run of its own that never adds
  const bool head = dpp_mov_i<DPP_ROW_SHR1>(key, ~key) != key;
  const bool last = dpp_mov_i<DPP_ROW_SHL1>(key, ~key) != key;
  int rs = head ? pos : 0;  // position of the head of this lane's run: inclusive max-scan of the head positions
  rs = max(rs, dpp_mov_i<DPP_ROW_SHR1>(rs, 0));
  rs = max(rs, dpp_mov_i<DPP_ROW_SHR2>(rs, 0));
  rs = max(rs, dpp_mov_i<DPP_ROW_SHR4>(rs, 0));
  rs = max(rs, dpp_mov_i<DPP_ROW_SHR8>(rs, 0));
  const float m1 = pos - 1 >= rs ? 1.0f : 0.0f, m2 = pos - 2 >= rs ? 1.0f : 0.0f;
  const float m4 = pos - 4 >= rs ? 1.0f : 0.0f, m8 = pos - 8 >= rs ? 1.0f : 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float x = valid ? v[k] : 0.0f;
    x = fmaf(dpp_mov<DPP_ROW_SHR1>(x), m1, x);
    x = fmaf(dpp_mov<DPP_ROW_SHR2>(x), m2, x);
    x = fmaf(dpp_mov<DPP_ROW_SHR4>(x), m4, x);
    x = fmaf(dpp_mov<DPP_ROW_SHR8>(x), m8, x);
    if (last && valid) atomicAdd(table + idx * stride + k, x);
  }
}

}  // namespace dimo
