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


// Wave reduction of 16 values per lane in halving steps: at every step a lane exchanges, with the lane whose number
// differs in one bit, the half of its values the partner is going to keep, so a step costs half of the previous one.
// Priced with the measured gfx950 issue rates (profiles/r02_valu_issue_rates.txt: v_cndmask / any DPP form 4 cycles
// per wave64 instruction, v_add 2, v_permlane*_swap 8) the order of the bits matters:
//   lane ^ 8   16 -> 8   v_add_f32_dpp row_ror:8, bank_mask selects the banks that keep the low / the high value:
//   lane ^ 4    8 -> 4   (row_shl:4 / row_shr:4)    TWO masked DPP adds per output, no v_cndmask      64 + 32 cycles
//   lane ^ 32   4 -> 2   v_permlane32_swap + v_add  (a swap moves both halves at once)                      20
//   lane ^ 16   2 -> 1   v_permlane16_swap + v_add                                                          10
//   lane ^ 1, lane ^ 2   plain quad_perm DPP adds on the one remaining value                                 8
// = 134 cycles.  (Round 1 halved over the low bits first, where only quad_perm reaches and every output costs two
// v_cndmask + a DPP add: 216 cycles.)  Lane l ends with the WAVE total of value reduce16_slot(l); the four lanes
// l & ~3 .. l | 3 hold the same one, so the 16 lanes with (l & 3) == 0 store the 16 totals.
// All in one asm block, in place on v[0..15]: the hazard recogniser does not look into inline asm, so the wait
// states (VALU write -> DPP / permlane read: 2) are spelled out where fewer than two instructions separate them.
__device__ __forceinline__ int reduce16_slot(int lane) {
  return 8 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1) + 2 * ((lane >> 5) & 1) + ((lane >> 4) & 1);
}
// NV = number of values in use (v[0 .. NV - 1]; 16, 13 or 10): the first halving step pairs value s with value s + 8,
// and where s + 8 >= NV there is nothing to fetch -- the lanes that would keep value s + 8 then hold garbage, which
// stays among the lanes of the unused slots (a step only ever combines lanes of the same slot); the caller must not
// read those slots.  Three (13) or six (10) DPP adds of 4 issue cycles fewer per reduction (134 -> 122 / 110 cycles).
// The blend backward uses 13 for both rasterizer flavours: its records carry 13 sums, and the 4-output flavour's three
// normal sums must read ZERO downstream (with 10 they were garbage: found by the 4-output parity test).
template <int NV = 16>
__device__ __forceinline__ float wave_reduce16(float (&v)[16]) {
  static_assert(NV == 16 || NV == 13 || NV == 10, "value counts the callers use");
#define DIMO_RA1(s) "v_add_f32_dpp %" #s ", %" #s ", %" #s " row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
#define DIMO_RA(s, h) DIMO_RA1(s) "v_add_f32_dpp %" #s ", %" #h ", %" #h " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define DIMO_RB(t, u) "v_add_f32_dpp %" #t ", %" #t ", %" #t " row_shl:4 row_mask:0xf bank_mask:0x5\n\t" \
                      "v_add_f32_dpp %" #t ", %" #u ", %" #u " row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
#define DIMO_RTAIL                                                                                      \
      DIMO_RB(0, 4) DIMO_RB(1, 5) DIMO_RB(2, 6) DIMO_RB(3, 7)                                           \
      "s_nop 1\n\t"                                                                                     \
      "v_permlane32_swap_b32 %0, %2\n\t"                                                                \
      "v_permlane32_swap_b32 %1, %3\n\t"                                                                \
      "s_nop 1\n\t"                                                                                     \
      "v_add_f32 %0, %0, %2\n\t"                                                                        \
      "v_add_f32 %1, %1, %3\n\t"                                                                        \
      "s_nop 1\n\t"                                                                                     \
      "v_permlane16_swap_b32 %0, %1\n\t"                                                                \
      "s_nop 1\n\t"                                                                                     \
      "v_add_f32 %0, %0, %1\n\t"                                                                        \
      "s_nop 1\n\t"                                                                                     \
      "v_add_f32_dpp %1, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                     \
      "s_nop 1\n\t"                                                                                     \
      "v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
#define DIMO_ROPS                                                                                                 \
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])           \
      : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15])
  if (NV == 16) {
    asm volatile("s_nop 1\n\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA(2, 10) DIMO_RA(3, 11) DIMO_RA(4, 12) DIMO_RA(5, 13)
                 DIMO_RA(6, 14) DIMO_RA(7, 15) DIMO_RTAIL DIMO_ROPS);
  } else if (NV == 13) {
    asm volatile("s_nop 1\n\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA(2, 10) DIMO_RA(3, 11) DIMO_RA(4, 12) DIMO_RA1(5)
                 DIMO_RA1(6) DIMO_RA1(7) DIMO_RTAIL DIMO_ROPS);
  } else {
    asm volatile("s_nop 1\n\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA1(2) DIMO_RA1(3) DIMO_RA1(4) DIMO_RA1(5) DIMO_RA1(6)
                 DIMO_RA1(7) DIMO_RTAIL DIMO_ROPS);
  }
#undef DIMO_RA1
#undef DIMO_RA
#undef DIMO_RB
#undef DIMO_RTAIL
#undef DIMO_ROPS
  return v[0];
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

// table[idx * stride + c] += sum over the wave's lanes with that idx of v[c], c < 8 -- by MATCHING: the lanes of one
// index are found with a ballot, two indices' values share one 16-value halving reduction, and eight lanes issue the
// adds.  For a wave whose lanes hold FEW distinct indices (Morton-ordered Gaussians: 2-4 control points per neighbour
// slot) this is ~230 VALU cycles per pair of indices, where one LDS float atomic per lane and value costs the CU's single
// LDS atomic unit ~4 cycles per active lane (the skinning backward was bound by exactly that).
constexpr int MATCH_ROUNDS = 3;  // pairs of indices matched before the remaining lanes fall back to plain atomics
__device__ __forceinline__ void wave_scatter_add_match8(float *table, int stride, int idx, const float (&v)[8],
                                                        bool valid, int lane) {
  unsigned long long rem = __ballot(valid);
  for (int round = 0; rem != 0ull; ++round) {
    if (round == MATCH_ROUNDS) {  // many distinct indices (unordered Gaussians): the rest by run-combined atomics
      wave_scatter_add<8>(table, stride, idx, v, (rem >> lane) & 1ull, lane);
      return;
    }
    const int A = __builtin_amdgcn_readlane(idx, __ffsll((long long)rem) - 1);
    const bool inA = valid && idx == A;
    const unsigned long long mA = __ballot(inA);
    rem &= ~mA;
    int B = A;
    bool inB = false;
    unsigned long long mB = 0ull;
    if (rem != 0ull) {  // (wave-uniform)
      B = __builtin_amdgcn_readlane(idx, __ffsll((long long)rem) - 1);
      inB = valid && idx == B;
      mB = __ballot(inB);
      rem &= ~mB;
    }
    float r[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = inA ? v[c] : 0.0f, r[8 + c] = inB ? v[c] : 0.0f;
    const float tot = wave_reduce16(r);
    const int s = reduce16_slot(lane);
    if ((lane & 3) == 0 && (s < 8 || mB != 0ull)) atomicAdd(table + (s < 8 ? A : B) * stride + (s & 7), tot);
  }
}

}  // namespace dimo
