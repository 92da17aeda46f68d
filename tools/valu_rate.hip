// Issue-rate probe for gfx950 VALU instruction classes (standalone: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip
// -o /tmp/valu_rate && /tmp/valu_rate).  Every wave runs ITERS x 32 independent instructions of one class over
// 8 accumulators; with W waves per SIMD resident on all 1024 SIMDs the printed figure is cycles per wave64
// instruction per SIMD (clock taken from the device's wall clock counter rate and the measured kernel time).
// Used to decide which counters price the blend kernels' VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2000;

#define BODY8(INS)                                                                                          \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                      \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                      \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                      \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                      \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
               : "v"(x), "v"(y))

#define I_FMA(k) "v_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define I_MUL(k) "v_mul_f32 %" #k ", %8, %" #k "\n\t"
#define I_ADD(k) "v_add_f32 %" #k ", %8, %" #k "\n\t"
#define I_MAC(k) "v_fmac_f32 %" #k ", %8, %9\n\t"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n\t"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define I_DPP(k) "v_mov_b32_dpp %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_ADDDPP(k) "v_add_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_CND(k) "v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n\t"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\t"
#define I_IADD(k) "v_add_u32 %" #k ", %8, %" #k "\n\t"
#define I_MAX(k) "v_max_f32 %" #k ", %8, %" #k "\n\t"
#define I_LSH(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
#define I_MIN(k) "v_min_f32 %" #k ", %8, %" #k "\n\t"
#define I_SUB(k) "v_sub_f32 %" #k ", %8, %" #k "\n\t"
#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k "\n\t"
#define I_OR(k) "v_or_b32 %" #k ", %8, %" #k "\n\t"
#define I_CVT(k) "v_cvt_f32_u32 %" #k ", %" #k "\n\t"
#define I_CNDS(k) "v_cndmask_b32 %" #k ", %8, %" #k ", s[10:11]\n\t"
#define I_CMPS(k) "v_cmp_lt_f32 s[10:11], %" #k ", %8\n\t"
#define I_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\tv_cndmask_b32 %" #k ", %8, %" #k ", vcc\n\t"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %8, %" #k "\n\t"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %8, %9, %" #k "\n\t"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 1, 5\n\t"
#define I_FMAK(k) "v_fma_f32 %" #k ", %" #k ", 0.5, %9\n\t"
#define I_FMANEG(k) "v_fma_f32 %" #k ", -%8, %9, %" #k "\n\t"
#define I_MULS(k) "v_mul_f32 %" #k ", s12, %" #k "\n\t"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n\t"
#define I_LOG(k) "v_log_f32 %" #k ", %" #k "\n\t"
#define I_RFL(k) "v_readfirstlane_b32 s12, %" #k "\n\t"
#define I_ROWSHR(k) "v_mov_b32_dpp %" #k ", %" #k " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_WAVESHR(k) "v_mov_b32_dpp %" #k ", %" #k " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_PERM(k) "v_permlane32_swap_b32 %" #k ", %8\n\t"

#define KERNEL(NAME, INS)                                                                      \
  __global__ void __launch_bounds__(256) NAME(float *out, float xin, float yin) {              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
          a6 = a0 + 6, a7 = a0 + 7;                                                            \
    const float x = xin, y = yin;                                                              \
    for (int i = 0; i < ITERS; ++i) BODY8(INS);                                                \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
  }

KERNEL(k_fma, I_FMA)
KERNEL(k_mul, I_MUL)
KERNEL(k_add, I_ADD)
KERNEL(k_mac, I_MAC)
KERNEL(k_exp, I_EXP)
KERNEL(k_rcp, I_RCP)
KERNEL(k_mov, I_MOV)
KERNEL(k_dpp, I_DPP)
KERNEL(k_adddpp, I_ADDDPP)
KERNEL(k_cnd, I_CND)
KERNEL(k_cmp, I_CMP)
KERNEL(k_iadd, I_IADD)
KERNEL(k_max, I_MAX)
KERNEL(k_lsh, I_LSH)
KERNEL(k_min, I_MIN)
KERNEL(k_sub, I_SUB)
KERNEL(k_and, I_AND)
KERNEL(k_or, I_OR)
KERNEL(k_cvt, I_CVT)
KERNEL(k_cnds, I_CNDS)
KERNEL(k_cmps, I_CMPS)
KERNEL(k_cmpcnd, I_CMPCND)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_bfe, I_BFE)
KERNEL(k_fmak, I_FMAK)
KERNEL(k_fmaneg, I_FMANEG)
KERNEL(k_muls, I_MULS)
KERNEL(k_sqrt, I_SQRT)
KERNEL(k_log, I_LOG)
KERNEL(k_rfl, I_RFL)
KERNEL(k_rowshr, I_ROWSHR)
KERNEL(k_waveshr, I_WAVESHR)

// permlane swaps: four independent register pairs, 32 swaps per iteration (operands last written 4 swaps earlier)
#define SWAP4(OP) OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t"
#define SWAPBODY(OP)                                                                                       \
  asm volatile(SWAP4(OP) SWAP4(OP) SWAP4(OP) SWAP4(OP) SWAP4(OP) SWAP4(OP) SWAP4(OP) SWAP4(OP)             \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define SWAPKERNEL(NAME, OP)                                                                   \
  __global__ void __launch_bounds__(256) NAME(float *out, float xin, float yin) {              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
          a6 = a0 + 6, a7 = a0 + 7;                                                            \
    for (int i = 0; i < ITERS; ++i) SWAPBODY(OP);                                              \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + xin + yin;  \
  }
SWAPKERNEL(k_swap32, "v_permlane32_swap_b32")
SWAPKERNEL(k_swap16, "v_permlane16_swap_b32")

// packed fp32: 4 accumulator pairs
__global__ void __launch_bounds__(256) k_pkfma(float *out, float xin, float yin) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
     a6 = a0 + 6.f, a7 = a0 + 7.f;
  const f2 x = {xin, xin}, y = {yin, yin};
#define I_PK(k) "v_pk_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
  for (int i = 0; i < ITERS; ++i) BODY8(I_PK);
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <class K>
static void run(const char *name, K kern, int waves_per_simd, float *out, double ghz) {
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = one per SIMD of a CU
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double insts_per_simd = (double)waves_per_simd * ITERS * 32;
  printf("%-10s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at %.2f GHz)\n", name,
         waves_per_simd, ms, ms * 1e-3 * ghz * 1e9 / insts_per_simd, ghz);
}

int main() {
  float *out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  int khz = 0;
  CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz * 1e-6;
  for (int w : {1, 4}) {
    run("fma", k_fma, w, out, ghz);
    run("mul", k_mul, w, out, ghz);
    run("add", k_add, w, out, ghz);
    run("fmac", k_mac, w, out, ghz);
    run("pk_fma", k_pkfma, w, out, ghz);
    run("exp", k_exp, w, out, ghz);
    run("rcp", k_rcp, w, out, ghz);
    run("mov", k_mov, w, out, ghz);
    run("mov_dpp", k_dpp, w, out, ghz);
    run("add_dpp", k_adddpp, w, out, ghz);
    run("cndmask", k_cnd, w, out, ghz);
    run("cmp", k_cmp, w, out, ghz);
    run("iadd", k_iadd, w, out, ghz);
    run("max", k_max, w, out, ghz);
    run("lshl", k_lsh, w, out, ghz);
    run("min", k_min, w, out, ghz);
    run("sub", k_sub, w, out, ghz);
    run("and", k_and, w, out, ghz);
    run("or", k_or, w, out, ghz);
    run("cvt_f32_u32", k_cvt, w, out, ghz);
    run("cndmask_sgpr", k_cnds, w, out, ghz);
    run("cmp_sgpr", k_cmps, w, out, ghz);
    run("cmp+cnd(x2)", k_cmpcnd, w, out, ghz);
    run("mul_lo_u32", k_mullo, w, out, ghz);
    run("mad_u32_u24", k_mad24, w, out, ghz);
    run("bfe", k_bfe, w, out, ghz);
    run("fma_const", k_fmak, w, out, ghz);
    run("fma_neg", k_fmaneg, w, out, ghz);
    run("mul_sgpr", k_muls, w, out, ghz);
    run("sqrt", k_sqrt, w, out, ghz);
    run("log", k_log, w, out, ghz);
    run("readfirstlane", k_rfl, w, out, ghz);
    run("dpp_row_shr1", k_rowshr, w, out, ghz);
    run("dpp_wave_shr1", k_waveshr, w, out, ghz);
    run("permlane32_swap", k_swap32, w, out, ghz);
    run("permlane16_swap", k_swap16, w, out, ghz);
  }
  return 0;
}
