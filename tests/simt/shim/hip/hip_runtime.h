// SIMT emulation shim: stands in for <hip/hip_runtime.h> when a .hip source of dimo_amd/csrc is compiled FOR THE HOST
// by tests/simt/build.py.  TEST INFRASTRUCTURE ONLY: it lets the CPU test suite run the integer logic of the binning
// kernels (the same source text hipcc compiles for gfx950) without a GPU.  Nothing in dimo_amd/ loads it.
//
// Execution model: a workgroup is 256 (blockDim.x) FIBERS inside one OS thread, switched cooperatively; wave-level
// operations (__ballot, __shfl_*, readlane) and workgroup barriers are rendezvous points -- a fiber that reaches one
// parks until every live lane of its wave (thread of its workgroup) has arrived.  Wave operations therefore must be
// reached by all live lanes of a wave (as the kernels are written); a wave that never converges is reported as a
// deadlock instead of hanging.  Workgroups of a launch run in any order on a pool of OS threads; global atomics are
// real atomics.  Timing, memory-model races inside a wave and LDS bank behaviour are NOT modelled.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __expf expf  // (the device's fast exponential: a few ulp from expf; tolerances, not bits, for float kernels)
#define __logf logf

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __constant__ static const
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipStreamWaitValueGte = 0 };

// Streams and events (runtime.cpp).  By default an operation runs when it is enqueued (a launch has run when
// hipLaunchKernelGGL returns; a wait must then already hold).  With SIMT_STREAMS=deferred[:seed | :lifo | :fifo] the
// operations QUEUE per stream and run when something synchronises (hipDeviceSynchronize, hipStreamSynchronize,
// hipEventSynchronize, a synchronous copy, simt_synchronize() from a test): one at a time, each time from a stream
// whose head operation is ready -- chosen by a seeded draw, or always the youngest / the oldest ready stream -- so a
// cross-stream dependency that was never enqueued lets a consumer run before its producer.  The null stream is a
// stream like the others (the executor's private streams are non-blocking).
namespace simt {
void submit(hipStream_t stream, std::function<void()> op);
hipStream_t stream_create();
hipEvent_t event_create();
void event_destroy(hipEvent_t e);
void event_record(hipEvent_t e, hipStream_t s);
void stream_wait_event(hipStream_t s, hipEvent_t e);
void stream_write_value(hipStream_t s, uint32_t *p, uint32_t v);
bool stream_wait_value(hipStream_t s, const uint32_t *p, uint32_t v, uint32_t mask);  // false: cannot ever hold (immediate mode)
void synchronize();
}  // namespace simt
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { simt::synchronize(); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { simt::synchronize(); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s) {
  simt::submit(s, [=]() { memset(p, v, n); });
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *src, size_t n, hipMemcpyKind, hipStream_t s) {
  simt::submit(s, [=]() { memcpy(d, src, n); });
  return hipSuccess;
}
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0, *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = simt::stream_create(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = simt::event_create(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { simt::event_destroy(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { simt::event_record(e, s); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { simt::synchronize(); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { simt::stream_wait_event(s, e); return hipSuccess; }
static inline hipError_t hipStreamWriteValue32(hipStream_t s, void *p, uint32_t v, unsigned) {
  simt::stream_write_value(s, static_cast<uint32_t *>(p), v);
  return hipSuccess;
}
static inline hipError_t hipStreamWaitValue32(hipStream_t s, void *p, uint32_t v, unsigned, uint32_t mask) {
  return simt::stream_wait_value(s, static_cast<const uint32_t *>(p), v, mask) ? hipSuccess : hipErrorUnknown;
}
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void *p) { simt::synchronize(); free(p); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { simt::synchronize(); memset(p, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
template <class T>
static inline hipError_t hipMemcpyFromSymbol(void *dst, T *sym, size_t n) { simt::synchronize(); memcpy(dst, sym, n); return hipSuccess; }
template <class T>
static inline hipError_t hipMemcpyToSymbol(T *sym, const void *src, size_t n) { simt::synchronize(); memcpy(sym, src, n); return hipSuccess; }

namespace simt {
struct Idx3 { unsigned x, y, z; };
struct Fiber;
extern thread_local Fiber *cur;
const Idx3 &thread_idx();
const Idx3 &block_idx();
const Idx3 &block_dim();
const Idx3 &grid_dim();
int lane_id();
// deposits v, parks until every live lane of the wave has arrived, returns the 64 deposited values
const uint64_t *wave_exchange(uint64_t v);
uint64_t wave_ballot(bool pred);
void wg_barrier();
int wg_barrier_or(int v);
int wg_barrier_count(int v);
void launch(dim3 grid, dim3 block, size_t dynamic_lds_bytes, const std::function<void()> &body);
void *dynamic_lds();  // the workgroup's dynamically sized LDS (extern __shared__)
}  // namespace simt

#define threadIdx (simt::thread_idx())
#define blockIdx (simt::block_idx())
#define blockDim (simt::block_dim())
#define gridDim (simt::grid_dim())
// (arguments are captured BY VALUE: the launch may run after the caller has returned)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                           \
  do {                                                                                        \
    const dim3 simt_grid_ = (grid), simt_block_ = (block);                                    \
    const size_t simt_lds_ = (size_t)(shmem);                                                 \
    simt::submit((stream), [=]() { simt::launch(simt_grid_, simt_block_, simt_lds_, [&]() { kernel(__VA_ARGS__); }); }); \
  } while (0)
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(simt::dynamic_lds());
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

static inline void __syncthreads() { simt::wg_barrier(); }
static inline int __syncthreads_or(int v) { return simt::wg_barrier_or(v); }
static inline int __syncthreads_count(int pred) { return simt::wg_barrier_count(pred); }
static inline void __threadfence_block() {}
static inline void __threadfence() {}
static inline void __threadfence_system() {}
static inline unsigned long long __ballot(int pred) { return simt::wave_ballot(pred != 0); }
static inline int __shfl(int x, int src, int width = 64) {
  (void)width;
  return (int)(uint32_t)simt::wave_exchange((uint32_t)x)[src & 63];
}
static inline int __shfl_xor(int x, int mask, int width = 64) {
  (void)width;
  return (int)(uint32_t)simt::wave_exchange((uint32_t)x)[(simt::lane_id() ^ mask) & 63];
}
static inline int __shfl_up(int x, int delta, int width = 64) {
  (void)width;
  const int l = simt::lane_id();
  const uint64_t *v = simt::wave_exchange((uint32_t)x);
  return l >= delta ? (int)(uint32_t)v[l - delta] : x;
}
static inline int __shfl_down(int x, int delta, int width = 64) {
  (void)width;
  const int l = simt::lane_id();
  const uint64_t *v = simt::wave_exchange((uint32_t)x);
  return l + delta < 64 ? (int)(uint32_t)v[l + delta] : x;
}
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline double __longlong_as_double(long long x) { double d; memcpy(&d, &x, 8); return d; }
static inline long long __double_as_longlong(double d) { long long x; memcpy(&x, &d, 8); return x; }
static inline double __hiloint2double(int hi, int lo) { return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo)); }
static inline int __double2hiint(double d) { return (int)((unsigned long long)__double_as_longlong(d) >> 32); }
static inline int __double2loint(double d) { return (int)(uint32_t)__double_as_longlong(d); }
static inline float __shfl_xor(float x, int mask, int width = 64) { return __int_as_float(__shfl_xor(__float_as_int(x), mask, width)); }
static inline float __shfl_up(float x, int d, int width = 64) { return __int_as_float(__shfl_up(__float_as_int(x), d, width)); }
static inline float __shfl_down(float x, int d, int width = 64) { return __int_as_float(__shfl_down(__float_as_int(x), d, width)); }
static inline unsigned __shfl_xor(unsigned x, int mask, int width = 64) { return (unsigned)__shfl_xor((int)x, mask, width); }
static inline unsigned __shfl_down(unsigned x, int d, int width = 64) { return (unsigned)__shfl_down((int)x, d, width); }
static inline unsigned __shfl_up(unsigned x, int d, int width = 64) { return (unsigned)__shfl_up((int)x, d, width); }
static inline long long __shfl_xor(long long x, int mask, int width = 64) {
  const uint64_t *v = simt::wave_exchange((uint64_t)x);
  (void)width;
  return (long long)v[(simt::lane_id() ^ mask) & 63];
}
static inline double __shfl_xor(double x, int mask, int width = 64) { return __longlong_as_double(__shfl_xor(__double_as_longlong(x), mask, width)); }
#define __builtin_amdgcn_ballot_w64(p) __ballot((p) ? 1 : 0)
static inline int simt_readlane(int x, int lane) { return (int)(uint32_t)simt::wave_exchange((uint32_t)x)[lane & 63]; }
static inline uint32_t simt_mbcnt_lo(uint32_t mask, uint32_t base) {
  const int l = simt::lane_id();
  return base + (uint32_t)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
static inline uint32_t simt_mbcnt_hi(uint32_t mask, uint32_t base) {
  const int l = simt::lane_id();
  return base + (l > 32 ? (uint32_t)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}
// v[lane J] = sval (sval is wave-uniform: the SGPR operand of v_writelane_b32)
static inline uint32_t simt_writelane(uint32_t v, uint32_t sval, int J) { return simt::lane_id() == J ? sval : v; }
// the fill's hand-written tile step (binning.hip: place_tile): lanes whose mask half has bit jb set take consecutive
// slots from lane j's counter, in lane order, and store their id
static inline void simt_place_tile(uint32_t mhalf, int jb, int j, uint32_t c, uint32_t id, uint32_t *vals) {
  const bool cov = (mhalf >> jb) & 1u;
  const unsigned long long bal = __ballot(cov);
  if (bal == 0) return;
  const uint32_t first = (uint32_t)simt_readlane((int)c, j);
  const int l = simt::lane_id();
  const uint32_t pos = first + (uint32_t)__builtin_popcountll(l ? (bal & (~0ull >> (64 - l))) : 0ull);
  if (cov) vals[pos] = id;
}
// DPP source lane of `lane` under the control word (quad_perm 0x00-0xff, row_shl 0x101-0x10f, row_shr 0x111-0x11f,
// row_ror 0x121-0x12f, the wave shifts / rotates by one, the row mirrors); -1 = out of the row (the lane is then left alone unless bound_ctrl is set)
static inline int simt_dpp_source(int lane, int ctrl) {
  const int row = lane & ~15, pos = lane & 15;
  if (ctrl < 0x100) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  if (ctrl == 0x130) return lane + 1 < 64 ? lane + 1 : -1;  // wave_shl:1
  if (ctrl == 0x134) return (lane + 1) & 63;                 // wave_rol:1
  if (ctrl == 0x138) return lane - 1;                        // wave_shr:1 (-1 for lane 0)
  if (ctrl == 0x13c) return (lane - 1) & 63;                 // wave_ror:1
  if (ctrl == 0x140) return row | (15 - pos);                // row_mirror
  if (ctrl == 0x141) return row | (pos & 8) | (7 - (pos & 7));  // row_half_mirror
  const int n = ctrl & 15, kind = ctrl & ~15;
  if (kind == 0x100) return pos + n < 16 ? row | (pos + n) : -1;
  if (kind == 0x110) return pos - n >= 0 ? row | (pos - n) : -1;
  if (kind == 0x120) return row | ((pos - n) & 15);
  abort();
}
static inline bool simt_dpp_enabled(int lane, int row_mask, int bank_mask) {
  return ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane & 15) >> 2)) & 1);
}
static inline int simt_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int l = simt::lane_id();
  const uint64_t *v = simt::wave_exchange((uint32_t)src);
  const int from = simt_dpp_source(l, ctrl);
  if (!simt_dpp_enabled(l, row_mask, bank_mask)) return old;
  if (from < 0) return bound_ctrl ? 0 : old;
  return (int)(uint32_t)v[from];
}
// v_add_f32_dpp dst, src0, src1 <ctrl> row_mask:0xf bank_mask:<bm>: dst = dpp(src0) + src1 in the enabled lanes whose
// source is inside the row; the others keep dst
static inline void simt_add_f32_dpp(float &dst, float src0, float src1, int ctrl, int bank_mask) {
  const int l = simt::lane_id();
  const uint64_t *v = simt::wave_exchange(__float_as_uint(src0));
  const int from = simt_dpp_source(l, ctrl);
  if (simt_dpp_enabled(l, 0xf, bank_mask) && from >= 0) dst = __uint_as_float((uint32_t)v[from]) + src1;
}
// v_permlane32_swap_b32 a, b: a's lanes 32-63 trade places with b's lanes 0-31; v_permlane16_swap_b32 a, b: a's odd
// rows of 16 trade places with b's even rows
static inline void simt_permlane_swap(float &a, float &b, int half) {
  const int l = simt::lane_id();
  const uint64_t *v = simt::wave_exchange(((uint64_t)__float_as_uint(a) << 32) | __float_as_uint(b));
  if (l & half) a = __uint_as_float((uint32_t)v[l - half]);        // b of the partner
  else b = __uint_as_float((uint32_t)(v[l + half] >> 32));          // a of the partner
}
// wave_ops.hpp's wave_reduce16, instruction for instruction (its asm block spelled as calls; the s_nop wait states
// have no meaning here).  NV as there: the first halving step leaves out the values that are not in use.
template <int NV>
static inline void simt_wave_reduce16(float (&v)[16]) {
  constexpr int ROR8 = 0x128, SHL4 = 0x104, SHR4 = 0x114;
  for (int s = 0; s < 8; ++s) {
    simt_add_f32_dpp(v[s], v[s], v[s], ROR8, 0x3);
    if (s + 8 < NV) simt_add_f32_dpp(v[s], v[s + 8], v[s + 8], ROR8, 0xc);
  }
  for (int t = 0; t < 4; ++t) {
    simt_add_f32_dpp(v[t], v[t], v[t], SHL4, 0x5);
    simt_add_f32_dpp(v[t], v[t + 4], v[t + 4], SHR4, 0xa);
  }
  simt_permlane_swap(v[0], v[2], 32);
  simt_permlane_swap(v[1], v[3], 32);
  v[0] = v[0] + v[2];
  v[1] = v[1] + v[3];
  simt_permlane_swap(v[0], v[1], 16);
  v[0] = v[0] + v[1];
  simt_add_f32_dpp(v[1], v[0], v[0], 0xB1, 0xf);
  simt_add_f32_dpp(v[0], v[1], v[1], 0x4E, 0xf);
}
// v_readfirstlane_b32: the value of the lowest live lane that takes part (the kernels use it on wave-uniform values)
static inline int simt_readfirstlane(int x) {
  const unsigned long long live = __ballot(1);
  return (int)(uint32_t)simt::wave_exchange((uint32_t)x)[__builtin_ctzll(live)];
}
#define __builtin_amdgcn_readfirstlane simt_readfirstlane
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_exp2f exp2f  // (v_exp_f32 / v_rcp_f32 are 1 ulp: tolerances, not bits)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// The fp32 MFMAs of the TimeNet kernels as wave rendezvous (CDNA3/4 ISA, matrix-core operand layouts; cbsz / abid /
// blgp = 0 as the kernels pass them).  Every lane deposits its (a, b), then computes the result registers it owns.
typedef float simt_f32x4 __attribute__((vector_size(16)));
typedef float simt_f32x16 __attribute__((vector_size(64)));
static inline const uint64_t *simt_mfma_operands(float a, float b) {
  return simt::wave_exchange(((uint64_t)__float_as_uint(a) << 32) | __float_as_uint(b));
}
#define SIMT_A(l) __uint_as_float((uint32_t)(ab[l] >> 32))
#define SIMT_B(l) __uint_as_float((uint32_t)ab[l])
// v_mfma_f32_16x16x4_f32: lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]; register r of lane l is
// D[i = 4 (l / 16) + r][j = l % 16]
static inline simt_f32x4 simt_mfma_16x16x4(float a, float b, simt_f32x4 c, int, int, int) {
  const int l = simt::lane_id(), j = l & 15;
  const uint64_t *ab = simt_mfma_operands(a, b);
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float s = c[r];
    for (int k = 0; k < 4; ++k) s = fmaf(SIMT_A(16 * k + i), SIMT_B(16 * k + j), s);
    c[r] = s;
  }
  return c;
}
// v_mfma_f32_32x32x2_f32: lane l holds A[i = l % 32][k = l / 32], B[k = l / 32][j = l % 32]; register r of lane l is
// D[i = 8 (r / 4) + 4 (l / 32) + r % 4][j = l % 32]
static inline simt_f32x16 simt_mfma_32x32x2(float a, float b, simt_f32x16 c, int, int, int) {
  const int l = simt::lane_id(), j = l & 31;
  const uint64_t *ab = simt_mfma_operands(a, b);
  for (int r = 0; r < 16; ++r) {
    const int i = 8 * (r >> 2) + 4 * (l >> 5) + (r & 3);
    float s = c[r];
    for (int k = 0; k < 2; ++k) s = fmaf(SIMT_A(32 * k + i), SIMT_B(32 * k + j), s);
    c[r] = s;
  }
  return c;
}
// v_mfma_f32_4x4x1_16b_f32: sixteen 4x4x1 blocks, block = l / 4: lane l holds A_block[i = l % 4], B_block[j = l % 4];
// register r of lane l is D_block[i = r][j = l % 4]
static inline simt_f32x4 simt_mfma_4x4x1(float a, float b, simt_f32x4 c, int, int, int) {
  const int l = simt::lane_id();
  const uint64_t *ab = simt_mfma_operands(a, b);
  for (int r = 0; r < 4; ++r) c[r] = fmaf(SIMT_A((l & ~3) + r), SIMT_B(l), c[r]);
  return c;
}
#undef SIMT_A
#undef SIMT_B
#define __builtin_amdgcn_mfma_f32_16x16x4f32 simt_mfma_16x16x4
#define __builtin_amdgcn_mfma_f32_32x32x2f32 simt_mfma_32x32x2
#define __builtin_amdgcn_mfma_f32_4x4x1f32 simt_mfma_4x4x1
#define __builtin_amdgcn_update_dpp simt_update_dpp
#define __builtin_amdgcn_readlane simt_readlane
#define __builtin_amdgcn_mbcnt_lo simt_mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi simt_mbcnt_hi
#define __builtin_amdgcn_s_memrealtime() 0ull

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

template <class T>
static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float *p, float v) {  // (the order of the adds is the schedule's, as on the device)
  uint32_t *u = reinterpret_cast<uint32_t *>(p), o = __atomic_load_n(u, __ATOMIC_RELAXED);
  while (!__atomic_compare_exchange_n(u, &o, __float_as_uint(__uint_as_float(o) + v), true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return __uint_as_float(o);
}
static inline float unsafeAtomicAdd(float *p, float v) { return atomicAdd(p, v); }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <class T>
static inline void __hip_atomic_store(T *p, T v, int order, int) { __atomic_store_n(p, v, order); }
template <class T>
static inline T __hip_atomic_load(const T *p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <class T>
static inline T atomicMax(T *p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T>
static inline T atomicMin(T *p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T>
static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return a < b ? a : b; }
static inline float max(float a, float b) { return a > b ? a : b; }
