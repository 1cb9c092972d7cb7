// common.hpp — shared host/device helpers for libwis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>

#include "wis_hip.h"

namespace wis {

// thread-local last-error string (wis_last_error)
void set_error(const char* fmt, ...);
const char* get_error();

#define WIS_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::wis::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return WIS_E_HIP;                                                                  \
    }                                                                                    \
  } while (0)

#define WIS_RET(expr)                    \
  do {                                   \
    int _rc = (expr);                    \
    if (_rc != WIS_OK) return _rc;       \
  } while (0)

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device-side helpers ---------------------------------------------------------------
// 64-lane reductions: four DPP steps inside each 16-lane row (quad_perm xor1, xor2, row_half_mirror, row_mirror)
// then the four row totals are combined through v_readlane -> a wave-uniform result in ~12 instructions,
// instead of six dependent ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  const float a = readlane_f(v, 0), b = readlane_f(v, 16), c = readlane_f(v, 32), d = readlane_f(v, 48);
  return (a + b) + (c + d);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  const float a = readlane_f(v, 0), b = readlane_f(v, 16), c = readlane_f(v, 32), d = readlane_f(v, 48);
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16), c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
// Selection keys: a (value, index) pair ordered "value descending, then index ascending" becomes ONE unsigned 64-bit number
// ordered ascending = better - the order-preserving image of the float in the high word, the complemented index in the low
// word.  A selection step is then a 64-bit compare and two selects, with no mask logic on the scalar unit (the (v > w) ||
// (v == w && i < j) form costs three compares, two SALU mask ops and the VALU <-> SALU hazards between them per element).
// Key 0 = "no entry" (below every real key, whose high word is at least 0x007fffff = the image of -inf).
typedef unsigned long long u64;
__device__ __forceinline__ unsigned ord_f(float x) { const unsigned u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ u64 sel_key(float v, int idx) { return ((u64)ord_f(v) << 32) | (unsigned)(~idx); }
__device__ __forceinline__ float key_value(u64 k) { return ord_inv((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_index(u64 k) { return (int)~(unsigned)k; }
__device__ __forceinline__ u64 key_max(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 wave_max_key(u64 k) {
  const unsigned hi = wave_max_u((unsigned)(k >> 32));
  const unsigned lo = wave_max_u((unsigned)(k >> 32) == hi ? (unsigned)k : 0u);
  return ((u64)hi << 32) | lo;
}
// wave-wide best of (value descending, index ascending): returns the winning pair on every lane
__device__ __forceinline__ void wave_argbest(float& v, int& idx) {
  const float mx = wave_max(v);
  const int mi = wave_min_i(v == mx ? idx : 0x7fffffff);
  v = mx; idx = mi;
}
// Tuning taps (wis_debug_phase_cycles / wis_debug_timeline): compiled in only with -DWIS_TAPS=1
// (WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1 python willow-inference-server_amd/build.py); the product build carries no tap code.
#ifndef WIS_TAPS
#define WIS_TAPS 0
#endif
// shader-clock stamp of workgroup 0 for the phase profiler
// A float at a workgroup-uniform address through the scalar data cache, REQUESTED here and waited for by the caller
// (uniform_load_wait() before the first use).  Written as inline asm on purpose: hipcc sinks an ordinary scalar load to its first use
// and waits for it there - five dependent scalar round trips in the middle of gemv_body instead of five requests that travel with the
// kernel's first vector loads - and it does not count asm-issued loads in its own s_waitcnt bookkeeping.  The scalar cache is
// invalidated at kernel start like the vector L1, so rows written by the previous launch are seen.
__device__ __forceinline__ float uniform_load_issue_f32(const float* p) {
  // (the builtin returns int: both halves go through `unsigned` before they are widened, or a low word with bit 31 set sign-extends into the high one)
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<unsigned long long>(p)));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<unsigned long long>(p) >> 32));
  const unsigned long long a = (unsigned long long)lo | ((unsigned long long)hi << 32);
  float v;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(a) : "memory");
  return v;
}
__device__ __forceinline__ unsigned uniform_load_issue_u32(const void* p) {      // (the dword at a 4-byte aligned uniform address)
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<unsigned long long>(p)));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<unsigned long long>(p) >> 32));
  const unsigned long long a = (unsigned long long)lo | ((unsigned long long)hi << 32);
  unsigned v;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(a) : "memory");
  return v;
}
__device__ __forceinline__ void uniform_load_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// ... and every value requested that way passes through uniform_load_landed() behind the wait before it is used: hipcc sees no data
// dependence between the wait and a register it believes was defined by the (long finished) asm statement, and did hoist the f16 -> f32
// conversion of a shift above the wait - reading the SGPR before the load had written it.  Volatile asm statements keep their order
// among themselves, and the "+s" operand makes every later use depend on this one.
__device__ __forceinline__ void uniform_load_landed(float& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void uniform_load_landed(unsigned& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void stamp(unsigned long long* prof, int i) {
#if WIS_TAPS
  if (prof) prof[i] = __builtin_amdgcn_s_memtime();
#endif
}
// whole-kernel span for the step timeline: thread 0 of EVERY workgroup folds the 100 MHz constant clock into slot 14
// (min = first workgroup start) / slot 15 (max = last workgroup end) of the kernel's stamp row
__device__ __forceinline__ void tl_begin(unsigned long long* prof) {
#if WIS_TAPS
  if (prof) atomicMin(prof + 14, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}
__device__ __forceinline__ void tl_end(unsigned long long* prof) {
#if WIS_TAPS
  if (prof) atomicMax(prof + 15, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}
// exact (erf) GELU, as torch.nn.functional.gelu default / CT2 GELU
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far inside the f16 rounding of every consumer): one v_exp, one v_rcp and
// a degree-5 polynomial instead of the ~40-instruction libm erff
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// ---- per-device context (stream + log-mel tables) -------------------------------------
struct DeviceCtx;
int get_ctx(int device, DeviceCtx** out);
hipStream_t ctx_stream(DeviceCtx* c);
std::mutex& ctx_op_mutex(DeviceCtx* c);   // serialises the single-kernel taps (wis_op_*) that share ctx_stream

// log-mel on `stream`: pcm (device) -> mel f32 [n_win][80][3000] (device, may be null) and/or
// conv1 input f16 [n_win][3002][96] (device, may be null; rows 0 and 3001 and cols 80..95 zero).
// d_logspec f32 [n_win][80][3000] and d_gmax u32 [n_win] are the caller's own scratch (never shared between in-flight calls).
int logmel_device(DeviceCtx* c, hipStream_t stream, float* d_logspec, unsigned* d_gmax, const float* d_pcm, int64_t stride,
                  const int64_t* d_nsamp, int n_win, float* d_mel, f16* d_conv_in);
// the two halves, for streaming sessions: log-spectrum of the 16-frame tiles [tile0, tile0 + n_tiles) of every window
// (accumulates the running maximum into d_gmax, which the caller zeroes once per window), then clamp / scale / emit
int logmel_frames(DeviceCtx* c, hipStream_t stream, float* d_logspec, unsigned* d_gmax, const float* d_pcm, int64_t stride,
                  const int64_t* d_nsamp, int n_win, int tile0, int n_tiles);
int logmel_finalize(hipStream_t stream, const float* d_logspec, const unsigned* d_gmax, int n_win, float* d_mel, f16* d_conv_in);

}  // namespace wis
