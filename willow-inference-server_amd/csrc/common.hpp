// common.hpp — shared host/device helpers for libwis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// exact (erf) GELU, as torch.nn.functional.gelu default / CT2 GELU
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- per-device context (stream + log-mel tables) -------------------------------------
struct DeviceCtx;
int get_ctx(int device, DeviceCtx** out);
hipStream_t ctx_stream(DeviceCtx* c);

// log-mel on `stream`: pcm (device) -> mel f32 [n_win][80][3000] (device, may be null) and/or
// conv1 input f16 [n_win][3002][96] (device, may be null; rows 0 and 3001 and cols 80..95 zero)
int logmel_device(DeviceCtx* c, hipStream_t stream, const float* d_pcm, int64_t stride,
                  const int64_t* d_nsamp, int n_win, float* d_mel, f16* d_conv_in);

}  // namespace wis
