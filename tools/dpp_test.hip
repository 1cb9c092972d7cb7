#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.hpp"
__global__ void k(const float* in, float* out) { float v = in[threadIdx.x]; out[threadIdx.x] = wis::wave_sum(v); out[64 + threadIdx.x] = wis::wave_max(v); }
int main() { float h[64], o[128]; float s = 0, m = -1e30; for (int i = 0; i < 64; ++i) { h[i] = (float)((i * 37) % 23) - 7.5f + 0.01f * i; s += h[i]; m = fmaxf(m, h[i]); }
  float *d, *e; hipMalloc(&d, 256); hipMalloc(&e, 512); hipMemcpy(d, h, 256, hipMemcpyHostToDevice); k<<<1, 64>>>(d, e); hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) { if (fabsf(o[i] - s) > 1e-3f || o[64 + i] != m) ++bad; }
  printf("wave_sum %f (expect %f) wave_max %f (expect %f) bad lanes %d\n", o[0], s, o[64], m, bad); return bad != 0; }
