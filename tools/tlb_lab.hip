// Round 6 lab: what does a kernel's FIRST touch of memory it has not seen for a while cost beyond the cache miss - i.e. do address translations matter for
// the decode chain, whose 229 launches per step each stream a different 3-13 MB weight matrix out of a 1.6 GB set?
//   tools/bin/tlb_lab [launches]
// A chain of dependent launches (same stream), each a skinny-GEMM-like read of REG bytes (240 workgroups x 4 waves x 10 one-KiB wave requests, all requested up
// front, summed, one store per workgroup) from region (i * STRIDE) mod SPAN of one big allocation:
//   SPAN = REG           the same bytes every launch (caches and translations warm)
//   SPAN = 512 MB        beyond L2 (32 MB) and the Infinity Cache (256 MB): data cold, 256 two-MB pages
//   SPAN = 2 / 8 / 24 GB data cold, 1 000 - 12 000 two-MB pages
// and the same with a RIDER: one wave of every launch also touches one cache line in each 2 MB page of the NEXT launch's region (translations warmed one launch ahead,
// 5 loads).  Output: us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr size_t REG = 240ull * 4 * 10 * 1024;      // 9.83 MB per launch (the QKV matrix's bytes)

__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ base, const char* __restrict__ next, int rider, unsigned* out) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const u32x4* p = base + ((size_t)(blockIdx.x * 4 + wave) * 10) * 64 + lane;
  u32x4 v[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
  unsigned r = 0;
  if (rider && blockIdx.x == 0 && wave == 0 && lane < 6) r = *reinterpret_cast<const unsigned*>(next + (size_t)lane * (2u << 20));      // one line per 2 MB page of the next region
  unsigned s = r;
#pragma unroll
  for (int u = 0; u < 10; ++u) s += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  if (s == 0x12345678u) out[blockIdx.x] = s;      // (never true on zero-filled memory: keeps the loads alive without a store on the path)
  if (tid == 0 && blockIdx.x == 0) out[0] = 1;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3000;
  size_t freeb = 0, total = 0; CK(hipMemGetInfo(&freeb, &total));
  const size_t cap = 26ull << 30;
  char* buf; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 0, cap));
  unsigned* out; CK(hipMalloc(&out, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t STRIDE = 12ull << 20;      // regions 12 MB apart (2 MB aligned)
  const size_t spans[] = {STRIDE, 512ull << 20, 2ull << 30, 8ull << 30, 24ull << 30};
  printf("chain of %d dependent launches, each reading %.2f MB (240 workgroups x 40 one-KiB wave requests, non-temporal), region stride 12 MB\n", n, REG / 1e6);
  for (int round = 0; round < 3; ++round)
   for (size_t span : spans)
    for (int rider = 0; rider < 2; ++rider) {
      const size_t nreg = span / STRIDE;
      for (int rep = 0; rep < 2; ++rep) {      // (first repetition warms what can be warmed)
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) {
          const size_t off = (size_t)(i % nreg) * STRIDE, nxt = (size_t)((i + 1) % nreg) * STRIDE;
          hipLaunchKernelGGL(stream_read, dim3(240), dim3(256), 0, st, reinterpret_cast<const u32x4*>(buf + off), buf + nxt, rider, out);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("  span %8.0f MB (%5zu regions) %s: %.3f us per launch\n", span / 1048576.0, nreg, rider ? "with the next region's pages touched one launch ahead" : "plain", 1e3 * ms / n);
      }
    }
  return 0;
}
