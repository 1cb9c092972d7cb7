// Persistent decode-layer SKELETON (round-3 review item 4: "build the persistent engine in tools/ and MEASURE it").
//
// Question it answers with a measurement on this chip, at Whisper large-v2's decode shapes (5 rows = beam 5 of one utterance,
// d = 1280): if a decoder layer ran as ONE persistent launch - every CU keeps its slice of the NEXT stage's weights streaming
// into LDS (LDS-DMA, non-temporal, loader wave) while it waits at the all-to-all edge of the current stage, edges are 8-byte
// {tag, value} granule all-gathers (MI355X_MICROARCH.md price list: allgather / prefetch-credit / engine-vs-launches) - what does
// one stage cost, against the same stage as its own kernel launch in a graph-replayed chain?
//
// The skeleton is a real, checkable computation with the traffic of the real layer but uniform stages:
//     x_{s+1}[r][f] = tanh-free "activation" of sum_k W_s[f][k] x_s[r][k],   r < 5 rows, f < 1280 features, k < 1280,
// i.e. a chain of 1280 x 1280 GEMVs on 5 rows.  256 workgroups (one per CU), workgroup c owns features 5c .. 5c+4 of every stage
// and streams a 16-feature MFMA tile of W_s (40 k-steps x 1 KiB fragments = 40 KiB per stage and CU: 10.5 MB per stage over the
// chip - a Whisper layer has 46 MB over 7 stages = 6.6 MB per stage, so the skeleton streams MORE than the real layer; `ksteps`
// scales it).  Every stage ends in the all-to-all edge the real layer has (each output needs the whole input row): 15 granules per
// CU (two f16 per granule), 3840 per stage = 30 KiB swept by every CU.
//   engine  : 1 loader wave + 3 consumer waves per CU; stage s+1's weights are requested at the top of stage s; consumers sweep the
//             granules of stage s into LDS, MFMA over the LDS-resident weights, reduce, publish; tags = stage index + 1 (monotonic:
//             nothing is reset), two granule buffers (a CU can be at most one stage ahead of the slowest reader).
//   chain   : the same stage as a kernel of 256 workgroups (weights prefetched into registers, rows read from the previous kernel's
//             output), S launches captured into a HIP graph - what the product's decode step is.
// Both are checked against a plain reference of the chain.  Every spin is bounded (error flag, no hang).
//
// MEASURED on MI355X (round 4, gpurun_out/r4b, r4c; 56 stages = eight 7-stage layers, weights beyond the Infinity Cache):
//     k-steps per CU and stage (MB per stage) | launch chain, us per stage | engine, us per stage | engine / chain
//       40 (10.5 MB)                          |  5.16                      |  6.01                |  1.16
//       20 ( 5.2 MB)                          |  4.28                      |  5.50                |  1.29
//        8 ( 2.1 MB)                          |  3.79                      |  5.38                |  1.42
//   The engine hides most of the weight stream (40 vs 8 k-steps: +0.6 us against the chain's +1.4 us) but its all-to-all edge - last
//   publish -> write-through store visible -> one 30 KiB sweep per CU -> barrier - costs 5.4 us where the chain pays 3.8 us for a
//   kernel boundary plus its first dependent loads.  A Whisper large-v2 decode stage streams 3.3-13.1 MB (0.5-2 us of HBM time): too
//   little for the prefetch credit to buy back the edge.  A second form (rows double-buffered, no closing barrier, consumers start
//   sweeping right after the reduction barrier) was SLOWER (7.8 / 7.2 / 7.1 us per stage): sweeps that start before the producers
//   have published only add failed 30 KiB passes in front of the one that succeeds (the guide's polling-cost row).
//   -> the product's decode step stays a graph-replayed launch chain.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/engine_lab tools/engine_lab.hip
// run:   tools/bin/engine_lab [stages=56] [ksteps=40] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int NCU = 256, D = 1280, ROWS = 5, FPC = D / NCU;          // 5 features per CU
constexpr int GPC = FPC * 3;                                        // granules per CU and stage: (feature, row pair) -> two f16
constexpr int NGRAN = NCU * GPC;                                    // 3840 per stage
constexpr int XSTR = D + 8;                                         // f16 row pitch of the activation rows in LDS
constexpr unsigned SPIN_LIMIT = 1u << 18;
// raw workgroup barrier: __syncthreads() would first drain the loader wave's LDS-DMA queue (the prefetch this structure lives on)
#define BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

typedef __attribute__((address_space(1))) const void* gbl_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned hash_u(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// weights: per stage and CU one 16-feature tile in MFMA A-fragment order [kstep][64 lanes][8 f16]; only tile rows 0..4 are features
// the CU owns, rows 5..15 are filler (streamed like the real tile, results unused)
__global__ void fill_w(f16* w, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    w[i] = (f16)(((float)(hash_u((unsigned)i * 2654435761u + 17u) & 0xffff) / 32768.0f - 1.0f) * 0.05f);
}
__device__ __forceinline__ float act(float v) { return v * 0.9f + 0.01f; }       // keeps the chain's values O(1)

// ---- plain reference / launch-chain stage: y[r][5c + f] = act(sum_k W[c][k][f] x[r][k]) ------------------------------------
// the chain kernel IS the product's structure in miniature: weight fragments prefetched into registers (non-temporal), rows from
// the previous kernel's output, 4 waves split K, LDS reduction
__global__ __launch_bounds__(256) void chain_stage(const f16* __restrict__ W, const f16* __restrict__ xin, f16* __restrict__ xout, int ksteps) {
  __shared__ __attribute__((aligned(16))) f16 xs[ROWS * XSTR];
  __shared__ float red[4][64][4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32x4* wp = reinterpret_cast<const u32x4*>(W) + (size_t)c * ksteps * 64 + lane;
  const int per = ksteps / 4;          // k-steps per wave (ksteps % 4 == 0)
  u32x4 wf[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) if (u < per) wf[u] = __builtin_nontemporal_load(wp + (size_t)(wave * per + u) * 64);
  for (int i = tid; i < ROWS * (D / 8); i += 256) { const int r = i / (D / 8), k8 = i % (D / 8); *reinterpret_cast<u32x4*>(xs + r * XSTR + 8 * k8) = *reinterpret_cast<const u32x4*>(xin + (size_t)r * D + 8 * k8); }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int xrow = ((lane & 15) < ROWS ? (lane & 15) : ROWS - 1) * XSTR + 8 * (lane >> 4);
#pragma unroll
  for (int u = 0; u < 10; ++u) if (u < per) {
    const f16x8 xb = *reinterpret_cast<const f16x8*>(xs + xrow + (wave * per + u) * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&wf[u]), xb, acc, 0, 0, 0);
  }
  *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc;
  __syncthreads();
  if (tid < 64) {      // D[i = feature][j = row]: lane holds row j = lane & 15, features 4 (lane >> 4) + reg
    const int r = lane & 15, f0 = 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = (red[0][lane][q] + red[1][lane][q]) + (red[2][lane][q] + red[3][lane][q]);
      if (r < ROWS && f0 + q < FPC) xout[(size_t)r * D + FPC * c + f0 + q] = (f16)act(v);
    }
  }
}

// ---- the persistent engine ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_gran(u64* p, unsigned tag, unsigned v) { __hip_atomic_store(p, ((u64)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_gran(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int KSTEPS>
__global__ __launch_bounds__(256) void engine(const f16* __restrict__ W, const f16* __restrict__ x0, f16* __restrict__ xout, u64* gran /*[2][NGRAN]*/,
                                              unsigned* err, int stages, size_t wstage /*f16 elements per stage*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f16* wbuf = reinterpret_cast<f16*>(smem);                                   // [2][KSTEPS][64][8]
  f16* xs = wbuf + 2 * KSTEPS * 512;                                          // [ROWS][XSTR]
  float* red = reinterpret_cast<float*>(xs + ROWS * XSTR + 8);                // [3][64][4]
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // stage 0: rows from global (the layer input), weights of stage 0 requested by the loader wave
  if (wave == 0) {
#pragma unroll 8
    for (int u = 0; u < KSTEPS; ++u)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(W + (size_t)c * KSTEPS * 512 + (size_t)u * 512 + lane * 8), (lds_ptr_t)(wbuf + u * 512), 16, 0, 2);
  } else {
    for (int i = tid - 64; i < ROWS * (D / 8); i += 192) { const int r = i / (D / 8), k8 = i % (D / 8); *reinterpret_cast<u32x4*>(xs + r * XSTR + 8 * k8) = *reinterpret_cast<const u32x4*>(x0 + (size_t)r * D + 8 * k8); }
  }
  const int xrow = ((lane & 15) < ROWS ? (lane & 15) : ROWS - 1) * XSTR + 8 * (lane >> 4);
  __shared__ int s_abort;
  if (tid == 0) s_abort = 0;
  for (int s = 0; s < stages; ++s) {
    const unsigned tag_in = (unsigned)s, tag_out = (unsigned)s + 1u;          // stage s consumes the granules tagged s (s >= 1), publishes tag s + 1
    if (wave == 0) {
      // loader: request stage s + 1 (its buffer was last read in stage s - 1: behind that stage's closing barrier), then wait for stage s
      if (s + 1 < stages) {
        const f16* wsrc = W + (size_t)(s + 1) * wstage + (size_t)c * KSTEPS * 512 + lane * 8;
        f16* wdst = wbuf + ((s + 1) & 1) * KSTEPS * 512;
#pragma unroll 8
        for (int u = 0; u < KSTEPS; ++u) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + (size_t)u * 512), (lds_ptr_t)(wdst + u * 512), 16, 0, 2);
        // vmcnt(KSTEPS): everything but the KSTEPS requests just issued has landed = stage s is in LDS
        constexpr int VM = KSTEPS;
        __builtin_amdgcn_s_waitcnt(((VM >> 4) << 14) | 0x0F70 | (VM & 15));
      } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
    } else if (s > 0) {
      // consumers: sweep the 3840 granules of the previous stage into the LDS rows (192 lanes x 20 granules)
      const u64* g = gran + (size_t)(s & 1) * NGRAN;
      unsigned spins = 0;
      bool done[20];
#pragma unroll
      for (int i = 0; i < 20; ++i) done[i] = false;
      for (;;) {
        bool ok = true;
        u64 v[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) { const int gi = (tid - 64) + 192 * i; v[i] = ld_gran(g + gi); }
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          const int gi = (tid - 64) + 192 * i;
          const bool hit = (unsigned)(v[i] >> 32) == tag_in;
          if (hit && !done[i]) {
            const int cc = gi / GPC, w = gi % GPC, f = w / 3, p = w % 3, col = FPC * cc + f;
            const unsigned val = (unsigned)v[i];
            xs[(2 * p) * XSTR + col] = *reinterpret_cast<const f16*>(&val);
            if (2 * p + 1 < ROWS) { const unsigned short hi = (unsigned short)(val >> 16); xs[(2 * p + 1) * XSTR + col] = *reinterpret_cast<const f16*>(&hi); }
            done[i] = true;
          }
          ok = ok & (hit | done[i]);
        }
        if (__all(ok)) break;
        if (++spins > SPIN_LIMIT) { if (lane == 0) { atomicOr(err, 1u); s_abort = 1; } break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    BAR();                                                                    // A: rows + stage-s weights are in LDS
    if (s_abort) break;                                                       // (uniform: read behind the barrier)
    if (wave > 0) {
      const f16* wb = wbuf + (s & 1) * KSTEPS * 512 + lane * 8;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int u = wave - 1; u < KSTEPS; u += 3) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(wb + u * 512);
        const f16x8 xb = *reinterpret_cast<const f16x8*>(xs + xrow + u * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb, acc, 0, 0, 0);
      }
      *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = acc;
    }
    BAR();                                                                    // R: partials written
    if (wave == 1) {
      // lane = (row j = lane & 15, features 4 (lane >> 4) + q): features 0..4 of the tile are real -> lanes 0..15 (q = 0..3) and 16..31 (q = 0)
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = act((red[(0 * 64 + lane) * 4 + q] + red[(1 * 64 + lane) * 4 + q]) + red[(2 * 64 + lane) * 4 + q]);
      const int r = lane & 15, fb = 4 * (lane >> 4);
      if (s + 1 == stages) {
        if (r < ROWS) for (int q = 0; q < 4; ++q) if (fb + q < FPC) xout[(size_t)r * D + FPC * c + fb + q] = (f16)v[q];
      } else {
        // granule (feature f, pair p) = rows 2p, 2p + 1: the partner row's value comes from lane + 1 (same feature group)
        u64* g = gran + (size_t)((s + 1) & 1) * NGRAN + (size_t)c * GPC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f16 me = (f16)v[q];
          const unsigned short mine = *reinterpret_cast<const unsigned short*>(&me);
          const unsigned other = (unsigned)__shfl_down((int)mine, 1);
          const int f = fb + q;
          if (f < FPC && r < ROWS && (r & 1) == 0) st_gran(g + f * 3 + (r >> 1), tag_out, (unsigned)mine | (r + 1 < ROWS ? (other << 16) : 0u));
        }
      }
    }
    BAR();                                                                    // B: stage closed (rows buffer and weight buffer s & 1 are free)
  }
}

int main(int argc, char** argv) {
  const int stages = argc > 1 ? atoi(argv[1]) : 56;
  const int ksteps = argc > 2 ? atoi(argv[2]) : 40;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  if (ksteps != 40 && ksteps != 20 && ksteps != 8) { printf("ksteps must be 40, 20 or 8\n"); return 1; }
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t wstage = (size_t)NCU * ksteps * 512;                  // f16 per stage
  f16 *W, *x0, *xa, *xb, *xe; u64* gran; unsigned* err;
  CK(hipMalloc(&W, wstage * stages * 2)); CK(hipMalloc(&x0, (size_t)ROWS * D * 2)); CK(hipMalloc(&xa, (size_t)ROWS * D * 2)); CK(hipMalloc(&xb, (size_t)ROWS * D * 2)); CK(hipMalloc(&xe, (size_t)ROWS * D * 2));
  CK(hipMalloc(&gran, (size_t)2 * NGRAN * 8)); CK(hipMalloc(&err, 4));
  fill_w<<<2048, 256, 0, st>>>(W, wstage * stages);
  std::vector<f16> hx((size_t)ROWS * D);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (f16)(0.5f * sinf(0.37f * (float)i) + 0.1f);
  CK(hipMemcpy(x0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(gran, 0, (size_t)2 * NGRAN * 8)); CK(hipMemset(err, 0, 4));
  printf("skeleton: %d stages x %d CUs, %d k-steps per CU and stage = %.1f MB of weights per stage (%.0f MB total: %s the 256 MiB Infinity Cache), edge = %d granules (%.1f KiB)\n",
         stages, NCU, ksteps, wstage * 2 / 1e6, wstage * 2.0 * stages / 1e6, wstage * 2.0 * stages > 268e6 ? "beyond" : "INSIDE", NGRAN, NGRAN * 8 / 1024.0);
  // ---- launch chain (graph)
  hipGraph_t graph; hipGraphExec_t gexec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < stages; ++s) {
    const f16* in = s == 0 ? x0 : ((s & 1) ? xa : xb); f16* out = (s & 1) ? xb : xa;
    hipLaunchKernelGGL(chain_stage, dim3(NCU), dim3(256), 0, st, W + (size_t)s * wstage, in, out, ksteps);
  }
  CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(gexec, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(gexec, st));
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms_chain; CK(hipEventElapsedTime(&ms_chain, e0, e1));
  f16* chain_out = ((stages - 1) & 1) ? xb : xa;
  std::vector<f16> ref((size_t)ROWS * D), got((size_t)ROWS * D);
  CK(hipMemcpy(ref.data(), chain_out, ref.size() * 2, hipMemcpyDeviceToHost));
  printf("launch chain (graph replay): %.3f us per stage (%.1f us per 7-stage layer)\n", 1e3 * ms_chain / reps / stages, 7e3 * ms_chain / reps / stages);
  // ---- engine
  const size_t lds = (size_t)2 * ksteps * 1024 + (size_t)ROWS * XSTR * 2 + 16 + 3 * 64 * 16;
  auto run_engine = [&]() {
    if (ksteps == 40) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine<40>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(engine<40>, dim3(NCU), dim3(256), lds, st, W, x0, xe, gran, err, stages, wstage); }
    else if (ksteps == 20) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine<20>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(engine<20>, dim3(NCU), dim3(256), lds, st, W, x0, xe, gran, err, stages, wstage); }
    else { hipLaunchKernelGGL(engine<8>, dim3(NCU), dim3(256), lds, st, W, x0, xe, gran, err, stages, wstage); }
  };
  // tags are stage indices: the granule buffers must not hold a previous run's tags -> cleared before every launch (a memset node in a product)
  CK(hipMemsetAsync(gran, 0, (size_t)2 * NGRAN * 8, st));
  run_engine(); CK(hipStreamSynchronize(st));
  unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(got.data(), xe, got.size() * 2, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; size_t nbit = 0;
  for (size_t i = 0; i < ref.size(); ++i) { maxd = fmax(maxd, fabs((double)got[i] - (double)ref[i])); maxv = fmax(maxv, fabs((double)ref[i])); nbit += memcmp(&got[i], &ref[i], 2) != 0; }
  printf("engine vs chain after %d stages: max |diff| %.3e (max |value| %.3f), %zu of %zu f16 outputs differ bitwise, spin error flag %u\n", stages, maxd, maxv, nbit, ref.size(), herr);
  float ms_eng = 0;
  if (!herr) {
    float best = 1e30f, sum = 0;
    for (int i = 0; i < reps; ++i) {
      CK(hipMemsetAsync(gran, 0, (size_t)2 * NGRAN * 8, st));
      CK(hipEventRecord(e0, st)); run_engine(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); sum += t; best = fminf(best, t);
    }
    ms_eng = sum / reps;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("persistent engine: %.3f us per stage mean, %.3f best (%.1f us per 7-stage layer); error flag %u\n", 1e3 * ms_eng / stages, 1e3 * best / stages, 7e3 * ms_eng / stages, herr);
    printf("engine / chain = %.3f\n", ms_eng / (ms_chain / reps));
  }
  return herr ? 2 : 0;
}
