// gemm_lab.hip — standalone tuning bench for the encoder GEMM main loop (gfx950).  NOT part of the product library: it carries loop
// variants of willow-inference-server_amd/csrc/enc_kernels.hip's gemm_f16_kernel with a plain f16 epilogue, times them on the four
// per-layer encoder shapes, checks every variant against variant 0, and (optionally) stamps the phases of one wave's k-loop with
// the shader clock.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/gemm_lab tools/gemm_lab.hip
// Run (GPU box):             gpurun_out/gemm_lab [M]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

constexpr int BK = 64, LSTR = 72;
constexpr int NSTAMP = 8, PROBE_ITERS = 24;

struct P { const f16* A; const f16* W; f16* C; int M, N, K; unsigned long long* prof; int probe_wg; };

// VAR 0: product loop (loads of tile k+1 at the top of iteration k, LDS stores at its end, one barrier)
// VAR 1: stores of tile k+1 at the TOP of iteration k (its loads were issued a whole iteration earlier), then loads of tile k+2
// VAR 2: as 1, with the stores / loads split into four groups behind the four 16-deep MFMA steps
template <int BM_, int BN_, int WM_, int WN_, int VAR, bool PROF>
__global__ __launch_bounds__(64 * WM_ * WN_) void gemm_lab_kernel(P p) {
  constexpr int T = 64 * WM_ * WN_;
  constexpr int TM = BM_ / WM_, TN = BN_ / WN_;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int RP = T / 8;
  constexpr int NA = BM_ / RP, NW = BN_ / RP;
  static_assert((NA == 2 || NA == 4) && (NW == 2 || NW == 4), "loader passes");
  __shared__ __attribute__((aligned(16))) f16 sA[2][BM_ * LSTR];
  __shared__ __attribute__((aligned(16))) f16 sW[2][BN_ * LSTR];
  __shared__ unsigned long long sprof[PROF ? PROBE_ITERS * NSTAMP : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_, l31 = lane & 31, hi = lane >> 5;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  int wg;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (wg % nmt) * BM_, n0 = (wg / nmt) * BN_;
  const bool probe = PROF && (int)blockIdx.x == p.probe_wg && tid == 0;
#define STAMP(it, i) do { if (PROF) { if (probe && (it) < PROBE_ITERS) sprof[(it) * NSTAMP + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)

  const int lrow = tid >> 3, lkc = (tid & 7) * 8;
  auto arow = [&](int i) -> const f16* {
    int lm = m0 + lrow + RP * i; if (lm > p.M - 1) lm = p.M - 1;
    return p.A + (int64_t)lm * p.K + lkc;
  };
  const f16* ga0 = arow(0); const f16* ga1 = arow(1);
  const f16* ga2 = arow(NA > 2 ? 2 : 0); const f16* ga3 = arow(NA > 2 ? 3 : 0);
  const f16* gw0 = p.W + (int64_t)(n0 + lrow) * p.K + lkc;
  const int64_t wrp = (int64_t)RP * p.K;
  const int soff = lrow * LSTR + lkc;
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define GLOAD_A(kt) ra0 = *reinterpret_cast<const uint4*>(ga0 + (kt) * BK); ra1 = *reinterpret_cast<const uint4*>(ga1 + (kt) * BK); \
  if (NA > 2) { ra2 = *reinterpret_cast<const uint4*>(ga2 + (kt) * BK); ra3 = *reinterpret_cast<const uint4*>(ga3 + (kt) * BK); }
#define GLOAD_W(kt) rw0 = *reinterpret_cast<const uint4*>(gw0 + (kt) * BK); rw1 = *reinterpret_cast<const uint4*>(gw0 + wrp + (kt) * BK); \
  if (NW > 2) { rw2 = *reinterpret_cast<const uint4*>(gw0 + 2 * wrp + (kt) * BK); rw3 = *reinterpret_cast<const uint4*>(gw0 + 3 * wrp + (kt) * BK); }
#define SSTORE_A(buf) *reinterpret_cast<uint4*>(&sA[buf][soff]) = ra0; *reinterpret_cast<uint4*>(&sA[buf][soff + RP * LSTR]) = ra1; \
  if (NA > 2) { *reinterpret_cast<uint4*>(&sA[buf][soff + 2 * RP * LSTR]) = ra2; *reinterpret_cast<uint4*>(&sA[buf][soff + 3 * RP * LSTR]) = ra3; }
#define SSTORE_W(buf) *reinterpret_cast<uint4*>(&sW[buf][soff]) = rw0; *reinterpret_cast<uint4*>(&sW[buf][soff + RP * LSTR]) = rw1; \
  if (NW > 2) { *reinterpret_cast<uint4*>(&sW[buf][soff + 2 * RP * LSTR]) = rw2; *reinterpret_cast<uint4*>(&sW[buf][soff + 3 * RP * LSTR]) = rw3; }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#define MMA_STEP(cur, kk) do { \
    f16x8 wf[NI], af[MI]; \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(&sW[cur][(wn * TN + i * 32 + l31) * LSTR + (kk) * 16 + hi * 8]); \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f16x8*>(&sA[cur][(wm * TM + i * 32 + l31) * LSTR + (kk) * 16 + hi * 8]); \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0); \
  } while (0)

  const int nk = p.K / BK;
  if (VAR == 0) {
    GLOAD_A(0) GLOAD_W(0) SSTORE_A(0) SSTORE_W(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      STAMP(kt, 0);
      if (kt + 1 < nk) { GLOAD_A(kt + 1) GLOAD_W(kt + 1) }
      STAMP(kt, 1);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) MMA_STEP(cur, kk);
      STAMP(kt, 2);
      if (PROF) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): separates the load wait from the store time in the stamps
      STAMP(kt, 3);
      if (kt + 1 < nk) { SSTORE_A(cur ^ 1) SSTORE_W(cur ^ 1) }
      if (PROF) __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
      STAMP(kt, 4);
      __syncthreads();
      STAMP(kt, 5);
    }
  } else {
    GLOAD_A(0) GLOAD_W(0) SSTORE_A(0) SSTORE_W(0)
    if (nk > 1) { GLOAD_A(1) GLOAD_W(1) }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      STAMP(kt, 0);
      if (VAR == 1) {
        if (PROF) __builtin_amdgcn_s_waitcnt(0x0F70);
        STAMP(kt, 1);
        if (kt + 1 < nk) { SSTORE_A(cur ^ 1) SSTORE_W(cur ^ 1) }
        if (kt + 2 < nk) { GLOAD_A(kt + 2) GLOAD_W(kt + 2) }
        STAMP(kt, 2);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) MMA_STEP(cur, kk);
        STAMP(kt, 3);
      } else {
        // split: A stores + A loads behind step 0, W stores + W loads behind step 1
        MMA_STEP(cur, 0);
        if (kt + 1 < nk) { SSTORE_A(cur ^ 1) }
        if (kt + 2 < nk) { GLOAD_A(kt + 2) }
        MMA_STEP(cur, 1);
        if (kt + 1 < nk) { SSTORE_W(cur ^ 1) }
        if (kt + 2 < nk) { GLOAD_W(kt + 2) }
        MMA_STEP(cur, 2);
        MMA_STEP(cur, 3);
        STAMP(kt, 3);
      }
      if (PROF) __builtin_amdgcn_s_waitcnt(0xC07F);
      STAMP(kt, 4);
      __syncthreads();
      STAMP(kt, 5);
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * TM + mi * 32 + l31;
      if (m < p.M) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int n = n0 + wn * TN + ni * 32 + 8 * r4 + 4 * hi;
          const f16x4 o = {(f16)acc[ni][mi][4 * r4], (f16)acc[ni][mi][4 * r4 + 1], (f16)acc[ni][mi][4 * r4 + 2], (f16)acc[ni][mi][4 * r4 + 3]};
          *reinterpret_cast<f16x4*>(p.C + (size_t)m * p.N + n) = o;
        }
      }
    }
  if (PROF) {
    __syncthreads();
    if (probe) for (int i = 0; i < PROBE_ITERS * NSTAMP; ++i) p.prof[i] = sprof[i];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Ping-pong workgroup: 512 threads = two groups of four waves; group g owns rows [128g, 128g + 128) of a 256 x 128 tile and the
// groups alternate roles every phase (a phase ends in a workgroup barrier): while one group runs its 16 MFMAs on k-tile k, the
// other stores its share of k-tile k+1 to LDS and requests k-tile k+2.  Wave w and wave w+4 share a SIMD, so every SIMD always has
// one wave on the matrix pipe and one on the memory pipes.
//   phase 2k  : G0 MFMA(k) reads buf k&1      | G1 writes tile k+1 -> buf (k+1)&1, loads tile k+2
//   phase 2k+1: G0 writes tile k+1, loads k+2 | G1 MFMA(k) reads buf k&1
// ABL (ablations, results wrong unless 0 / 4): 1 no staging traffic at all, 2 loads only, 3 LDS stores only,
// 4 = two register sets, staggered: waves 0,1 of the staging group request k-tile k+2 first and store k-tile k+1 second, waves 2,3
// the other way round, so the texture path and the LDS store path of the CU work at the same time
template <int BN_, bool PROF, int ABL>
__global__ __launch_bounds__(512) void gemm_pp_kernel(P p) {
  constexpr int BM_ = 256;
  constexpr int TN = BN_ / 2, NI = TN / 32;        // wave tile 64 x TN: group = 2 x 2 waves over 128 x BN_
  constexpr int NWL = BN_ / 64;                     // W loads per thread: a group loads BN_/2 rows = 32 per pass
  __shared__ __attribute__((aligned(16))) f16 sA[2][BM_ * LSTR];
  __shared__ __attribute__((aligned(16))) f16 sW[2][BN_ * LSTR];
  __shared__ unsigned long long sprof[PROF ? PROBE_ITERS * NSTAMP : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, gtid = tid & 255;
  const int wm = (wave & 3) >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  int wg;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (wg % nmt) * BM_, n0 = (wg / nmt) * BN_;
  const bool probe = PROF && (int)blockIdx.x == p.probe_wg && (tid == 0 || tid == 256);
  const int pofs = tid == 256 ? 4 : 0;     // group 1's stamps in slots 4..7
#define STAMP2(it, i) do { if (PROF) { if (probe && (it) < PROBE_ITERS) sprof[(it) * NSTAMP + pofs + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
  const int lrow = gtid >> 3, lkc = (gtid & 7) * 8;
  auto arow = [&](int i) -> const f16* {
    int lm = m0 + 128 * grp + lrow + 32 * i; if (lm > p.M - 1) lm = p.M - 1;
    return p.A + (int64_t)lm * p.K + lkc;
  };
  const f16* ga0 = arow(0); const f16* ga1 = arow(1); const f16* ga2 = arow(2); const f16* ga3 = arow(3);
  const f16* gw0 = p.W + (int64_t)(n0 + (BN_ / 2) * grp + lrow) * p.K + lkc;
  const int64_t wrp = (int64_t)32 * p.K;
  const int soffA = (128 * grp + lrow) * LSTR + lkc, soffW = ((BN_ / 2) * grp + lrow) * LSTR + lkc;
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
  uint4 qa0, qa1, qa2, qa3, qw0, qw1, qw2, qw3;      // second register set (ABL 4)
#define QLOAD(kt) qa0 = *reinterpret_cast<const uint4*>(ga0 + (kt) * BK); qa1 = *reinterpret_cast<const uint4*>(ga1 + (kt) * BK); \
  qa2 = *reinterpret_cast<const uint4*>(ga2 + (kt) * BK); qa3 = *reinterpret_cast<const uint4*>(ga3 + (kt) * BK); \
  qw0 = *reinterpret_cast<const uint4*>(gw0 + (kt) * BK); qw1 = *reinterpret_cast<const uint4*>(gw0 + wrp + (kt) * BK); \
  if (NWL > 2) { qw2 = *reinterpret_cast<const uint4*>(gw0 + 2 * wrp + (kt) * BK); qw3 = *reinterpret_cast<const uint4*>(gw0 + 3 * wrp + (kt) * BK); }
#define QSTORE(buf) *reinterpret_cast<uint4*>(&sA[buf][soffA]) = qa0; *reinterpret_cast<uint4*>(&sA[buf][soffA + 32 * LSTR]) = qa1; \
  *reinterpret_cast<uint4*>(&sA[buf][soffA + 64 * LSTR]) = qa2; *reinterpret_cast<uint4*>(&sA[buf][soffA + 96 * LSTR]) = qa3; \
  *reinterpret_cast<uint4*>(&sW[buf][soffW]) = qw0; *reinterpret_cast<uint4*>(&sW[buf][soffW + 32 * LSTR]) = qw1; \
  if (NWL > 2) { *reinterpret_cast<uint4*>(&sW[buf][soffW + 64 * LSTR]) = qw2; *reinterpret_cast<uint4*>(&sW[buf][soffW + 96 * LSTR]) = qw3; }
#define PLOAD(kt) ra0 = *reinterpret_cast<const uint4*>(ga0 + (kt) * BK); ra1 = *reinterpret_cast<const uint4*>(ga1 + (kt) * BK); \
  ra2 = *reinterpret_cast<const uint4*>(ga2 + (kt) * BK); ra3 = *reinterpret_cast<const uint4*>(ga3 + (kt) * BK); \
  rw0 = *reinterpret_cast<const uint4*>(gw0 + (kt) * BK); rw1 = *reinterpret_cast<const uint4*>(gw0 + wrp + (kt) * BK); \
  if (NWL > 2) { rw2 = *reinterpret_cast<const uint4*>(gw0 + 2 * wrp + (kt) * BK); rw3 = *reinterpret_cast<const uint4*>(gw0 + 3 * wrp + (kt) * BK); }
#define PSTORE(buf) *reinterpret_cast<uint4*>(&sA[buf][soffA]) = ra0; *reinterpret_cast<uint4*>(&sA[buf][soffA + 32 * LSTR]) = ra1; \
  *reinterpret_cast<uint4*>(&sA[buf][soffA + 64 * LSTR]) = ra2; *reinterpret_cast<uint4*>(&sA[buf][soffA + 96 * LSTR]) = ra3; \
  *reinterpret_cast<uint4*>(&sW[buf][soffW]) = rw0; *reinterpret_cast<uint4*>(&sW[buf][soffW + 32 * LSTR]) = rw1; \
  if (NWL > 2) { *reinterpret_cast<uint4*>(&sW[buf][soffW + 64 * LSTR]) = rw2; *reinterpret_cast<uint4*>(&sW[buf][soffW + 96 * LSTR]) = rw3; }
  f32x16 acc[NI][2];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int arow0 = 128 * grp + wm * 64;
#define PMMA(cur) do { \
    _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk) { \
      f16x8 wf[NI], af[2]; \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(&sW[cur][(wn * TN + i * 32 + l31) * LSTR + kk * 16 + hi * 8]); \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(&sA[cur][(arow0 + i * 32 + l31) * LSTR + kk * 16 + hi * 8]); \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0); \
    } } while (0)
  const int nk = p.K / BK;
  const bool early = (wave & 3) < 2;
#define KEEP4(r) asm volatile("" :: "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w));
  // ABL 4: tile k+1 sits in set R when k is even, in set Q when k is odd (tile 1 -> R in the prologue)
#define PMEM(kt, cur) do { \
    if (ABL == 0) { if ((kt) + 1 < nk) { PSTORE((cur) ^ 1) } if ((kt) + 2 < nk) { PLOAD((kt) + 2) } } \
    else if (ABL == 2) { if ((kt) + 2 < nk) { PLOAD((kt) + 2) } KEEP4(ra0) KEEP4(ra1) KEEP4(ra2) KEEP4(ra3) KEEP4(rw0) KEEP4(rw1) if (NWL > 2) { KEEP4(rw2) KEEP4(rw3) } } \
    else if (ABL == 3) { if ((kt) + 1 < nk) { PSTORE((cur) ^ 1) } } \
    else if (ABL == 4) { \
      if (((kt) & 1) == 0) { \
        if (early) { if ((kt) + 2 < nk) { QLOAD((kt) + 2) } if ((kt) + 1 < nk) { PSTORE((cur) ^ 1) } } \
        else       { if ((kt) + 1 < nk) { PSTORE((cur) ^ 1) } if ((kt) + 2 < nk) { QLOAD((kt) + 2) } } \
      } else { \
        if (early) { if ((kt) + 2 < nk) { PLOAD((kt) + 2) } if ((kt) + 1 < nk) { QSTORE((cur) ^ 1) } } \
        else       { if ((kt) + 1 < nk) { QSTORE((cur) ^ 1) } if ((kt) + 2 < nk) { PLOAD((kt) + 2) } } \
      } } \
  } while (0)
  PLOAD(0) PSTORE(0)
  if (nk > 1) { PLOAD(1) }
  __syncthreads();
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      STAMP2(kt, 0);
      PMMA(cur);
      STAMP2(kt, 1);
      __syncthreads();
      STAMP2(kt, 2);
      PMEM(kt, cur);
      STAMP2(kt, 3);
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      STAMP2(kt, 0);
      PMEM(kt, cur);
      STAMP2(kt, 1);
      __syncthreads();
      STAMP2(kt, 2);
      PMMA(cur);
      STAMP2(kt, 3);
      __syncthreads();
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + arow0 + mi * 32 + l31;
      if (m < p.M) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int n = n0 + wn * TN + ni * 32 + 8 * r4 + 4 * hi;
          const f16x4 o = {(f16)acc[ni][mi][4 * r4], (f16)acc[ni][mi][4 * r4 + 1], (f16)acc[ni][mi][4 * r4 + 2], (f16)acc[ni][mi][4 * r4 + 3]};
          *reinterpret_cast<f16x4*>(p.C + (size_t)m * p.N + n) = o;
        }
      }
    }
  if (PROF) {
    __syncthreads();
    if ((int)blockIdx.x == p.probe_wg && tid == 0) for (int i = 0; i < PROBE_ITERS * NSTAMP; ++i) p.prof[i] = sprof[i];
  }
}
template <int BN_, bool PROF, int ABL = 0>
static void launch_pp(const P& p, hipStream_t st) {
  const int nwg = ((p.M + 255) / 256) * (p.N / BN_);
  hipLaunchKernelGGL((gemm_pp_kernel<BN_, PROF, ABL>), dim3(nwg), dim3(512), 0, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------------
// LDS-DMA staging (global_load_lds_dwordx4): operands go HBM/L2 -> LDS without passing through VGPRs or ds_write.  A DMA piece is
// one wave instruction = 64 lanes x 16 B = 8 tile rows x 128 B, written lane-linear, so the LDS image is unpadded [rows][64] f16
// and bank conflicts are avoided by an XOR swizzle applied on the SOURCE address: 16-byte chunk c of row r lands in slot
// c ^ ((r >> 1) & 7) (conflict-free for the four 16-lane groups ds_read_b128 is served in).  NBUF = 3: two k-tiles in flight
// across the single barrier per k-tile (counted vmcnt, raw s_barrier); ONE __shared__ array (a second object makes hipcc drain
// vmcnt before every fragment read).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int BM_, int BN_, int WM_, int WN_, int NBUF>
__global__ __launch_bounds__(64 * WM_ * WN_) void gemm_g_kernel(P p) {
  constexpr int NWAVE = WM_ * WN_;
  constexpr int TM = BM_ / WM_, TN = BN_ / WN_, MI = TM / 32, NI = TN / 32;
  constexpr int TILE_A = BM_ * 64, STAGE = (BM_ + BN_) * 64;      // f16 elements
  constexpr int PA = BM_ / 8 / NWAVE, PW = BN_ / 8 / NWAVE;       // DMA pieces per wave and k-tile
  static_assert(PA >= 1 && PW >= 1 && PA <= 4 && PW <= 4, "pieces");
  __shared__ __attribute__((aligned(1024))) f16 smem[NBUF * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_, l31 = lane & 31, hi = lane >> 5;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  int wg;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (wg % nmt) * BM_, n0 = (wg / nmt) * BN_;
  const int prow = lane >> 3, slot = lane & 7;
  const f16* gA[PA]; const f16* gW[PW];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int row = (wave * PA + j) * 8 + prow;
    int m = m0 + row; if (m > p.M - 1) m = p.M - 1;
    gA[j] = p.A + (int64_t)m * p.K + ((slot ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int row = (wave * PW + j) * 8 + prow;
    gW[j] = p.W + (int64_t)(n0 + row) * p.K + ((slot ^ ((row >> 1) & 7)) << 3);
  }
  auto issue = [&](int kt, int buf) {
    f16* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PA; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gA[j] + kt * BK), (lds_ptr_t)(base + (wave * PA + j) * 8 * 64), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gW[j] + kt * BK), (lds_ptr_t)(base + TILE_A + (wave * PW + j) * 8 * 64), 16, 0, 0);
  };
  // fragment read offsets (f16 elements inside a stage): row * 64 + ((2 kk + hi) ^ sw) * 8
  const int sw = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = ((2 * kk + hi) ^ sw) << 3;
  const int arow = (wm * TM + l31) * 64, wrow = TILE_A + (wn * TN + l31) * 64;
  f32x16 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int nk = p.K / BK;
  issue(0, 0);
  if (NBUF == 3 && nk > 1) issue(1, 1);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (NBUF == 3 && kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (PA + PW)); else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    if (NBUF == 3) { if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1); }
    else { if (kt + 1 < nk) issue(kt + 1, buf ^ 1); }
    const f16* st = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f16x8 wf[NI], af[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const f16x8*>(st + wrow + i * 32 * 64 + koff[kk]);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f16x8*>(st + arow + i * 32 * 64 + koff[kk]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    if (NBUF == 3) buf = buf == 2 ? 0 : buf + 1; else buf ^= 1;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * TM + mi * 32 + l31;
      if (m < p.M) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int n = n0 + wn * TN + ni * 32 + 8 * r4 + 4 * hi;
          const f16x4 o = {(f16)acc[ni][mi][4 * r4], (f16)acc[ni][mi][4 * r4 + 1], (f16)acc[ni][mi][4 * r4 + 2], (f16)acc[ni][mi][4 * r4 + 3]};
          *reinterpret_cast<f16x4*>(p.C + (size_t)m * p.N + n) = o;
        }
      }
    }
}
template <int BM, int BN, int WM, int WN, int NBUF>
static void launch_g(const P& p, hipStream_t st) {
  const int nwg = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL((gemm_g_kernel<BM, BN, WM, WN, NBUF>), dim3(nwg), dim3(64 * WM * WN), 0, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------------
// 8-phase 256 x 256 workgroup (cdna_hip_programming.md, "The 256^2 8-phase template"): 512 threads = 8 waves as 2 (M) x 4 (N), wave
// tile 128 x 64 on v_mfma_f32_16x16x32_f16, BK = 64, LDS-DMA staging (global_load_lds_dwordx4) into two 64 KiB k-tile buffers.
// A k-tile is staged as FOUR half-tiles of 128 rows x 128 B, in the order of first use:
//   h0 = N-lo (weight rows wc*64 + 0..31 of every wave column), h1 = M-lo (activation rows wr*128 + 0..63 of both wave rows),
//   h2 = N-hi (wc*64 + 32..63), h3 = M-hi (wr*128 + 64..127)
// and computed in four phases of 16 MFMAs (one 64 x 32 quadrant of the wave tile each over the 64-deep k-tile):
//   phase 0: reads N-lo (4 x ds_read_b128) then M-lo (8), stages h3 of tile t+1    -> MFMA M-lo x N-lo
//   phase 1: reads N-hi (4),                       stages h0 of tile t+2    -> MFMA M-lo x N-hi
//   phase 2: reads M-hi (8),                       stages h1 of tile t+2    -> MFMA M-hi x N-hi
//   phase 3: no reads,                             stages h2 of tile t+2, vmcnt(6) (tile t+1 landed) -> MFMA M-hi x N-lo
// i.e. the staging runs SEVEN half-tiles ahead of the compute, three of them still in flight across the once-per-k-tile counted
// wait; a phase is {reads + stage, s_barrier, lgkmcnt(0), setprio 1, 16 MFMAs, setprio 0, s_barrier} with RAW barriers (a
// __syncthreads() would drain the LDS-DMA queue).  The two wave rows run staggered by one barrier (wave row 1 executes one extra
// s_barrier up front): on every SIMD one wave is in its MFMA segment while the other reads / stages.  LDS image of a half-tile:
// [128 rows][64 f16] unpadded (the DMA writes lane-linear), 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7): the swizzle is
// applied to the per-lane SOURCE address and to the fragment read address (conflict-free for the four 16-lane groups a
// ds_read_b128 is served in).  Restaging distances (with the stagger a region may be restaged two phases after its last read, or
// one phase after when the read was retired before the reading phase's first barrier - N-lo, by the lgkmcnt(8) of phase 0).
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// EP: 0 = every lane stores its 4-column pieces (8 B) straight from the accumulators; 1 = no stores (ablation: what the epilogue costs);
// 2 = the wave tile goes through LDS (XOR-swizzled [128][64] f16 image per wave) and leaves as 16-byte stores of whole 128-byte row segments.
// The kernel loops over output tiles (tile = virtual workgroup id, + gridDim.x per round): launched with one workgroup per tile it is the
// one-tile-per-workgroup form, launched with min(tiles, 256) workgroups it is PERSISTENT - the next tile's prologue DMAs are requested
// before this tile's stores (EP 0), which then drain while the matrix cores work on the next tile.
template <bool STAGGER, bool PRIO, int EP>
__global__ __launch_bounds__(512) void gemm_8p_kernel(P p) {
  constexpr int BM_ = 256, BN_ = 256, HALF = 128 * 64, BUF = 4 * HALF;
  __shared__ __attribute__((aligned(1024))) f16 smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, kq = lane >> 4;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  const int nk = p.K / BK;
  const f16 *sNl0, *sNl1, *sNh0, *sNh1, *sMl0, *sMl1, *sMh0, *sMh1;
  int m0 = 0, n0 = 0;
  auto setup = [&](int v) {
    const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    m0 = (wg % nmt) * BM_; n0 = (wg / nmt) * BN_;
    auto src = [&](int j, const f16** nl, const f16** nh, const f16** ml, const f16** mh) {
      const int rl = (wave * 2 + j) * 8 + (lane >> 3), c = (lane & 7) ^ ((rl >> 1) & 7);
      const int n = n0 + (rl >> 5) * 64 + (rl & 31);
      *nl = p.W + (int64_t)n * p.K + c * 8;
      *nh = p.W + (int64_t)(n + 32) * p.K + c * 8;
      int mlo = m0 + (rl >> 6) * 128 + (rl & 63), mhi = mlo + 64;
      if (mlo > p.M - 1) mlo = p.M - 1;
      if (mhi > p.M - 1) mhi = p.M - 1;
      *ml = p.A + (int64_t)mlo * p.K + c * 8;
      *mh = p.A + (int64_t)mhi * p.K + c * 8;
    };
    src(0, &sNl0, &sNh0, &sMl0, &sMh0);
    src(1, &sNl1, &sNh1, &sMl1, &sMh1);
  };
#define WIS_DMA(src, dst) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src), (lds_ptr_t)(dst), 16, 0, 0)
#define WIS_STAGE(PA, PB, h, bo) do { \
    f16* d_ = smem + (bo) + (h) * HALF + wave * 1024; \
    WIS_DMA(PA, d_); WIS_DMA(PB, d_ + 512); PA += BK; PB += BK; } while (0)
#define WIS_PROLOGUE() do { \
    WIS_STAGE(sNl0, sNl1, 0, 0); WIS_STAGE(sMl0, sMl1, 1, 0); WIS_STAGE(sNh0, sNh1, 2, 0); WIS_STAGE(sMh0, sMh1, 3, 0); \
    WIS_STAGE(sNl0, sNl1, 0, BUF); WIS_STAGE(sMl0, sMl1, 1, BUF); WIS_STAGE(sNh0, sNh1, 2, BUF); } while (0)
  const int fo0 = l15 * 64 + ((kq ^ (l15 >> 1)) << 3);
  const int oN0 = wc * 32 * 64 + fo0, oM0 = wr * 64 * 64 + fo0;
  f32x4v acc[8][4];
  f16x8v nlo[2][2], nhi[2][2], mlo[4][2], mhi[4][2];
#define WIS_FRAG(base, h, blk, kb) (*reinterpret_cast<const f16x8v*>(smem + ((kb) ? base##1 : base##0) + (h) * HALF + (blk) * 1024))
#define WIS_MMA16(MF, NF, MB0, NB0) do { \
    __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    if (PRIO) __builtin_amdgcn_s_setprio(1); \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
      _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
          acc[(MB0) + mb][(NB0) + nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(NF[nb][kb], MF[mb][kb], acc[(MB0) + mb][(NB0) + nb], 0, 0, 0); \
    if (PRIO) __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
  int v = blockIdx.x;
  setup(v);
  WIS_PROLOGUE();
  for (;;) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6): at most 6 operations outstanding => k-tile 0's eight DMAs are done
    __builtin_amdgcn_s_barrier();
    if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one barrier behind wave row 0
    int cb = 0;      // element offset of the buffer holding k-tile t
    for (int t = 0; t < nk; ++t) {
      const int rNb0 = oN0 + cb, rNb1 = rNb0 ^ 32, rMb0 = oM0 + cb, rMb1 = rMb0 ^ 32;
      const int ob = cb ^ BUF;      // the other buffer
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) nlo[nb][kb] = WIS_FRAG(rNb, 0, nb, kb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mlo[mb][kb] = WIS_FRAG(rMb, 1, mb, kb);
      if (t + 1 < nk) WIS_STAGE(sMh0, sMh1, 3, ob);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8)
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mlo, nlo, 0, 0);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) nhi[nb][kb] = WIS_FRAG(rNb, 2, nb, kb);
      if (t + 2 < nk) WIS_STAGE(sNl0, sNl1, 0, cb);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mlo, nhi, 0, 2);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mhi[mb][kb] = WIS_FRAG(rMb, 3, mb, kb);
      if (t + 2 < nk) WIS_STAGE(sMl0, sMl1, 1, cb);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mhi, nhi, 4, 2);
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nk) { WIS_STAGE(sNh0, sNh1, 2, cb); __builtin_amdgcn_s_waitcnt(0x0F76); }
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mhi, nlo, 4, 0);
      __builtin_amdgcn_s_barrier();
      cb = ob;
    }
    if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();      // pairs with wave row 1's last barrier
    const int mt = m0, ntl = n0, vn = v + (int)gridDim.x;
    const bool more = vn < nwg;
    if (more && EP != 2) { setup(vn); WIS_PROLOGUE(); }      // (the LDS epilogue needs the buffers itself: its prologue follows it)
    // D[i = n][j = m]: lane holds m = l15, n = 4 kq + r
    if (EP == 1) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" :: "v"(acc[mb][nb][0]), "v"(acc[mb][nb][1]), "v"(acc[mb][nb][2]), "v"(acc[mb][nb][3]));
    } else if (EP == 2) {
      f16* reg = smem + wave * 8192;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const int row = (mb >> 2) * 64 + (mb & 3) * 16 + l15;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int col = (nb >> 1) * 32 + (nb & 1) * 16 + 4 * kq;
          const f16x4 o = {(f16)acc[mb][nb][0], (f16)acc[mb][nb][1], (f16)acc[mb][nb][2], (f16)acc[mb][nb][3]};
          *reinterpret_cast<f16x4*>(reg + row * 64 + (((col >> 3) ^ ((row >> 1) & 7)) << 3) + (col & 7)) = o;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = i * 8 + (lane >> 3), ch = lane & 7;
        const uint4 vv = *reinterpret_cast<const uint4*>(reg + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3));
        const int m = mt + wr * 128 + row;
        if (m < p.M) *reinterpret_cast<uint4*>(p.C + (size_t)m * p.N + ntl + wc * 64 + ch * 8) = vv;
      }
      if (more) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); setup(vn); WIS_PROLOGUE(); }      // every wave's LDS reads are done before the DMAs overwrite them
    } else {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const int m = mt + wr * 128 + (mb >> 2) * 64 + (mb & 3) * 16 + l15;
        if (m < p.M) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            const int n = ntl + wc * 64 + (nb >> 1) * 32 + (nb & 1) * 16 + 4 * kq;
            const f16x4 o = {(f16)acc[mb][nb][0], (f16)acc[mb][nb][1], (f16)acc[mb][nb][2], (f16)acc[mb][nb][3]};
            *reinterpret_cast<f16x4*>(p.C + (size_t)m * p.N + n) = o;
          }
        }
      }
    }
    if (!more) break;
    v = vn;
  }
#undef WIS_MMA16
#undef WIS_FRAG
#undef WIS_PROLOGUE
#undef WIS_STAGE
#undef WIS_DMA
}
template <bool STAGGER, bool PRIO, int EP = 0, bool PERSIST = false>
static void launch_8p(const P& p, hipStream_t st) {
  int nwg = ((p.M + 255) / 256) * (p.N / 256);
  if (PERSIST && nwg > 256) nwg = 256;
  hipLaunchKernelGGL((gemm_8p_kernel<STAGGER, PRIO, EP>), dim3(nwg), dim3(512), 0, st, p);
}

struct Shape { const char* name; int N, K; };

template <int BM, int BN, int WM, int WN, int VAR, bool PROF>
static void launch(const P& p, hipStream_t st) {
  const int nwg = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL((gemm_lab_kernel<BM, BN, WM, WN, VAR, PROF>), dim3(nwg), dim3(64 * WM * WN), 0, st, p);
}
typedef void (*LaunchFn)(const P&, hipStream_t);

// weights rotate through a pool larger than the Infinity Cache (every launch streams its W from HBM, as in the encoder, where a
// layer's weights were last touched an utterance ago); the A operand stays warm (the previous kernel just produced it)
static const f16* g_wpool[64]; static int g_nrot = 1;
static double time_us(LaunchFn fn, const P& p0, hipStream_t st, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  P p = p0;
  for (int i = 0; i < 5; ++i) { p.W = g_wpool[i % g_nrot]; fn(p, st); }
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) { p.W = g_wpool[(i + 5) % g_nrot]; fn(p, st); }
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1000.0 * ms / reps;
}

static void fill(std::vector<f16>& v, uint32_t seed) {
  uint32_t s = seed;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (f16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f); }
}

static void run_stamps(LaunchFn fn, P p, hipStream_t st, const char* tag) {
  unsigned long long* dprof; CK(hipMalloc(&dprof, PROBE_ITERS * NSTAMP * 8)); CK(hipMemset(dprof, 0, PROBE_ITERS * NSTAMP * 8));
  p.prof = dprof;
  for (int i = 0; i < 3; ++i) fn(p, st);
  CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> h(PROBE_ITERS * NSTAMP);
  CK(hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost));
  const int nk = p.K / BK < PROBE_ITERS ? p.K / BK : PROBE_ITERS;
  printf("  stamps %s (probe wg %d; s_memtime ticks, per iteration: d01 d12 d23 d34 d45 | iteration total)\n", tag, p.probe_wg);
  double sum[NSTAMP] = {0};
  for (int it = 2; it < nk - 2; ++it) {
    const unsigned long long* r = &h[it * NSTAMP];
    for (int j = 0; j + 1 < NSTAMP; ++j) { const long long dlt = (long long)(r[j + 1] - r[j]); sum[j] += (dlt > -100000 && dlt < 100000) ? (double)dlt : 0.0; }
    sum[NSTAMP - 1] += (double)(long long)(h[(it + 1) * NSTAMP] - r[0]);
  }
  const int cnt = nk - 4 > 0 ? nk - 4 : 1;
  printf("    steady mean deltas:");
  for (int j = 0; j + 1 < NSTAMP; ++j) printf(" %6.0f", sum[j] / cnt);
  printf(" | iteration %6.0f\n", sum[NSTAMP - 1] / cnt);
  CK(hipFree(dprof));
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1500;
  const int d = argc > 4 ? atoi(argv[4]) : 1280;
  const Shape shapes[4] = {{"qkv ", 3 * d, d}, {"out ", d, d}, {"ffn1", 4 * d, d}, {"ffn2", d, 4 * d}};
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<f16> hA((size_t)M * 4 * d), hW((size_t)4 * d * 4 * d);
  fill(hA, 1); fill(hW, 2);
  f16 *dA, *dW, *dC, *dC0;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2));
  CK(hipMalloc(&dC, (size_t)M * 4 * d * 2)); CK(hipMalloc(&dC0, (size_t)M * 4 * d * 2));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  struct V { const char* name; LaunchFn fn; LaunchFn fn_prof; int bm, bn; };
  const V vs[] = {
    {"128x128 v0", launch<128, 128, 2, 2, 0, false>, launch<128, 128, 2, 2, 0, true>, 128, 128},
    {"128x128 v1", launch<128, 128, 2, 2, 1, false>, launch<128, 128, 2, 2, 1, true>, 128, 128},
    {" 64x128 v0", launch<64, 128, 2, 2, 0, false>, launch<64, 128, 2, 2, 0, true>, 64, 128},
    {" 64x128 v1", launch<64, 128, 2, 2, 1, false>, launch<64, 128, 2, 2, 1, true>, 64, 128},
    {"128x128 g2", launch_g<128, 128, 2, 2, 2>, nullptr, 128, 128},
    {"256x128 g3", launch_g<256, 128, 4, 2, 3>, nullptr, 256, 128},
    {"256x128 pp", launch_pp<128, false>, launch_pp<128, true>, 256, 128},
    {"256x256 pp", launch_pp<256, false>, launch_pp<256, true>, 256, 256},
    {"256x256 v0", launch<256, 256, 2, 4, 0, false>, launch<256, 256, 2, 4, 0, true>, 256, 256},
    {"256x256 v1", launch<256, 256, 2, 4, 1, false>, launch<256, 256, 2, 4, 1, true>, 256, 256},
    {"256x256 8p ", launch_8p<true, true>, nullptr, 256, 256},
    {"256x256 8p-nostagger", launch_8p<false, true>, nullptr, 256, 256},
    {"256x256 8p-noprio", launch_8p<true, false>, nullptr, 256, 256},
    {"256x256 8p-nostore (ablation: wrong)", launch_8p<true, true, 1>, nullptr, 256, 256},
    {"256x256 8p-ldsep", launch_8p<true, true, 2>, nullptr, 256, 256},
    {"256x256 8p-persist", launch_8p<true, true, 0, true>, nullptr, 256, 256},
    {"256x256 8p-persist-ldsep", launch_8p<true, true, 2, true>, nullptr, 256, 256},
    {"256x256 8p-persist-nostore (ablation: wrong)", launch_8p<true, true, 1, true>, nullptr, 256, 256},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]);
  const bool stamps = argc > 2 ? atoi(argv[2]) != 0 : true;
  const bool cold = argc > 3 ? atoi(argv[3]) != 0 : true;
  for (const Shape& s : shapes) {
    const size_t wbytes = (size_t)s.N * s.K * 2;
    g_nrot = cold ? (int)((640u << 20) / wbytes) : 1; if (g_nrot > 64) g_nrot = 64; if (g_nrot < 1) g_nrot = 1;
    g_wpool[0] = dW;
    for (int i = 1; i < g_nrot; ++i) { f16* w; CK(hipMalloc(&w, wbytes)); CK(hipMemcpy(w, dW, wbytes, hipMemcpyDeviceToDevice)); g_wpool[i] = w; }
    P p{dA, dW, dC0, M, s.N, s.K, nullptr, 0};
    const double gflop = 2.0 * M * s.N * (double)s.K * 1e-9;
    printf("%s M=%d N=%d K=%d  (%.1f GFLOP)\n", s.name, M, s.N, s.K, gflop);
    // reference: variant 0 of the 128x128 tile; spot-check it on the host
    launch<128, 128, 2, 2, 0, false>(p, st); CK(hipStreamSynchronize(st));
    std::vector<f16> c0((size_t)M * s.N); CK(hipMemcpy(c0.data(), dC0, c0.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
      const int m = (t * 977) % M, n = (t * 613 + 5) % s.N;
      double ref = 0; for (int k = 0; k < s.K; ++k) ref += (double)(float)hA[(size_t)m * s.K + k] * (double)(float)hW[(size_t)n * s.K + k];
      const double e = fabs(ref - (double)(float)c0[(size_t)m * s.N + n]); if (e > worst) worst = e;
    }
    printf("  host spot check of v0: max abs err %.4f\n", worst);
    for (int v = 0; v < nv; ++v) {
      if (s.N % vs[v].bn) continue;
      const int tiles = ((M + vs[v].bm - 1) / vs[v].bm) * (s.N / vs[v].bn);
      P q = p; q.C = dC;
      CK(hipMemsetAsync(dC, 0, (size_t)M * s.N * 2, st));
      vs[v].fn(q, st); CK(hipStreamSynchronize(st));
      std::vector<f16> c((size_t)M * s.N); CK(hipMemcpy(c.data(), dC, c.size() * 2, hipMemcpyDeviceToHost));
      size_t bad = 0; double maxd = 0; for (size_t i = 0; i < c.size(); ++i) { bad += memcmp(&c[i], &c0[i], 2) != 0; const double dd = fabs((double)(float)c[i] - (double)(float)c0[i]); if (!(dd <= maxd)) maxd = dd; }
      const double us = time_us(vs[v].fn, q, st, 40);
      printf("  %s  tiles %4d  %7.2f us  %6.0f TFLOP/s  mismatches vs v0: %zu (max abs diff %.4g)\n", vs[v].name, tiles, us, gflop / us * 1e3, bad, maxd);
      if (stamps && vs[v].fn_prof && (&s == &shapes[0] || &s == &shapes[3])) { q.probe_wg = tiles / 3; run_stamps(vs[v].fn_prof, q, st, vs[v].name); }
    }
    for (int i = 1; i < g_nrot; ++i) CK(hipFree((void*)g_wpool[i]));
  }
  return 0;
}
