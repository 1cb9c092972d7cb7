// enc_kernels.hip — Whisper encoder kernels for gfx950.
//
// The reference runs this inside ctranslate2.models.Whisper.generate (call site reference
// main.py:687-693; forward math SURVEY §3.4 / Appendix B): conv1+GELU, conv2(stride 2)+GELU
// +sinusoid positions, L x pre-LN {fused QKV, non-causal MHA over T=1500, out-proj, GELU-FFN},
// final LN, then per decoder layer the cross-attention K/V projection of the encoder memory.
//
// MI355X mapping:
//  * one MFMA GEMM template (v_mfma_f32_32x32x16_f16, fp32 accumulate, 64-deep k-tiles, 64-wide
//    waves with a 64x64 or 128x64 wave tile, LDS double-buffered with register staging, 144-byte
//    row pitch = conflict-free ds_read_b128 fragment reads; tile shapes chosen by row count) serves conv1/conv2 (implicit im2col:
//    the time-major, zero-padded activation makes a k=3 window ONE contiguous 3C-long row, so
//    a conv is a GEMM whose A rows overlap: row pitch C for stride 1, 2C for stride 2), QKV,
//    out-proj, FFN and the cross-K/V projection.  The weight tile is the MFMA A operand and the
//    activation tile the B operand, so each lane ends up with 4 CONSECUTIVE output features of
//    one row: bias / GELU / residual / position add / layout scatter are fused in the epilogue.
//  * the residual stream is fp32 in HBM (LayerNorm reads fp32, writes fp16 for the next GEMM).
//  * attention is flash-style per (128 queries, head): S^T = K.Q^T so a lane owns ONE query
//    (softmax max/sum are in-lane + one half-swap), P feeds the PV MFMA straight from the
//    accumulator registers (the key order inside a 16-key MFMA step is permuted identically for
//    P and V^T), V arrives pre-transposed ([dh][T]) from the QKV epilogue.
#include <stdlib.h>

#include "common.hpp"
#include "kernels.hpp"

namespace wis {

// GEMM kernels take the GemmP fields as leading SCALAR arguments (rebuilt into the struct inside): hipcc's kernarg preload
// (-amdgpu-kernarg-preload-count, build.py) hands scalars / pointers to the wave in SGPRs at launch but never a by-value struct, and the
// first DMA / fragment addresses need A, W, the shape and the row mapping before anything else.  12 + 2 dwords fit the preload window;
// n_phase (an epilogue quantity) is the first one outside it.
#define WIS_GP_DECL(s) const f16* s##A, int64_t s##a_bs, int s##a_rs, int s##a_rpb, const f16* s##W, int s##M, int s##N, int s##K, int s##klen, int s##n_span, int s##n_period, int s##n_phase
#define WIS_GP_ARGS(q) (q).A, (q).a_bs, (q).a_rs, (q).a_rpb, (q).W, (q).M, (q).N, (q).K, (q).klen, (q).n_span, (q).n_period, (q).n_phase
#define WIS_GP_MAKE(p, s) GemmP p; p.A = s##A; p.a_bs = s##a_bs; p.a_rs = s##a_rs; p.a_rpb = s##a_rpb; p.W = s##W; p.M = s##M; p.N = s##N; p.K = s##K; p.klen = s##klen; \
                          p.n_span = s##n_span; p.n_period = s##n_period; p.n_phase = s##n_phase


// =======================================================================================
// LayerNorm: fp32 [M][d] -> f16 [M][d]; one 64-lane wave per row, two-pass in registers.
#define WIS_PIN4(r) asm volatile("" :: "v"((r).x), "v"((r).y), "v"((r).z), "v"((r).w))
template <bool AFFINE>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, f16* __restrict__ y, int M, int d) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int n4 = d >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)row * d);
  // loads are unconditional (clamped index, masked value): a guarded `if (idx < n4) v = x4[idx]` per float4 compiles into a branch
  // and a wait each, which serialises the row's eight memory round trips
  // (the same goes for gamma / beta: requested with the row, not one round trip per output quad behind the statistics)
  float4 v[8], g[AFFINE ? 8 : 1], be[AFFINE ? 8 : 1];
  const float4* g4 = reinterpret_cast<const float4*>(gamma);      // !AFFINE: plain normalisation (the affine part is folded into
  const float4* b4 = reinterpret_cast<const float4*>(beta);       // the weights of the projection that follows)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i, ic = idx < n4 ? idx : n4 - 1;
    float4 a = x4[ic];
    if (AFFINE) { g[i] = g4[ic]; be[i] = b4[ic]; }
    if (idx >= n4) a = make_float4(0.f, 0.f, 0.f, 0.f);
    v[i] = a;
    s += (a.x + a.y) + (a.z + a.w);
  }
  if (AFFINE) {      // pin the gamma / beta requests here: hipcc otherwise sinks them below the statistics (a second round trip)
#pragma unroll
    for (int i = 0; i < 8; ++i) { WIS_PIN4(g[i]); WIS_PIN4(be[i]); }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    const float t = (a * a + b * b) + (c * c + e * e);
    q += idx < n4 ? t : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
  f16x4* y4 = reinterpret_cast<f16x4*>(y + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    const float4 gg = AFFINE ? g[i] : make_float4(1.f, 1.f, 1.f, 1.f), bb = AFFINE ? be[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    f16x4 o;
    o[0] = (f16)((v[i].x - mean) * rstd * gg.x + bb.x);
    o[1] = (f16)((v[i].y - mean) * rstd * gg.y + bb.y);
    o[2] = (f16)((v[i].z - mean) * rstd * gg.z + bb.z);
    o[3] = (f16)((v[i].w - mean) * rstd * gg.w + bb.w);
    if (idx < n4) y4[idx] = o;
  }
}

int launch_layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, f16* y, int M, int d) {
  if (d % 4 || d > 2048) { set_error("layernorm: d=%d unsupported", d); return WIS_E_UNSUPPORTED; }
  if (gamma) hipLaunchKernelGGL((layernorm_kernel<true>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, y, M, d);
  else hipLaunchKernelGGL((layernorm_kernel<false>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, y, M, d);
  return WIS_OK;
}

// =======================================================================================
// GEMM  C[m][n] = epi( sum_k A(m)[k] * W[n][k] ),  A rows addressed as
//   A + (m / a_rpb) * a_bs + (m % a_rpb) * a_rs     (implicit im2col for the convs).
constexpr int BK = 64, LSTR = 72;  // 64-deep k-tiles; LDS row pitch 72 f16 = 144 B (conflict-free ds_read_b128)
// BM_ = 128 (default) or 64: the 64-row variant doubles the workgroup count for the N = d GEMMs (out-proj, conv2), which
// would otherwise launch only ceil(1500/128) * d/128 = 120 workgroups on 256 CUs.
// Tile order: the grid is 1-D; workgroup ids are first regrouped so that the ids an XCD receives (id % 8, observed dispatch
// order — a speed assumption only) form one contiguous range, then mapped m-fastest, so an XCD works on a few W column
// panels (<= ~2 MB, L2-resident) against all of A.

// Tile shapes (BM_ x BN_, WM_ x WN_ waves, wave tile (BM_/WM_) x (BN_/WN_)); gemm_pick_tile chooses by row count:
//   128x128, 2x2 waves   default (two workgroups co-reside per CU and cover each other's waits)
//    64x128, 2x2 waves   N = d GEMMs of one utterance (out-proj, conv2): 240 instead of 120 workgroups
//   256x256, 2x4 waves   from 150 tiles on (two or more utterances): 128 FLOP per operand byte fetched instead of 64
//   256x128 ping-pong    gemm_pp_kernel below: 128-256 tiles (one utterance, N = 3d / 4d, the K-slices of FFN2)
// What bounds the loop (tools/gemm_lab.hip: the variants side by side on the four per-layer shapes, weights rotated through
// 640 MB so that W streams from HBM as it does in the encoder; shader-clock stamps of one wave):
//  * one k-tile of the 128x128 workgroup = 415 cycles issuing its 8 loads per thread (the CU's texture path takes a 1 KiB wave
//    request per ~13-16 cycles) + 810 for 16 MFMAs and their fragment reads (512 of MFMA pipe) + 70 waiting for the loads + 400
//    for the 8 ds_write_b128 (13 cycles per wave instruction on the CU's one LDS store path) + 135 barrier = 1970: the phases of
//    a workgroup run one after the other, all four waves being in the same phase;
//  * per CU that is 16-20 bytes of operand per cycle for EVERY tile shape and staging mechanism tried - register staging in
//    either order, LDS-DMA (global_load_lds) with 2 or 3 buffers, the ping-pong groups: within 10 % of each other;
//  * with all staging traffic removed (ablation) the ping-pong loop still takes 1820 cycles per 256x128 k-tile (56 % of the MFMA
//    pipe; 256x256: 66 %): fragment-read latency after each barrier and the barriers themselves.  The 256x256 tile runs at 3900
//    cycles per k-tile = 53 %, the figure the guide's plain-HIP 8-phase template reaches on random data; hipBLASLt's 21.7 us on
//    the QKV shape (tools/mm_bench.py) corresponds to ~80 % per CU on 90 such tiles - assembly-level scheduling.
// us per GEMM at M = 1500 | 3000 | 12000 (large-v2 widths, early staging order where this kernel uses it):
//   QKV   128x128 29.8 | 54.5 | 209    256x128 ping-pong 27.6 | 53.2 | 193    256x256 41.3 | 46.3 | 169
//   FFN1  128x128 32.4 | 67.0 | 267    256x128 ping-pong 30.3 | 61.3 | 261    256x256 43.2 | 54.8 | 222
//   out   128x128 20.8 | 22.6 | 65.9    64x128 13.9 | 21.8 | 81.4             256x256 36.9 | 38.1 | 50.3
// Epilogue of the register-staged tiles (32 x 32 MFMA accumulators: lane holds row m = l31 and features (r & 3) + 8 (r >> 2) + 4 hi of
// each 32 x 32 block).  Every operand the functor needs from memory - the bias quads, and the residual quads where the functor has
// them - is requested BEFORE the first store and retired by ONE wait: gfx9 has one in-order counter for loads and stores, so a load
// between stores (the functor called block by block) makes its wait a wait for every store issued before it - the epilogue then
// runs one store round trip per row block (measured on the 8-phase kernel, whose epilogue is built the same way).
#define WIS_PINV4(r) asm volatile("" :: "v"((r)[0]), "v"((r)[1]), "v"((r)[2]), "v"((r)[3]))
template <class Epi, int NI, int MI>
__device__ __forceinline__ void epilogue_32(const Epi& epi, const f32x16 (&acc)[NI][MI], int mrow, int ncol, int M) {
  f32x4 bb[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) bb[ni][r4] = epi.bias4(ncol + ni * 32 + 8 * r4);
  if (epi.has_res()) {
    // residual quads: all of the tile's at once where that is <= 64 registers, otherwise one 32-column block per round
    constexpr int RND = NI * MI <= 4 ? 1 : NI, NPR = NI / RND;
#pragma unroll
    for (int rnd = 0; rnd < RND; ++rnd) {
      f32x4 rr[NPR][MI][4];
#pragma unroll
      for (int nj = 0; nj < NPR; ++nj)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          int m = mrow + mi * 32;
          if (m > M - 1) m = M - 1;      // clamped, unconditional: the guard is on the stores
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) rr[nj][mi][r4] = epi.res4(m, ncol + (rnd * NPR + nj) * 32 + 8 * r4);
        }
#pragma unroll
      for (int nj = 0; nj < NPR; ++nj) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) WIS_PINV4(bb[rnd * NPR + nj][r4]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) WIS_PINV4(rr[nj][mi][r4]);
      }
#pragma unroll
      for (int nj = 0; nj < NPR; ++nj)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = mrow + mi * 32, ni = rnd * NPR + nj;
          if (m < M) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              f32x4 v = {acc[ni][mi][4 * r4], acc[ni][mi][4 * r4 + 1], acc[ni][mi][4 * r4 + 2], acc[ni][mi][4 * r4 + 3]};
              epi.fin4r(m, ncol + ni * 32 + 8 * r4, v, bb[ni][r4], rr[nj][mi][r4]);
            }
          }
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) WIS_PINV4(bb[ni][r4]);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = mrow + mi * 32;
      if (m < M) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f32x4 v = {acc[ni][mi][4 * r4], acc[ni][mi][4 * r4 + 1], acc[ni][mi][4 * r4 + 2], acc[ni][mi][4 * r4 + 3]};
          epi.fin4(m, ncol + ni * 32 + 8 * r4, v, bb[ni][r4]);
        }
      }
    }
}

template <class Epi, int BM_, int BN_, int WM_, int WN_>
__global__ __launch_bounds__(64 * WM_ * WN_) void gemm_f16_kernel(WIS_GP_DECL(g_), Epi epi) {
  WIS_GP_MAKE(p, g_);
  constexpr int T = 64 * WM_ * WN_;            // threads
  constexpr int TM = BM_ / WM_, TN = BN_ / WN_;   // wave tile
  constexpr int MI = TM / 32, NI = TN / 32;    // 32x32 MFMA sub-tiles per wave
  constexpr int RP = T / 8;                    // tile rows covered by one 16-byte-per-thread pass
  constexpr int NA = BM_ / RP, NW = BN_ / RP;  // passes per k-tile (A, W): 2 or 4
  static_assert((NA == 2 || NA == 4) && (NW == 2 || NW == 4), "loader passes");
  __shared__ __attribute__((aligned(16))) f16 sA[2][BM_ * LSTR];
  __shared__ __attribute__((aligned(16))) f16 sW[2][BN_ * LSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_, l31 = lane & 31, hi = lane >> 5;
  // XCD-aware, bijective regrouping of the linear workgroup id (guide T1), then m-fastest tile coordinates
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  int wg;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (wg % nmt) * BM_, n0 = (wg / nmt) * BN_;

  // loader mapping: 16-byte chunk c -> row c>>3, k-offset (c&7)*8; rows lrow + RP*i (explicitly named registers: arrays
  // here get demoted to scratch / LDS by the compiler)
  const int lrow = tid >> 3, lkc = (tid & 7) * 8;
  auto arow = [&](int i) -> const f16* {
    int lm = m0 + lrow + RP * i; if (lm > p.M - 1) lm = p.M - 1;
    return p.A + (int64_t)(lm / p.a_rpb) * p.a_bs + (int64_t)(lm % p.a_rpb) * p.a_rs + lkc;
  };
  const int kbeg = p.klen > 0 ? (int)blockIdx.z * p.klen : 0;
  const f16* ga0 = arow(0) + kbeg; const f16* ga1 = arow(1) + kbeg;
  const f16* ga2 = arow(NA > 2 ? 2 : 0) + kbeg; const f16* ga3 = arow(NA > 2 ? 3 : 0) + kbeg;
  const f16* gw0 = p.W + (int64_t)(n0 + lrow) * p.K + lkc + kbeg;
  const int64_t wrp = (int64_t)RP * p.K;
  const int soff = lrow * LSTR + lkc;
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define WIS_GLOAD(kt)                                                                         \
  ra0 = *reinterpret_cast<const uint4*>(ga0 + (kt) * BK); ra1 = *reinterpret_cast<const uint4*>(ga1 + (kt) * BK);   \
  if (NA > 2) { ra2 = *reinterpret_cast<const uint4*>(ga2 + (kt) * BK); ra3 = *reinterpret_cast<const uint4*>(ga3 + (kt) * BK); } \
  rw0 = *reinterpret_cast<const uint4*>(gw0 + (kt) * BK); rw1 = *reinterpret_cast<const uint4*>(gw0 + wrp + (kt) * BK);   \
  if (NW > 2) { rw2 = *reinterpret_cast<const uint4*>(gw0 + 2 * wrp + (kt) * BK); rw3 = *reinterpret_cast<const uint4*>(gw0 + 3 * wrp + (kt) * BK); }
#define WIS_SSTORE(buf)                                                                       \
  *reinterpret_cast<uint4*>(&sA[buf][soff]) = ra0; *reinterpret_cast<uint4*>(&sA[buf][soff + RP * LSTR]) = ra1;     \
  if (NA > 2) { *reinterpret_cast<uint4*>(&sA[buf][soff + 2 * RP * LSTR]) = ra2; *reinterpret_cast<uint4*>(&sA[buf][soff + 3 * RP * LSTR]) = ra3; } \
  *reinterpret_cast<uint4*>(&sW[buf][soff]) = rw0; *reinterpret_cast<uint4*>(&sW[buf][soff + RP * LSTR]) = rw1;     \
  if (NW > 2) { *reinterpret_cast<uint4*>(&sW[buf][soff + 2 * RP * LSTR]) = rw2; *reinterpret_cast<uint4*>(&sW[buf][soff + 3 * RP * LSTR]) = rw3; }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (p.klen > 0 ? p.klen : p.K) / BK;
  // Staging order.  EARLY (64x128 and 256x256 tiles): k-tile k+1 is stored to LDS at the TOP of iteration k - its loads were
  // requested a whole iteration earlier - and k-tile k+2 is requested right behind the stores, so the request has the whole
  // iteration to land.  The 128x128 tile (two workgroups per CU) keeps "request at the top, store at the bottom": the two
  // co-resident workgroups already cover each other's waits and the early order measured slower there.
  // tools/gemm_lab.hip on MI355X, us per GEMM, late | early order:
  //   M = 1500   64x128: out-proj 15.5 | 13.6   FFN2 (K = 4d) 51.2 | 42.9     128x128: QKV 28.4 | 30.5   FFN1 31.0 | 33.3
  //   M = 12000 256x256: QKV 182 | 169   out-proj 56.3 | 51.2   FFN1 242 | 223   FFN2 190 | 171
  constexpr bool EARLY = BM_ != 128;
  WIS_GLOAD(0) WIS_SSTORE(0)
  if (EARLY && nk > 1) { WIS_GLOAD(1) }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (EARLY) {
      if (kt + 1 < nk) { WIS_SSTORE(cur ^ 1) }
      if (kt + 2 < nk) { WIS_GLOAD(kt + 2) }
    } else if (kt + 1 < nk) { WIS_GLOAD(kt + 1) }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      f16x8 wf[NI], af[MI];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        wf[i] = *reinterpret_cast<const f16x8*>(&sW[cur][(wn * TN + i * 32 + l31) * LSTR + kk * 16 + hi * 8]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const f16x8*>(&sA[cur][(wm * TM + i * 32 + l31) * LSTR + kk * 16 + hi * 8]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    if (!EARLY && kt + 1 < nk) { WIS_SSTORE(cur ^ 1) }
    __syncthreads();
  }
#undef WIS_GLOAD
#undef WIS_SSTORE
  // D[i = n][j = m]: lane holds m = l31, n = (r&3) + 8*(r>>2) + 4*hi
  epilogue_32<Epi, NI, MI>(epi, acc, m0 + wm * TM + l31, n0 + wn * TN + 4 * hi, p.M);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Ping-pong workgroup for row counts that give 128-256 tiles of 256 x 128 (one utterance of the larger models: QKV 180, FFN1 240,
// the four K-slices of FFN2 240): 512 threads = two groups of four waves; group g owns rows [128 g, 128 g + 128) of the tile and
// the groups alternate roles every phase (a phase ends in a workgroup barrier): while one group runs its 16 MFMAs on k-tile k,
// the other stores its share of k-tile k+1 to LDS and requests k-tile k+2.  Wave w and wave w + 4 share a SIMD, so every SIMD has
// one wave on the matrix pipe and one on the memory pipes, with ONE workgroup per CU (the 128 x 128 tile gets that overlap only
// on the CUs that happen to hold two workgroups):
//   phase 2k   : G0 MFMA(k) reads buf k&1          | G1 stores tile k+1 -> buf (k+1)&1, requests tile k+2
//   phase 2k+1 : G0 stores tile k+1, requests k+2  | G1 MFMA(k) reads buf k&1
// Same MFMA order per output element as gemm_f16_kernel: bit-identical results.  tools/gemm_lab.hip, weights streamed from HBM,
// M = 1500: QKV 29.8 -> 27.6 us, FFN1 32.4 -> 30.3 us against the 128 x 128 tile.
template <class Epi>
__global__ __launch_bounds__(512) void gemm_pp_kernel(WIS_GP_DECL(g_), Epi epi) {
  WIS_GP_MAKE(p, g_);
  constexpr int BM_ = 256, BN_ = 128;
  __shared__ __attribute__((aligned(16))) f16 sA[2][BM_ * LSTR];
  __shared__ __attribute__((aligned(16))) f16 sW[2][BN_ * LSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, gtid = tid & 255;
  const int wm = (wave & 3) >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  int wg;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = (wg % nmt) * BM_, n0 = (wg / nmt) * BN_;
  // loader: group g loads rows 128 g + lrow + 32 i (i < 4) of A and rows 64 g + lrow + 32 i (i < 2) of W
  const int lrow = gtid >> 3, lkc = (gtid & 7) * 8;
  const int kbeg = p.klen > 0 ? (int)blockIdx.z * p.klen : 0;
  auto arow = [&](int i) -> const f16* {
    int lm = m0 + 128 * grp + lrow + 32 * i; if (lm > p.M - 1) lm = p.M - 1;
    return p.A + (int64_t)(lm / p.a_rpb) * p.a_bs + (int64_t)(lm % p.a_rpb) * p.a_rs + lkc + kbeg;
  };
  const f16* ga0 = arow(0); const f16* ga1 = arow(1); const f16* ga2 = arow(2); const f16* ga3 = arow(3);
  const f16* gw0 = p.W + (int64_t)(n0 + 64 * grp + lrow) * p.K + lkc + kbeg;
  const int64_t wrp = (int64_t)32 * p.K;
  const int soffA = (128 * grp + lrow) * LSTR + lkc, soffW = (64 * grp + lrow) * LSTR + lkc;
  uint4 ra0, ra1, ra2, ra3, rw0, rw1;
#define WIS_PLOAD(kt) ra0 = *reinterpret_cast<const uint4*>(ga0 + (kt) * BK); ra1 = *reinterpret_cast<const uint4*>(ga1 + (kt) * BK); \
  ra2 = *reinterpret_cast<const uint4*>(ga2 + (kt) * BK); ra3 = *reinterpret_cast<const uint4*>(ga3 + (kt) * BK); \
  rw0 = *reinterpret_cast<const uint4*>(gw0 + (kt) * BK); rw1 = *reinterpret_cast<const uint4*>(gw0 + wrp + (kt) * BK);
#define WIS_PSTORE(buf) *reinterpret_cast<uint4*>(&sA[buf][soffA]) = ra0; *reinterpret_cast<uint4*>(&sA[buf][soffA + 32 * LSTR]) = ra1; \
  *reinterpret_cast<uint4*>(&sA[buf][soffA + 64 * LSTR]) = ra2; *reinterpret_cast<uint4*>(&sA[buf][soffA + 96 * LSTR]) = ra3; \
  *reinterpret_cast<uint4*>(&sW[buf][soffW]) = rw0; *reinterpret_cast<uint4*>(&sW[buf][soffW + 32 * LSTR]) = rw1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int arow0 = 128 * grp + wm * 64;
#define WIS_PMMA(cur) do { \
    _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk) { \
      f16x8 wf[2], af[2]; \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const f16x8*>(&sW[cur][(wn * 64 + i * 32 + l31) * LSTR + kk * 16 + hi * 8]); \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(&sA[cur][(arow0 + i * 32 + l31) * LSTR + kk * 16 + hi * 8]); \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0); \
    } } while (0)
  const int nk = (p.klen > 0 ? p.klen : p.K) / BK;
  WIS_PLOAD(0) WIS_PSTORE(0)
  if (nk > 1) { WIS_PLOAD(1) }
  __syncthreads();
  // both groups execute exactly two barriers per k-tile
  if (grp == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      WIS_PMMA(cur);
      __syncthreads();
      if (kt + 1 < nk) { WIS_PSTORE(cur ^ 1) }
      if (kt + 2 < nk) { WIS_PLOAD(kt + 2) }
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) { WIS_PSTORE(cur ^ 1) }
      if (kt + 2 < nk) { WIS_PLOAD(kt + 2) }
      __syncthreads();
      WIS_PMMA(cur);
      __syncthreads();
    }
  }
#undef WIS_PLOAD
#undef WIS_PSTORE
#undef WIS_PMMA
  epilogue_32<Epi, 2, 2>(epi, acc, m0 + arow0 + l31, n0 + wn * 64 + 4 * hi, p.M);
}
// ---------------------------------------------------------------------------------------------------------------------------
// 8-phase 256 x 256 workgroup for the batched encoder (>= 150 tiles: two or more utterances), after the guide's plain-HIP template
// (cdna_hip_programming.md "The 256^2 8-phase template"; tools/gemm_lab.hip carries the same loop for tuning).  512 threads = 8 waves
// as 2 (M) x 4 (N), wave tile 128 x 64 on v_mfma_f32_16x16x32_f16, BK = 64, operands staged by LDS-DMA (global_load_lds_dwordx4: no
// VGPR round trip, no ds_write pass) into two 64 KiB k-tile buffers.  A k-tile is staged as FOUR half-tiles of 128 rows x 128 B, in
// the order of first use -
//   h0 = N-lo (weight rows wc*64 + 0..31 of every wave column)     h1 = M-lo (activation rows wr*128 + 0..63 of both wave rows)
//   h2 = N-hi (wc*64 + 32..63)                                      h3 = M-hi (wr*128 + 64..127)
// - and computed in four phases of 16 MFMAs (one 64 x 32 quadrant of the wave tile over the 64-deep k-tile each):
//   phase 0: reads N-lo (4 x ds_read_b128), then M-lo (8); stages h3 of k-tile t+1              MFMA M-lo x N-lo
//   phase 1: reads N-hi (4);                          stages h0 of k-tile t+2              MFMA M-lo x N-hi
//   phase 2: reads M-hi (8);                          stages h1 of k-tile t+2              MFMA M-hi x N-hi
//   phase 3: no reads;                                stages h2 of k-tile t+2, vmcnt(6)    MFMA M-hi x N-lo (N-lo kept in registers)
// The staging runs SEVEN half-tiles ahead of the compute and three of them stay in flight across the one counted wait per k-tile
// (never vmcnt(0) in the loop).  A phase is {fragment reads + one half-tile of DMA, s_barrier, lgkmcnt(0), s_setprio 1, 16 MFMAs,
// s_setprio 0, s_barrier} with RAW barriers - __syncthreads() would drain the DMA queue.  The two wave rows run staggered by one
// barrier (wave row 1 executes one extra s_barrier up front): wave w and wave w + 4 share a SIMD, so every SIMD has one wave in its
// MFMA segment while the other reads / stages.  Ordering rules this relies on (guide, "Read a staged buffer one phase AFTER the wait
// that retires it"): k-tile t+1 is read from phase 0 of t+1 on, its wait sits in phase 3 of t, and both wave rows pass a barrier in
// between; with the stagger a region is restaged two phases after its last read (M-lo, N-hi, M-hi) or one phase after when the read
// was retired before the reading phase's first barrier (N-lo: the lgkmcnt(8) of phase 0, its four reads being issued first).
// LDS image of a half-tile: [128 rows][64 f16] unpadded (the DMA writes lane-linear); the 16-byte chunk c of row r sits in slot
// c ^ ((r >> 1) & 7) - applied to the per-lane SOURCE address and to the fragment read address (the same involution on both sides),
// conflict-free for the four 16-lane groups a ds_read_b128 is served in.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// TR: the MFMA operands are swapped - D[i = row][j = feature] - and the tile leaves through Epi::store_t (4 consecutive ROWS of one feature per
// lane): the transposed V images of the QKV and cross-K/V projections, whose plain layout would scatter single 2-byte stores.  A launch is
// either all plain or all transposed tiles (GemmP::n_span / n_period / n_phase select its output columns).
template <class Epi, bool TR>
__global__ __launch_bounds__(512) void gemm_8p_kernel(WIS_GP_DECL(g_), Epi epi) {
  WIS_GP_MAKE(p, g_);
  constexpr int BM_ = 256, BN_ = 256, HALF = 128 * 64, BUF = 4 * HALF;
  __shared__ __attribute__((aligned(1024))) f16 smem[2 * BUF];      // ONE LDS object (a second one makes hipcc drain vmcnt before fragment reads)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, kq = lane >> 4;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  const int kbeg = p.klen > 0 ? (int)blockIdx.z * p.klen : 0;
  const int nk = (p.klen > 0 ? p.klen : p.K) / BK;
  // PERSISTENT over output tiles: the grid is min(tiles, CUs) workgroups and workgroup g computes the tiles g, g + grid, ...  The
  // stores of a tile (128 KiB per workgroup) are issued back to back and left to drain while the matrix cores are already on the
  // next tile, whose first seven half-tiles were requested BEFORE those stores.  (That only holds since the epilogue carries no load
  // between its stores - see the bias / residual handling below: with `load, vmcnt(0), store` per row block every wait also covered
  // the stores before it and the next tile's DMAs, and the loop overlapped nothing.)
  // staging sources: wave w stages local rows (2 w + j) * 8 + (lane >> 3), j = 0, 1, of every half-tile; each pointer is used once
  // per k-tile and advances by one k-tile per use
  const f16 *sNl0, *sNl1, *sNh0, *sNh1, *sMl0, *sMl1, *sMh0, *sMh1;
  int m0 = 0, n0 = 0;
  auto setup = [&](int v) {      // virtual workgroup id -> tile (XCD-aware, bijective regrouping: guide T1; then m-fastest) and its sources
    const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    m0 = (wg % nmt) * BM_; n0 = (wg / nmt) * BN_;
    if (p.n_span) n0 = (n0 / p.n_span) * p.n_period + p.n_phase + n0 % p.n_span;      // (n_span is a multiple of the tile width)
    auto src = [&](int j, const f16** nl, const f16** nh, const f16** ml, const f16** mh) {
      const int rl = (wave * 2 + j) * 8 + (lane >> 3), c = (lane & 7) ^ ((rl >> 1) & 7);
      // which weight row feeds MFMA row i of 16-row block nb of a wave's 32-row half: 8 (i >> 2) + 4 nb + (i & 3), so that the lane
      // holding MFMA rows 4 kq .. 4 kq + 3 of BOTH blocks of a half owns the 8 consecutive features 8 kq .. 8 kq + 7 - its two
      // accumulator quads leave as one 16-byte store (half the store instructions for the f16 outputs)
      const int i16 = rl & 15, nb16 = (rl >> 4) & 1;
      const int n = n0 + (rl >> 5) * 64 + 8 * (i16 >> 2) + 4 * nb16 + (i16 & 3);
      *nl = p.W + (int64_t)n * p.K + kbeg + c * 8;
      *nh = p.W + (int64_t)(n + 32) * p.K + kbeg + c * 8;
      int mlo = m0 + (rl >> 6) * 128 + (rl & 63), mhi = mlo + 64;
      if (mlo > p.M - 1) mlo = p.M - 1;
      if (mhi > p.M - 1) mhi = p.M - 1;
      *ml = p.A + (int64_t)(mlo / p.a_rpb) * p.a_bs + (int64_t)(mlo % p.a_rpb) * p.a_rs + kbeg + c * 8;
      *mh = p.A + (int64_t)(mhi / p.a_rpb) * p.a_bs + (int64_t)(mhi % p.a_rpb) * p.a_rs + kbeg + c * 8;
    };
    src(0, &sNl0, &sNh0, &sMl0, &sMh0);
    src(1, &sNl1, &sNh1, &sMl1, &sMh1);
  };
#define WIS_DMA(src, dst) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src), (lds_ptr_t)(dst), 16, 0, 0)
#define WIS_STAGE(PA, PB, h, bo) do { \
    f16* d_ = smem + (bo) + (h) * HALF + wave * 1024; \
    WIS_DMA(PA, d_); WIS_DMA(PB, d_ + 512); PA += BK; PB += BK; } while (0)
  // k-tile 0 complete, three half-tiles of k-tile 1 in flight (the launcher guarantees nk >= 2)
#define WIS_PROLOGUE() do { \
    WIS_STAGE(sNl0, sNl1, 0, 0); WIS_STAGE(sMl0, sMl1, 1, 0); WIS_STAGE(sNh0, sNh1, 2, 0); WIS_STAGE(sMh0, sMh1, 3, 0); \
    WIS_STAGE(sNl0, sNl1, 0, BUF); WIS_STAGE(sMl0, sMl1, 1, BUF); WIS_STAGE(sNh0, sNh1, 2, BUF); } while (0)
  // fragment read offsets (f16 elements inside a half-tile region): row * 64 + ((4 kb + kq) ^ (l15 >> 1)) * 8; k-block 1 = offset ^ 32
  const int fo0 = l15 * 64 + ((kq ^ (l15 >> 1)) << 3);
  const int oN0 = wc * 32 * 64 + fo0, oM0 = wr * 64 * 64 + fo0;
  f32x4 acc[8][4];
  f16x8 nlo[2][2], nhi[2][2], mlo[4][2], mhi[4][2];
#define WIS_FRAG(base, h, blk, kb) (*reinterpret_cast<const f16x8*>(smem + ((kb) ? base##1 : base##0) + (h) * HALF + (blk) * 1024))
#define WIS_MMA16(MF, NF, MB0, NB0) do { \
    __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_setprio(1); \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
      _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
          acc[(MB0) + mb][(NB0) + nb] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(MF[mb][kb], NF[nb][kb], acc[(MB0) + mb][(NB0) + nb], 0, 0, 0) \
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(NF[nb][kb], MF[mb][kb], acc[(MB0) + mb][(NB0) + nb], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
  int v = blockIdx.x;
  setup(v);
  WIS_PROLOGUE();
  for (;;) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // at most 6 vector-memory operations outstanding => at least 8 of the 14 prologue DMAs are done, i.e. k-tile 0 (loads complete
    // in order among themselves; the previous tile's stores, requested after the prologue, may complete in any order: they can only
    // make this wait longer, never shorter)
    __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one barrier behind wave row 0
    int cb = 0;      // element offset of the buffer holding k-tile t
    // the lane's bias values: requested in phase 2 of the LAST k-tile (nothing is staged there any more; its closing vmcnt(0) retires them) - a
    // load inside the store sequence of the epilogue would cost a vmcnt(0) wait, and on gfx9 that one in-order counter also holds
    // the stores issued before it: the epilogue would run one store round trip per row block
    f32x4 bq[2][2]; float bt[4];
    for (int t = 0; t < nk; ++t) {
      const int rNb0 = oN0 + cb, rNb1 = rNb0 ^ 32, rMb0 = oM0 + cb, rMb1 = rMb0 ^ 32;
      const int ob = cb ^ BUF;      // the other buffer
      // ---- phase 0
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
      __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8): the four N-lo reads are retired before the first barrier
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mlo, nlo, 0, 0);
      __builtin_amdgcn_s_barrier();
      // ---- phase 1
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) nhi[nb][kb] = WIS_FRAG(rNb, 2, nb, kb);
      if (t + 2 < nk) WIS_STAGE(sNl0, sNl1, 0, cb);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mlo, nhi, 0, 2);
      __builtin_amdgcn_s_barrier();
      // ---- phase 2
      if (t + 1 == nk) {      // (the M-lo fragments are dead from here on: registers for the bias values)
        if (TR) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bt[nb] = epi.bias1(n0 + wc * 64 + (nb >> 1) * 32 + 8 * (l15 >> 2) + 4 * (nb & 1) + (l15 & 3));
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) { bq[h][0] = epi.bias4(n0 + wc * 64 + h * 32 + 8 * kq); bq[h][1] = epi.bias4(n0 + wc * 64 + h * 32 + 8 * kq + 4); }
        }
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) mhi[mb][kb] = WIS_FRAG(rMb, 3, mb, kb);
      if (t + 2 < nk) WIS_STAGE(sMl0, sMl1, 1, cb);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mhi, nhi, 4, 2);
      __builtin_amdgcn_s_barrier();
      // ---- phase 3
      if (t + 2 < nk) { WIS_STAGE(sNh0, sNh1, 2, cb); __builtin_amdgcn_s_waitcnt(0x0F76); }      // k-tile t+1 has landed (this wave's share); three half-tiles of t+2 fly on
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      WIS_MMA16(mhi, nlo, 4, 0);
      __builtin_amdgcn_s_barrier();
      cb = ob;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with wave row 1's last barrier: every wave is past its last fragment read
    // next tile's prologue first (the k-tile buffers are free), then this tile's stores
    const int mt = m0, ntl = n0, vn = v + (int)gridDim.x;
    const bool more = vn < nwg;
    // retire the bias loads HERE, on every path (they were requested in the last k-tile and are covered by its vmcnt(0)) - left to the
    // compiler, each row-guarded store block below would wait for them on its own with vmcnt(0), i.e. for the stores of the block
    // before it (and for the next tile's DMAs)
    if (TR) { asm volatile("" :: "v"(bt[0]), "v"(bt[1]), "v"(bt[2]), "v"(bt[3])); }
    else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(bq[h][i][0]), "v"(bq[h][i][1]), "v"(bq[h][i][2]), "v"(bq[h][i][3]));
    }
    if (more) { setup(vn); WIS_PROLOGUE(); }
    if (TR) {
      // D[i = row][j = MFMA column]: lane holds rows 4 kq + r of row block mb and MFMA column l15 of block nb = feature 8 (l15 >> 2) + 4 (nb & 1) + (l15 & 3) of half nb >> 1
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const int m = mt + wr * 128 + (mb >> 2) * 64 + (mb & 3) * 16 + 4 * kq;
        if (m < p.M) {      // (M is a multiple of 4 wherever a transposed epilogue exists: M = utterances x 1500)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) epi.fin_t(m, ntl + wc * 64 + (nb >> 1) * 32 + 8 * (l15 >> 2) + 4 * (nb & 1) + (l15 & 3), acc[mb][nb], bt[nb]);
        }
      }
      if (!more) break;
      v = vn;
      continue;
    }
    // D[i = MFMA row][j = m]: lane holds m = l15 and MFMA rows 4 kq + r of the blocks 2 h, 2 h + 1 = features 8 kq + 4 (nb & 1) + r of half h
    if constexpr (Epi::HAS_RES) {      // residual values of TWO row blocks per round (32 registers: the tile's 128 accumulators stay live), then their stores
#pragma unroll
      for (int rnd = 0; rnd < 4; ++rnd) {
        f32x4 rr[2][2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          int m = mt + wr * 128 + (rnd >> 1) * 64 + ((rnd & 1) * 2 + q) * 16 + l15;
          if (m > p.M - 1) m = p.M - 1;      // clamped, unconditional: the guard is on the stores
#pragma unroll
          for (int h = 0; h < 2; ++h) epi.res8(m, ntl + wc * 64 + h * 32 + 8 * kq, rr[q][h][0], rr[q][h][1]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(rr[q][h][i][0]), "v"(rr[q][h][i][1]), "v"(rr[q][h][i][2]), "v"(rr[q][h][i][3]));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int mb = rnd * 2 + q, m = mt + wr * 128 + (rnd >> 1) * 64 + ((rnd & 1) * 2 + q) * 16 + l15;
          if (m < p.M) {
#pragma unroll
            for (int h = 0; h < 2; ++h) epi.fin2(m, ntl + wc * 64 + h * 32 + 8 * kq, acc[mb][2 * h], acc[mb][2 * h + 1], bq[h][0], bq[h][1], rr[q][h][0], rr[q][h][1]);
          }
        }
      }
      if (!more) break;
      v = vn;
      continue;
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int m = mt + wr * 128 + (mb >> 2) * 64 + (mb & 3) * 16 + l15;
      if (m < p.M) {
#pragma unroll
        for (int h = 0; h < 2; ++h) epi.fin(m, ntl + wc * 64 + h * 32 + 8 * kq, acc[mb][2 * h], acc[mb][2 * h + 1], bq[h][0], bq[h][1]);
      }
    }
    if (!more) break;
    v = vn;
  }
#undef WIS_MMA16
#undef WIS_PROLOGUE
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same structure on a 128 (M) x 256 (N) tile for ONE utterance (M = 1500: the 256 x 256 tile would leave two thirds of the CUs
// without work - QKV 90, FFN1 120 tiles - while this one gives 180 / 240).  8 waves as 2 (M) x 4 (N), wave tile 64 x 64: the kernel
// above without its M-hi half.  A k-tile is THREE half-tiles (g0 = N-lo, g1 = M, g2 = N-hi: 48 KiB) and two phases of 16 MFMAs:
//   phase 0: reads N-lo (4) + M (8); stages g0 and g1 of k-tile t+2        MFMA M x N-lo
//   phase 1: reads N-hi (4);         stages g2 of k-tile t+2, vmcnt(6)     MFMA M x N-hi
// The k-tile is half as long in time as the 256 x 256 one, so the staging runs TWO WHOLE k-tiles ahead through a ring of three
// buffers (144 KiB): a buffer is restaged one k-tile after its last read (g0 / g1 read in phase 0 of t-1, restaged in phase 0 of t;
// g2 read in phase 1 of t-1, restaged in phase 1 of t - two phases apart, the rule of the kernel above without its special cases).
// One counted wait per k-tile: six DMAs of k-tile t+2 may stay in flight, everything older (k-tile t+1) has landed.  Same stagger of
// the two wave rows, same swizzle, same permuted weight rows / 16-byte stores.  Per k-tile the fragment reads (16 ds_read_b128 per
// wave = 1024 LDS cycles per CU) equal the MFMA time (32 MFMAs x 16 cycles x 2 waves per SIMD): the tile is at the LDS-bandwidth
// balance point, which is why the batched encoder keeps the 256 x 256 tile (1536 LDS vs 2048 MFMA cycles).
// Functors with a transposed part (the V images) run both column sets in ONE launch: workgroups >= tiles_a take `pb` and the swapped
// MFMA operand order (a wave-uniform branch; with 64 accumulator registers both MFMA clusters fit without spills).
template <class Epi>
__global__ __launch_bounds__(512) void gemm_8pn_kernel(GemmP pa, GemmP pb, int tiles_a, Epi epi) {
  constexpr int BM_ = 128, BN_ = 256, HALF = 128 * 64, BUF = 3 * HALF, RING = 3 * BUF;
  __shared__ __attribute__((aligned(1024))) f16 smem[RING];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, kq = lane >> 4;
  const bool tr = Epi::HAS_T && (int)blockIdx.x >= tiles_a;
  const GemmP& p = tr ? pb : pa;
  const int nmt = (p.M + BM_ - 1) / BM_, nwg = nmt * (p.N / BN_);
  const int kbeg = p.klen > 0 ? (int)blockIdx.z * p.klen : 0;
  const int nk = (p.klen > 0 ? p.klen : p.K) / BK;
  int m0, n0;
  {
    const int v = (int)blockIdx.x - (tr ? tiles_a : 0);
    const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    m0 = (wg % nmt) * BM_; n0 = (wg / nmt) * BN_;
    if (p.n_span) n0 = (n0 / p.n_span) * p.n_period + p.n_phase + n0 % p.n_span;
  }
  const f16 *sNl0, *sNl1, *sNh0, *sNh1, *sM0, *sM1;
  {
    auto src = [&](int j, const f16** nl, const f16** nh, const f16** ms) {
      const int rl = (wave * 2 + j) * 8 + (lane >> 3), c = (lane & 7) ^ ((rl >> 1) & 7);
      const int i16 = rl & 15, nb16 = (rl >> 4) & 1;
      const int n = n0 + (rl >> 5) * 64 + 8 * (i16 >> 2) + 4 * nb16 + (i16 & 3);      // permuted weight rows: see gemm_8p_kernel
      *nl = p.W + (int64_t)n * p.K + kbeg + c * 8;
      *nh = p.W + (int64_t)(n + 32) * p.K + kbeg + c * 8;
      int m = m0 + rl;
      if (m > p.M - 1) m = p.M - 1;
      *ms = p.A + (int64_t)(m / p.a_rpb) * p.a_bs + (int64_t)(m % p.a_rpb) * p.a_rs + kbeg + c * 8;
    };
    src(0, &sNl0, &sNh0, &sM0);
    src(1, &sNl1, &sNh1, &sM1);
  }
  // the lane's bias values, requested ahead of everything else: a load between the stores of the epilogue would cost a vmcnt(0) wait,
  // which on gfx9 also waits for the stores before it
  // (both forms unconditionally - every index is a valid column: a branch here makes hipcc wait for the loads on the spot)
  f32x4 bq[2][2]; float bt[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bt[nb] = epi.bias1(n0 + wc * 64 + (nb >> 1) * 32 + 8 * (l15 >> 2) + 4 * (nb & 1) + (l15 & 3));
#pragma unroll
  for (int h = 0; h < 2; ++h) { bq[h][0] = epi.bias4(n0 + wc * 64 + h * 32 + 8 * kq); bq[h][1] = epi.bias4(n0 + wc * 64 + h * 32 + 8 * kq + 4); }
  // k-tiles 0 and 1 (the launcher guarantees nk >= 2)
  WIS_STAGE(sNl0, sNl1, 0, 0); WIS_STAGE(sM0, sM1, 1, 0); WIS_STAGE(sNh0, sNh1, 2, 0);
  WIS_STAGE(sNl0, sNl1, 0, BUF); WIS_STAGE(sM0, sM1, 1, BUF); WIS_STAGE(sNh0, sNh1, 2, BUF);
  const int fo0 = l15 * 64 + ((kq ^ (l15 >> 1)) << 3);
  const int oN0 = wc * 32 * 64 + fo0, oM0 = wr * 64 * 64 + fo0;
  f32x4 acc[4][4];
  f16x8 nlo[2][2], nhi[2][2], mf[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#define WIS_MMA16N(NF, NB0) do { \
    __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_setprio(1); \
    if (tr) { \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) \
          _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
            acc[mb][(NB0) + nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf[mb][kb], NF[nb][kb], acc[mb][(NB0) + nb], 0, 0, 0); \
    } else { \
      _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) \
          _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
            acc[mb][(NB0) + nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(NF[nb][kb], mf[mb][kb], acc[mb][(NB0) + nb], 0, 0, 0); \
    } \
    __builtin_amdgcn_s_setprio(0); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
  __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6): k-tile 0 (this wave's share) has landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();      // wave row 1 runs one barrier behind wave row 0
  int cb = 0, sb = 2 * BUF;      // buffer of k-tile t; buffer k-tile t+2 is staged into (= the one k-tile t-1 was read from)
  for (int t = 0; t < nk; ++t) {
    const int rNb0 = oN0 + cb, rNb1 = rNb0 ^ 32, rMb0 = oM0 + cb, rMb1 = rMb0 ^ 32;
    // ---- phase 0
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) nlo[nb][kb] = WIS_FRAG(rNb, 0, nb, kb);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) mf[mb][kb] = WIS_FRAG(rMb, 1, mb, kb);
    if (t + 2 < nk) { WIS_STAGE(sNl0, sNl1, 0, sb); WIS_STAGE(sM0, sM1, 1, sb); }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    WIS_MMA16N(nlo, 0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) nhi[nb][kb] = WIS_FRAG(rNb, 2, nb, kb);
    if (t + 2 < nk) { WIS_STAGE(sNh0, sNh1, 2, sb); __builtin_amdgcn_s_waitcnt(0x0F76); }      // k-tile t+1 has landed; the six DMAs of t+2 fly on
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    WIS_MMA16N(nhi, 2);
    __builtin_amdgcn_s_barrier();
    sb = cb; cb = cb + BUF; if (cb == RING) cb = 0;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();      // pairs with wave row 1's last barrier
  if (tr) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wr * 64 + mb * 16 + 4 * kq;
      if (m < p.M) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) epi.fin_t(m, n0 + wc * 64 + (nb >> 1) * 32 + 8 * (l15 >> 2) + 4 * (nb & 1) + (l15 & 3), acc[mb][nb], bt[nb]);
      }
    }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wr * 64 + mb * 16 + l15;
    if (m < p.M) {
#pragma unroll
      for (int h = 0; h < 2; ++h) epi.fin(m, n0 + wc * 64 + h * 32 + 8 * kq, acc[mb][2 * h], acc[mb][2 * h + 1], bq[h][0], bq[h][1]);
    }
  }
#undef WIS_MMA16N
#undef WIS_FRAG
#undef WIS_STAGE
#undef WIS_DMA
}

// 128-256 ping-pong tiles: every tile has a CU of its own and at least half of the CUs have one (medium, M = 1500: QKV at 144
// tiles 21.8 us against 24.0 for 288 tiles of 128 x 128)
static bool gemm_pp_fits(const GemmP& p, int splits) {
  const int t = cdiv(p.M, 256) * (p.N / 128) * splits;
  return p.N % 128 == 0 && t >= 128 && t <= 256 && p.M >= 256;
}

// Tile choice (WIS_GEMM_TILE=BMxBN overrides with one of the instantiated shapes, tuning only).
static void gemm_pick_tile(const GemmP& p, int* bm, int* bn) {
  static int ebm = -1, ebn = -1;
  if (ebm < 0) { ebm = 0; ebn = 0; if (const char* e = getenv("WIS_GEMM_TILE")) sscanf(e, "%dx%d", &ebm, &ebn); }
  const bool known = (ebm == 64 && ebn == 128) || (ebm == 128 && ebn == 128) || (ebm == 256 && ebn == 256) || (ebm == 256 && ebn == 128);
  if (known && p.N % ebn == 0 && (ebm != 64 || p.M > 64)) { *bm = ebm; *bn = ebn; return; }
  // batched encoder; from 150 tiles on the dense tile beats more, smaller ones (two utterances, QKV: 180 tiles 46.3 us, 720 tiles
  // of 128 x 128 54.5 us)
  if (p.N % 256 == 0 && (p.N / 256) * cdiv(p.M, 256) >= 150) { *bm = 256; *bn = 256; return; }
  if (gemm_pp_fits(p, 1)) { *bm = 256; *bn = 128; return; }                                           // one utterance, wide N
  const bool small = (p.N / 128) * cdiv(p.M, 128) < 200 && p.M > 64;
  *bm = small ? 64 : 128; *bn = 128;
}

// One launch of the 8-phase kernel over `q`'s columns (persistent over tiles: at most one workgroup per CU, a multiple of 8 so that a
// workgroup's tiles stay on its XCD's share)
template <class Epi, bool TR>
static void launch_8p_part(hipStream_t st, const GemmP& q, const Epi& epi) {
  static const bool persist = !(getenv("WIS_GEMM_PERSIST") && atoi(getenv("WIS_GEMM_PERSIST")) == 0);
  static int n_cu = 0;
  if (!n_cu) { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256; n_cu = v & ~7; }
  dim3 g8((q.N / 256) * cdiv(q.M, 256), 1, q.klen > 0 ? q.K / q.klen : 1);
  if (persist && (int)g8.x > n_cu) g8.x = n_cu;
  hipLaunchKernelGGL((gemm_8p_kernel<Epi, TR>), g8, dim3(512), 0, st, WIS_GP_ARGS(q), epi);
}
// Functors with a transposed part (Epi::HAS_T: the V images) get TWO launches - their plain columns, then their transposed columns with
// swapped MFMA operands; Epi::split describes the two column sets.
template <class Epi>
static int launch_gemm_8p(hipStream_t st, const GemmP& p, const Epi& epi) {
  if constexpr (Epi::HAS_T) {
    GemmP a = p, b = p;
    epi.split(p.N, &a, &b);
    if (a.N % 256 || b.N % 256 || a.n_span % 256 || b.n_span % 256) return 1;      // column sets not tile aligned (d = 384): the caller takes the register-staged tile
    launch_8p_part<Epi, false>(st, a, epi);
    launch_8p_part<Epi, true>(st, b, epi);
  } else {
    launch_8p_part<Epi, false>(st, p, epi);
  }
  return WIS_OK;
}

// The 128 x 256 form of the 8-phase kernel: one utterance, wide N.  Returns 1 when the shape is not its case (the caller goes on to
// the register-staged tiles): N a multiple of 256, two or more k-tiles, and between 120 tiles and two rounds of the chip (fewer: the
// 64- / 128-row tiles fill more CUs; more: the 256 x 256 tile's case).  Measured against the ping-pong 256 x 128 tile at M = 1500: FFN1
// 29.2 vs 33.5 us (taken), QKV 33.9 vs 29.8 us (not taken: functors with a transposed part), FFN2's K slices 33.2 vs 32.5 (not taken).
// Both tiles deliver ~35-55 GB/s of operands per CU, the rate the LDS-DMA / L1 path sustains with two k-tiles in flight against a
// 35 % L2 miss rate (profiles/r03_pmc_encoder_sq.md): at one utterance the GEMMs are bound by operand ingest per CU, not by the
// matrix pipe (walking K from a different start per workgroup, so that the sharers of a panel do not miss together, changed nothing).
template <class Epi>
static int launch_gemm_8pn(hipStream_t st, const GemmP& p, const Epi& epi) {
  static const bool use = !(getenv("WIS_GEMM_8PN") && atoi(getenv("WIS_GEMM_8PN")) == 0);
  const int splits = p.klen > 0 ? p.K / p.klen : 1;
  static const bool use_t = getenv("WIS_GEMM_8PN_T") && atoi(getenv("WIS_GEMM_8PN_T")) != 0;      // functors with a transposed part: measured slower than the ping-pong tile (QKV 33.9 vs 29.8 us)
  if (Epi::HAS_T && !use_t) return 1;
  if (!use || p.N % 256 || (p.klen > 0 ? p.klen : p.K) / BK < 2 || p.M < 128) return 1;
  const int tiles = cdiv(p.M, 128) * (p.N / 256) * splits;
  if (tiles < 120 || tiles > 512) return 1;
  GemmP a = p, b = p;
  int tiles_a = cdiv(p.M, 128) * (p.N / 256);
  if constexpr (Epi::HAS_T) {
    epi.split(p.N, &a, &b);
    if (a.N % 256 || b.N % 256 || a.n_span % 256 || b.n_span % 256) return 1;
    tiles_a = cdiv(p.M, 128) * (a.N / 256);
  }
  hipLaunchKernelGGL((gemm_8pn_kernel<Epi>), dim3(cdiv(p.M, 128) * (p.N / 256), 1, splits), dim3(512), 0, st, a, b, tiles_a, epi);
  return WIS_OK;
}

template <class Epi>
static int launch_gemm_t(hipStream_t st, const GemmP& p, const Epi& epi) {
  if (p.N % 128 || p.K % BK || p.M <= 0) { set_error("gemm: M=%d N=%d K=%d unsupported (N%%128, K%%64)", p.M, p.N, p.K); return WIS_E_UNSUPPORTED; }
  int bm, bn; gemm_pick_tile(p, &bm, &bn);
  const dim3 grid((p.N / bn) * cdiv(p.M, bm), 1, p.klen > 0 ? p.K / p.klen : 1);
  // 256 x 256: the 8-phase LDS-DMA kernel (WIS_GEMM_8P=0: the register-staged 2 x 4-wave tile, A/B tuning switch); it needs two k-tiles
  static const bool use_8p = !(getenv("WIS_GEMM_8P") && atoi(getenv("WIS_GEMM_8P")) == 0);
  const int nk_ = (p.klen > 0 ? p.klen : p.K) / BK;
  // (the run-time-flag functor keeps to the register-staged tile: beside the 8-phase loop's 128 accumulators its five operands spill, and a
  // scratch access inside that loop would also move its hand-counted vmcnt waits)
  if constexpr (Epi::USE_8P) if (bm == 256 && bn == 256 && use_8p && nk_ >= 2) {
    const int rc = launch_gemm_8p(st, p, epi);
    if (rc <= 0) return rc;
  }
  if (bm == 256 && bn == 128) {      // one utterance, wide N: the 128 x 256 8-phase tile where it applies
    const int rc = launch_gemm_8pn(st, p, epi);
    if (rc <= 0) return rc;
  }
  if (bm == 256 && bn == 256) hipLaunchKernelGGL((gemm_f16_kernel<Epi, 256, 256, 2, 4>), grid, dim3(512), 0, st, WIS_GP_ARGS(p), epi);
  else if (bm == 256) hipLaunchKernelGGL((gemm_pp_kernel<Epi>), grid, dim3(512), 0, st, WIS_GP_ARGS(p), epi);
  else if (bm == 64) hipLaunchKernelGGL((gemm_f16_kernel<Epi, 64, 128, 2, 2>), grid, dim3(256), 0, st, WIS_GP_ARGS(p), epi);
  else hipLaunchKernelGGL((gemm_f16_kernel<Epi, 128, 128, 2, 2>), grid, dim3(256), 0, st, WIS_GP_ARGS(p), epi);
  return WIS_OK;
}
// f32x4 sums of the epilogues, component by component and pinned in VGPRs: `a + b` on the vector type compiles to v_pk_add_f32, which the
// library keeps away from MFMA kernels that can run three waves per SIMD (tools/isa_lint.py)
__device__ __forceinline__ float add_s(float a, float b) { return a + b; }
__device__ __forceinline__ float add_p(float a, float b) { float t = a + b; asm("" : "+v"(t)); return t; }
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) { return f32x4{add_p(a[0], b[0]), add_s(a[1], b[1]), add_p(a[2], b[2]), add_s(a[3], b[3])}; }
__device__ __forceinline__ f32x4 ld4(const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); return f32x4{t.x, t.y, t.z, t.w}; }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4h(f16* p, f32x4 v) { f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}; *reinterpret_cast<f16x4*>(p) = o; }
__device__ __forceinline__ void st8h(f16* p, f32x4 a, f32x4 b) {      // 8 consecutive f16 = one 16-byte store
  f16x8 o = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)b[0], (f16)b[1], (f16)b[2], (f16)b[3]};
  *reinterpret_cast<f16x8*>(p) = o;
}
__device__ __forceinline__ f32x4 gelu4(f32x4 v) { return f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])}; }

// generic runtime-flag epilogue (wis_op_gemm, FFN, out-proj)
struct EpiGeneric {
  static constexpr bool USE_8P = false, HAS_T = false, HAS_RES = false;
  __device__ void store_t(int, int, f32x4) const {}
  const float* bias; const float* resid; void* C; int N; int flags;  // 1 gelu, 2 resid, 4 out f32
  __device__ void operator()(int m, int n, f32x4 v) const {
    if (bias) v = add4(v, ld4(bias + n));
    if (flags & 1) v = gelu4(v);
    const size_t o = (size_t)m * N + n;
    if (flags & 2) v = add4(v, ld4(resid + o));
    if (flags & 4) st4(reinterpret_cast<float*>(C) + o, v); else st4h(reinterpret_cast<f16*>(C) + o, v);
  }
  // register-staged tiles (epilogue_32): bias and residual quads arrive in registers
  __device__ bool has_res() const { return (flags & 2) != 0; }
  __device__ f32x4 res4(int m, int n) const { return ld4(resid + (size_t)m * N + n); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 b) const {
    v = add4(v, b);
    if (flags & 1) v = gelu4(v);
    const size_t o = (size_t)m * N + n;
    if (flags & 4) st4(reinterpret_cast<float*>(C) + o, v); else st4h(reinterpret_cast<f16*>(C) + o, v);
  }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4 r) const {
    v = add4(v, b);
    if (flags & 1) v = gelu4(v);
    v = add4(v, r);
    const size_t o = (size_t)m * N + n;
    if (flags & 4) st4(reinterpret_cast<float*>(C) + o, v); else st4h(reinterpret_cast<f16*>(C) + o, v);
  }
  // the same with the bias of the 8 features already in registers (bias4 / bias1: requested before the k-loop by the 128 x 256 kernel -
  // a load inside the store sequence costs an s_waitcnt vmcnt(0), which on gfx9 also waits for every store issued before it)
  __device__ f32x4 bias4(int n) const { return bias ? ld4(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ float bias1(int) const { return 0.f; }
  __device__ void fin_t(int, int, f32x4, float) const {}
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb) const {
    a = add4(a, ba); b = add4(b, bb);
    if (flags & 1) { a = gelu4(a); b = gelu4(b); }
    const size_t o = (size_t)m * N + n;
    if (flags & 2) { a = add4(a, ld4(resid + o)); b = add4(b, ld4(resid + o + 4)); }
    if (flags & 4) { st4(reinterpret_cast<float*>(C) + o, a); st4(reinterpret_cast<float*>(C) + o + 4, b); } else st8h(reinterpret_cast<f16*>(C) + o, a, b);
  }
  // 8 consecutive features of one row (the 8-phase kernel: n % 8 == 0): f16 outputs leave as ONE 16-byte store
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const {
    if (bias) { a = add4(a, ld4(bias + n)); b = add4(b, ld4(bias + n + 4)); }
    if (flags & 1) { a = gelu4(a); b = gelu4(b); }
    const size_t o = (size_t)m * N + n;
    if (flags & 2) { a = add4(a, ld4(resid + o)); b = add4(b, ld4(resid + o + 4)); }
    if (flags & 4) { st4(reinterpret_cast<float*>(C) + o, a); st4(reinterpret_cast<float*>(C) + o + 4, b); } else st8h(reinterpret_cast<f16*>(C) + o, a, b);
  }
};
// bias + GELU -> f16 rows (the encoder's FFN1 on every tile shape): the generic functor without its run-time flags, residual and fp32
// output - three pointers and two words fewer to keep alive across the k-loop of the 256-register kernels
struct EpiGelu {
  static constexpr bool USE_8P = true, HAS_T = false, HAS_RES = false;
  __device__ void store_t(int, int, f32x4) const {}
  __device__ void fin_t(int, int, f32x4, float) const {}
  const float* bias; f16* C; int N;
  __device__ void operator()(int m, int n, f32x4 v) const { st4h(C + (size_t)m * N + n, gelu4(add4(v, ld4(bias + n)))); }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const { fin(m, n, a, b, ld4(bias + n), ld4(bias + n + 4)); }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ float bias1(int) const { return 0.f; }
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb) const { st8h(C + (size_t)m * N + n, gelu4(add4(a, ba)), gelu4(add4(b, bb))); }
  __device__ bool has_res() const { return false; }
  __device__ f32x4 res4(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 b) const { st4h(C + (size_t)m * N + n, gelu4(add4(v, b))); }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4) const { fin4(m, n, v, b); }
};
// bias + fp32 residual -> fp32 (out-projection, FFN2) for the 8-phase kernel: HAS_RES = the kernel requests the residual values of TWO
// row blocks (res8) before it stores any of them (fin2) - four load rounds per tile instead of one load + wait per store
struct EpiResid {
  static constexpr bool USE_8P = true, HAS_T = false, HAS_RES = true;
  const float* bias; const float* resid; float* C; int N;
  __device__ void store_t(int, int, f32x4) const {}
  __device__ void fin_t(int, int, f32x4, float) const {}
  __device__ float bias1(int) const { return 0.f; }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ void res8(int m, int n, f32x4& ra, f32x4& rb) const { const size_t o = (size_t)m * N + n; ra = ld4(resid + o); rb = ld4(resid + o + 4); }
  __device__ void fin2(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb, f32x4 ra, f32x4 rb) const {
    const size_t o = (size_t)m * N + n;
    st4(C + o, add4(add4(a, ba), ra)); st4(C + o + 4, add4(add4(b, bb), rb));
  }
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb) const { f32x4 ra, rb; res8(m, n, ra, rb); fin2(m, n, a, b, ba, bb, ra, rb); }
  __device__ bool has_res() const { return true; }
  __device__ f32x4 res4(int m, int n) const { return ld4(resid + (size_t)m * N + n); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 b) const { st4(C + (size_t)m * N + n, add4(add4(v, b), res4(m, n))); }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4 r) const { st4(C + (size_t)m * N + n, add4(add4(v, b), r)); }
  __device__ void operator()(int m, int n, f32x4 v) const { const size_t o = (size_t)m * N + n; st4(C + o, add4(add4(v, ld4(bias + n)), ld4(resid + o))); }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const { fin(m, n, a, b, bias4(n), bias4(n + 4)); }
};
// conv1: GELU(acc + b) -> f16 time-major padded image [B][T+2][N], row t+1
struct EpiConv1 {
  static constexpr bool USE_8P = true, HAS_T = false, HAS_RES = false;
  __device__ void store_t(int, int, f32x4) const {}
  const float* bias; f16* C; int N; int T;
  __device__ void operator()(int m, int n, f32x4 v) const {
    v = gelu4(add4(v, ld4(bias + n)));
    const int b = m / T, t = m - b * T;
    st4h(C + ((size_t)b * (T + 2) + t + 1) * N + n, v);
  }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 c) const {
    a = gelu4(add4(a, ld4(bias + n))); c = gelu4(add4(c, ld4(bias + n + 4)));
    const int b = m / T, t = m - b * T;
    st8h(C + ((size_t)b * (T + 2) + t + 1) * N + n, a, c);
  }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ float bias1(int) const { return 0.f; }
  __device__ void fin_t(int, int, f32x4, float) const {}
  __device__ void fin(int m, int n, f32x4 a, f32x4 c, f32x4 ba, f32x4 bb) const {
    a = gelu4(add4(a, ba)); c = gelu4(add4(c, bb));
    const int b = m / T, t = m - b * T;
    st8h(C + ((size_t)b * (T + 2) + t + 1) * N + n, a, c);
  }
  __device__ bool has_res() const { return false; }
  __device__ f32x4 res4(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4) const { fin4(m, n, v, b); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 bq) const {
    v = gelu4(add4(v, bq));
    const int b = m / T, t = m - b * T;
    st4h(C + ((size_t)b * (T + 2) + t + 1) * N + n, v);
  }
};
// conv2: GELU(acc + b) + pos[t] -> fp32 residual stream [B*T][N]
struct EpiConv2 {
  static constexpr bool USE_8P = true, HAS_T = false, HAS_RES = false;
  __device__ void store_t(int, int, f32x4) const {}
  const float* bias; const float* pos; float* X; int N; int T;
  __device__ void operator()(int m, int n, f32x4 v) const {
    v = gelu4(add4(v, ld4(bias + n)));
    const int t = m % T;
    v = add4(v, ld4(pos + (size_t)t * N + n));
    st4(X + (size_t)m * N + n, v);
  }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const { (*this)(m, n, a); (*this)(m, n + 4, b); }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ float bias1(int) const { return 0.f; }
  __device__ void fin_t(int, int, f32x4, float) const {}
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb) const {
    const int t = m % T;
    a = add4(gelu4(add4(a, ba)), ld4(pos + (size_t)t * N + n)); b = add4(gelu4(add4(b, bb)), ld4(pos + (size_t)t * N + n + 4));
    st4(X + (size_t)m * N + n, a); st4(X + (size_t)m * N + n + 4, b);
  }
  // (the positional rows play the residual's part: one quad per (row, column quad))
  __device__ bool has_res() const { return true; }
  __device__ f32x4 res4(int m, int n) const { return ld4(pos + (size_t)(m % T) * N + n); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 b) const { fin4r(m, n, v, b, res4(m, n)); }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4 r) const { st4(X + (size_t)m * N + n, add4(gelu4(add4(v, b)), r)); }
};
// fused QKV: [Q*s | K] -> f16 [M][2d]; V -> V^T f16 [B][H][64][Tpad]
struct EpiQKV {
  // V tiles of the 8-phase kernel run their MFMAs with the operands swapped (D transposed): a lane then holds 4 CONSECUTIVE rows (keys) of
  // one feature, i.e. 8 contiguous bytes of the transposed V image instead of four scattered 2-byte stores
  static constexpr bool USE_8P = true, HAS_T = true, HAS_RES = false;
  void split(int N, GemmP* plain, GemmP* tr) const {      // [Q | K] = columns [0, 2d) plain, V = [2d, 3d) transposed
    plain->N = 2 * d; plain->n_span = 2 * d; plain->n_period = 0; plain->n_phase = 0;
    tr->N = N - 2 * d; tr->n_span = N - 2 * d; tr->n_period = 0; tr->n_phase = 2 * d;
  }
  __device__ void store_t(int m, int n, f32x4 v) const {      // rows m .. m + 3 (m % 4 == 0: one utterance, one 4-group of the key swizzle) of feature n
    const float bv = bias[n];
    const int nn = n - 2 * d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
    st4h(vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + tp, add4(v, f32x4{bv, bv, bv, bv}));
  }
  const float* bias; f16* qk; f16* vt; int d; int T; int Tpad; int H;
  __device__ void operator()(int m, int n, f32x4 v) const {
    v = add4(v, ld4(bias + n));
    if (n < 2 * d) { st4h(qk + (size_t)m * 2 * d + n, v); return; }
    const int nn = n - 2 * d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    // keys are stored with bits 2 and 3 of their index swapped inside every group of 16: the 8 keys a lane of the attention
    // kernel feeds to one P.V MFMA step (16 s + 8 (j >> 2) + 4 hi + (j & 3), the order its score accumulators come in) are then
    // contiguous - one conflict-free ds_read_b128 instead of two bank-conflicting 8-byte reads
    const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
    f16* o = vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + tp;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(size_t)j * Tpad] = (f16)v[j];
  }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const {
    // only reached for the Q | K tiles (the 8-phase kernel sends the V tiles through store_t)
    a = add4(a, ld4(bias + n)); b = add4(b, ld4(bias + n + 4)); st8h(qk + (size_t)m * 2 * d + n, a, b);
  }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ float bias1(int n) const { return bias[n]; }
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4 ba, f32x4 bb) const { st8h(qk + (size_t)m * 2 * d + n, add4(a, ba), add4(b, bb)); }
  __device__ bool has_res() const { return false; }
  __device__ f32x4 res4(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4) const { fin4(m, n, v, b); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4 bq) const {
    v = add4(v, bq);
    if (n < 2 * d) { st4h(qk + (size_t)m * 2 * d + n, v); return; }
    const int nn = n - 2 * d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
    f16* o = vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + tp;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(size_t)j * Tpad] = (f16)v[j];
  }
  __device__ void fin_t(int m, int n, f32x4 v, float bv) const {
    const int nn = n - 2 * d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
    st4h(vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + tp, add4(v, f32x4{bv, bv, bv, bv}));
  }
};
// cross-attention K/V projection of the encoder memory for ONE decoder layer:
//   K -> Kx f16 [B][H][8][T][8]    (16-byte dh-groups contiguous along T = the MFMA A-fragment rows of the decode kernel)
//   V -> V^T f16 [B][H][64][Tpad]  (keys contiguous, zero padded to Tpad)
// The projections of ALL decoder layers run as ONE GEMM over the concatenated weights [L * 2d][d] (7680 tiles for large-v2 at
// B = 1: thirty full rounds of the chip, instead of 32 launches of 240 tiles): n -> (layer n / 2d, column n % 2d); a 128- or
// 256-wide tile never straddles layers because 2d is a multiple of 256 for every Whisper size but tiny (768: multiple of 128
// and of 256).
struct EpiCrossKV {
  static constexpr bool USE_8P = true, HAS_T = true, HAS_RES = false;
  void split(int N, GemmP* plain, GemmP* tr) const {      // per layer: K = columns [0, d) plain, V = [d, 2d) transposed; N = layers x 2d
    plain->N = N / 2; plain->n_span = d; plain->n_period = 2 * d; plain->n_phase = 0;
    tr->N = N / 2; tr->n_span = d; tr->n_period = 2 * d; tr->n_phase = d;
  }
  __device__ void store_t(int m, int n_all, f32x4 v) const {
    const float bv = bias[n_all];
    const int l = n_all / (2 * d), nn = n_all - l * 2 * d - d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    st4h(vt + l * vt_lstride + ((size_t)(b * H + h) * 64 + dh) * Tpad + t, add4(v, f32x4{bv, bv, bv, bv}));
  }
  const float* bias; f16* kx; f16* vt; int d; int T; int Tpad; int H; int64_t kx_lstride, vt_lstride;
  __device__ void operator()(int m, int n_all, f32x4 v) const {
    v = add4(v, ld4(bias + n_all));
    const int l = n_all / (2 * d), n = n_all - l * 2 * d;
    f16* kx = this->kx + l * kx_lstride; f16* vt = this->vt + l * vt_lstride;
    const int b = m / T, t = m - b * T;
    if (n < d) {
      const int h = n >> 6, dh = n & 63, g = dh >> 3, j = dh & 7;
      st4h(kx + ((((size_t)(b * H + h) * 8 + g) * T + t) * 8 + j), v);
    } else {
      const int nn = n - d, h = nn >> 6, dh = nn & 63;
      f16* o = vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + t;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[(size_t)j * Tpad] = (f16)v[j];
    }
  }
  __device__ void operator()(int m, int n_all, f32x4 a, f32x4 c) const {
    // only reached for the K tiles (the V tiles go through store_t): the 8 values are one (t, dh-group) cell of the K image - a single 16-byte store
    const int l = n_all / (2 * d), n = n_all - l * 2 * d;
    a = add4(a, ld4(bias + n_all)); c = add4(c, ld4(bias + n_all + 4));
    const int b = m / T, t = m - b * T, h = n >> 6, g = (n & 63) >> 3;
    st8h(kx + l * kx_lstride + ((((size_t)(b * H + h) * 8 + g) * T + t) * 8), a, c);
  }
  __device__ f32x4 bias4(int n) const { return ld4(bias + n); }
  __device__ float bias1(int n) const { return bias[n]; }
  __device__ void fin(int m, int n_all, f32x4 a, f32x4 c, f32x4 ba, f32x4 bb) const {
    const int l = n_all / (2 * d), n = n_all - l * 2 * d;
    const int b = m / T, t = m - b * T, h = n >> 6, g = (n & 63) >> 3;
    st8h(kx + l * kx_lstride + ((((size_t)(b * H + h) * 8 + g) * T + t) * 8), add4(a, ba), add4(c, bb));
  }
  __device__ void fin_t(int m, int n_all, f32x4 v, float bv) const {
    const int l = n_all / (2 * d), nn = n_all - l * 2 * d - d, h = nn >> 6, dh = nn & 63, b = m / T, t = m - b * T;
    st4h(vt + l * vt_lstride + ((size_t)(b * H + h) * 64 + dh) * Tpad + t, add4(v, f32x4{bv, bv, bv, bv}));
  }
  __device__ bool has_res() const { return false; }
  __device__ f32x4 res4(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4) const { fin4(m, n, v, b); }
  __device__ void fin4(int m, int n_all, f32x4 v, f32x4 bq) const {
    v = add4(v, bq);
    const int l = n_all / (2 * d), n = n_all - l * 2 * d;
    f16* kx = this->kx + l * kx_lstride; f16* vt = this->vt + l * vt_lstride;
    const int b = m / T, t = m - b * T;
    if (n < d) {
      const int h = n >> 6, dh = n & 63, g = dh >> 3, j = dh & 7;
      st4h(kx + ((((size_t)(b * H + h) * 8 + g) * T + t) * 8 + j), v);
    } else {
      const int nn = n - d, h = nn >> 6, dh = nn & 63;
      f16* o = vt + ((size_t)(b * H + h) * 64 + dh) * Tpad + t;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[(size_t)j * Tpad] = (f16)v[j];
    }
  }
};

// split-K partial tile: fp32 [blockIdx.z][M][N]
struct EpiPartial {
  static constexpr bool USE_8P = true, HAS_T = false, HAS_RES = false;
  __device__ void store_t(int, int, f32x4) const {}
  float* C; int N; int64_t zstride;
  __device__ void operator()(int m, int n, f32x4 v) const { st4(C + (int64_t)blockIdx.z * zstride + (size_t)m * N + n, v); }
  __device__ void operator()(int m, int n, f32x4 a, f32x4 b) const { (*this)(m, n, a); (*this)(m, n + 4, b); }
  __device__ f32x4 bias4(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ float bias1(int) const { return 0.f; }
  __device__ void fin_t(int, int, f32x4, float) const {}
  __device__ void fin(int m, int n, f32x4 a, f32x4 b, f32x4, f32x4) const { (*this)(m, n, a); (*this)(m, n + 4, b); }
  __device__ bool has_res() const { return false; }
  __device__ f32x4 res4(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ void fin4r(int m, int n, f32x4 v, f32x4 b, f32x4) const { fin4(m, n, v, b); }
  __device__ void fin4(int m, int n, f32x4 v, f32x4) const { (*this)(m, n, v); }
};
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t zstride, const float* __restrict__ bias,
                                     const float* resid, float* X, int64_t n4, int N) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4*>(part)[i];
    for (int z = 1; z < splits; ++z) { const float4 t = reinterpret_cast<const float4*>(part + z * zstride)[i]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    const int n = (int)((i * 4) % N);
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    const float4 r = reinterpret_cast<const float4*>(resid)[i];
    reinterpret_cast<float4*>(X)[i] = make_float4(a.x + b.x + r.x, a.y + b.y + r.y, a.z + b.z + r.z, a.w + b.w + r.w);
  }
}
// The same reduction with the LayerNorm that consumes the new residual rows fused in: one wave per row, the row stays in registers
// between "x = sum of partials + bias + residual" (written back in fp32) and "y = LN(x) gamma + beta" (f16, the A operand of the
// next projection) - one launch and one 7.7 MB read of x less per use.  Same summation order as splitk_reduce_kernel and the same
// two-pass statistics as layernorm_kernel.  Loads are unconditional (clamped column index, masked result): a guarded load per
// float4 compiles into a branch and a wait each.
template <int SPLITS>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ part, int64_t zstride, const float* __restrict__ bias,
                                                               const float* resid, float* X, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, f16* __restrict__ Y, int M, int d) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int n4 = d >> 2;
  const int64_t z4 = zstride >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(part) + (size_t)row * n4;
  const float4* r4 = reinterpret_cast<const float4*>(resid) + (size_t)row * n4;
  const float4* b4 = reinterpret_cast<const float4*>(bias);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* e4 = reinterpret_cast<const float4*>(beta);
  float4 v[8], g[8], be[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i, ic = idx < n4 ? idx : n4 - 1;
    float4 a = p4[ic];
#pragma unroll
    for (int z = 1; z < SPLITS; ++z) { const float4 t = p4[z * z4 + ic]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    const float4 b = b4[ic], r = r4[ic];
    g[i] = g4[ic]; be[i] = e4[ic];
    a = make_float4(a.x + b.x + r.x, a.y + b.y + r.y, a.z + b.z + r.z, a.w + b.w + r.w);
    if (idx >= n4) a = make_float4(0.f, 0.f, 0.f, 0.f);
    v[i] = a;
    s += (a.x + a.y) + (a.z + a.w);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { WIS_PIN4(g[i]); WIS_PIN4(be[i]); }      // as in layernorm_kernel
  float4* x4 = reinterpret_cast<float4*>(X) + (size_t)row * n4;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int idx = lane + 64 * i; if (idx < n4) x4[idx] = v[i]; }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    const float t = (a * a + b * b) + (c * c + e * e);
    q += idx < n4 ? t : 0.f;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
  f16x4* y4 = reinterpret_cast<f16x4*>(Y + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i;
    f16x4 o;
    o[0] = (f16)((v[i].x - mean) * rstd * g[i].x + be[i].x);
    o[1] = (f16)((v[i].y - mean) * rstd * g[i].y + be[i].y);
    o[2] = (f16)((v[i].z - mean) * rstd * g[i].z + be[i].z);
    o[3] = (f16)((v[i].w - mean) * rstd * g[i].w + be[i].w);
    if (idx < n4) y4[idx] = o;
  }
}
// ln_gamma / ln_beta / Y given: the fused reduction + LayerNorm (N <= 2048); otherwise the plain reduction.
int launch_gemm_splitk_resid(hipStream_t st, const GemmP& p0, int splits, float* scratch, const float* bias, const float* resid, float* X,
                             const float* ln_gamma, const float* ln_beta, f16* Y) {
  GemmP p = p0;
  if (splits < 2 || p.K % (splits * BK) || p.N % 128) { set_error("splitk: K=%d splits=%d unsupported", p.K, splits); return WIS_E_UNSUPPORTED; }
  p.klen = p.K / splits;
  const int64_t zs = (int64_t)p.M * p.N;
  EpiPartial e{scratch, p.N, zs};
  // (the 128 x 256 8-phase tile measured no better here: 33.2 vs 32.5 us for the four K slices of FFN2)
  if (gemm_pp_fits(p, splits)) hipLaunchKernelGGL((gemm_pp_kernel<EpiPartial>), dim3((p.N / 128) * cdiv(p.M, 256), 1, splits), dim3(512), 0, st, WIS_GP_ARGS(p), e);
  else hipLaunchKernelGGL((gemm_f16_kernel<EpiPartial, 128, 128, 2, 2>), dim3((p.N / 128) * cdiv(p.M, 128), 1, splits), dim3(256), 0, st, WIS_GP_ARGS(p), e);
  if (Y) {
    if (!ln_gamma || !ln_beta || p.N > 2048 || (splits != 2 && splits != 4)) { set_error("splitk: fused LayerNorm needs gamma, beta, N <= 2048 and 2 or 4 splits"); return WIS_E_ARG; }
    if (splits == 2) hipLaunchKernelGGL((splitk_reduce_ln_kernel<2>), dim3(cdiv(p.M, 4)), dim3(256), 0, st, scratch, zs, bias, resid, X, ln_gamma, ln_beta, Y, p.M, p.N);
    else hipLaunchKernelGGL((splitk_reduce_ln_kernel<4>), dim3(cdiv(p.M, 4)), dim3(256), 0, st, scratch, zs, bias, resid, X, ln_gamma, ln_beta, Y, p.M, p.N);
    return WIS_OK;
  }
  const int64_t n4 = zs / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, scratch, splits, zs, bias, resid, X, n4, p.N);
  return WIS_OK;
}

int launch_gemm_generic(hipStream_t st, const GemmP& p, const float* bias, const float* resid, void* C, int flags) {
  if (flags == (2 | 4) && bias && resid && p.N % 256 == 0 && p.K % BK == 0 && p.klen == 0 && p.K / BK >= 2) {      // fp32 residual epilogue on the 8-phase tile: its own functor
    static const bool use_8p = !(getenv("WIS_GEMM_8P") && atoi(getenv("WIS_GEMM_8P")) == 0);
    int bm, bn; gemm_pick_tile(p, &bm, &bn);
    static const bool use_res = !(getenv("WIS_GEMM_RESID") && atoi(getenv("WIS_GEMM_RESID")) == 0);      // 0: the generic functor (A/B switch)
    if (use_res && use_8p && bm == 256 && bn == 256) { launch_8p_part<EpiResid, false>(st, p, EpiResid{bias, resid, reinterpret_cast<float*>(C), p.N}); return WIS_OK; }
  }
  if (flags == 1 && bias) return launch_gemm_t(st, p, EpiGelu{bias, reinterpret_cast<f16*>(C), p.N});      // FFN1
  EpiGeneric e{bias, resid, C, p.N, flags};
  return launch_gemm_t(st, p, e);
}
int launch_gemm_conv1(hipStream_t st, const GemmP& p, const float* bias, f16* C, int T) {
  EpiConv1 e{bias, C, p.N, T};
  return launch_gemm_t(st, p, e);
}
int launch_gemm_conv2(hipStream_t st, const GemmP& p, const float* bias, const float* pos, float* X, int T) {
  EpiConv2 e{bias, pos, X, p.N, T};
  return launch_gemm_t(st, p, e);
}
int launch_gemm_qkv(hipStream_t st, const GemmP& p, const float* bias, f16* qk, f16* vt, int d, int T, int Tpad, int H) {
  EpiQKV e{bias, qk, vt, d, T, Tpad, H};
  return launch_gemm_t(st, p, e);
}
int launch_gemm_crosskv(hipStream_t st, const GemmP& p, const float* bias, f16* kx, f16* vt, int d, int T, int Tpad, int H, int64_t kx_lstride, int64_t vt_lstride) {
  if (p.N % (2 * d)) { set_error("crosskv: N=%d is not a whole number of layers", p.N); return WIS_E_ARG; }
  EpiCrossKV e{bias, kx, vt, d, T, Tpad, H, kx_lstride, vt_lstride};
  return launch_gemm_t(st, p, e);
}

// =======================================================================================
// Encoder self-attention (non-causal, dh = 64).  grid (ceil(T/128), H, B), block 256.
// The loop is bound by the softmax arithmetic, not by the matrix pipe: per 64-key tile a wave issues 16 MFMAs (512 cycles) and, on
// the vector ALU, 32 maxima, 32 fma + v_exp_f32 (a quarter-rate instruction: 16 cycles per wave) and 32 sums and conversions -
// about 1100 cycles - in one dependent chain Q.K^T -> softmax -> P.V.  A second wave on the SIMD fills one pipe while the first
// uses the other (eight utterances: 19.5 us per utterance and layer against 33 us for one), so:
// SPLIT (one or two utterances, <= 2 workgroups per CU otherwise): the keys of a (query tile, head) are divided between TWO
// workgroups (grid.x doubles: 480 workgroups for one utterance = two per CU, free-running against each other); each ends with its
// own online-softmax state (m, l, O) and publishes it with write-through stores; the one that draws the second ticket merges the
// other's state into its registers and writes the rows (dec_cross_attn_kernel's hand-off: relaxed ticket, one acquire, plain
// loads).  The merge is symmetric in the two states, so the result does not depend on which workgroup arrives last.
constexpr int AKT = 64, ASTR = 72;  // 64-key tiles, LDS row pitch 72 f16 = 144 B
constexpr int ENC_PART_FLOATS = 34 * 256;     // per workgroup: m, l, O[32] per thread

template <bool SPLIT>
__global__ __launch_bounds__(256) void enc_attn_kernel(const f16* __restrict__ qk, const f16* __restrict__ vt,
                                                       f16* __restrict__ out, int T, int Tpad, int H, int d, float* part, unsigned* counters) {
  __shared__ __attribute__((aligned(16))) f16 sK[2][AKT * ASTR];
  __shared__ __attribute__((aligned(16))) f16 sV[2][64 * ASTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qt = SPLIT ? (int)blockIdx.x >> 1 : (int)blockIdx.x, half = SPLIT ? (int)blockIdx.x & 1 : 0;
  const int q_row = qt * 128 + wave * 32 + l31;
  const int q_c = q_row < T ? q_row : T - 1;
  const int ld = 2 * d;

  f16x8 qf[4];
  {
    const f16* qp = qk + (size_t)(b * T + q_c) * ld + h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const f16x8*>(qp + kk * 16 + hi * 8);
  }
  // tile loaders: 64 rows x 8 chunks of 16 B for K and for V^T
  const f16* kbase = qk + (size_t)b * T * ld + d + h * 64;
  const f16* vbase = vt + (size_t)(b * H + h) * 64 * Tpad;
  const int lrow0 = tid >> 3, lrow1 = (tid + 256) >> 3, lch = (tid & 7) * 8;
  const f16* vp0 = vbase + (size_t)lrow0 * Tpad + lch;
  const f16* vp1 = vbase + (size_t)lrow1 * Tpad + lch;
  const int so0 = lrow0 * ASTR + lch, so1 = lrow1 * ASTR + lch;
  // The next tile's K / V pieces are fetched by inline-asm loads that hipcc does not count: with compiler-visible loads its
  // waitcnt pass put `s_waitcnt vmcnt(3..0)` between the Q.K^T MFMAs at the TOP of the iteration (the ds_read destinations there
  // alias the address temporaries of the in-flight loads), so every tile waited out the L2 latency of its own prefetch before
  // computing.  The wait is stated by hand where the data is consumed - in front of the LDS stores at the bottom, behind all 16
  // MFMAs and the softmax (cdna_hip_programming.md 5.7, form (iii): "=v" loads, "+v" operands on the wait).
  u32x4 rk0, rk1, rv0, rv1;
#define WIS_GLOAD(kt)                                                                      \
  {                                                                                        \
    int key0 = (kt) * AKT + lrow0; if (key0 > T - 1) key0 = T - 1;                          \
    int key1 = (kt) * AKT + lrow1; if (key1 > T - 1) key1 = T - 1;                          \
    const f16* a0_ = kbase + (size_t)key0 * ld + lch; const f16* a1_ = kbase + (size_t)key1 * ld + lch;   \
    const f16* a2_ = vp0 + (kt) * AKT; const f16* a3_ = vp1 + (kt) * AKT;                   \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"  \
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"      \
                 : "=&v"(rk0), "=&v"(rk1), "=&v"(rv0), "=&v"(rv1) : "v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_) : "memory"); \
  }
#define WIS_SSTORE(buf)                                                                    \
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(rk0), "+v"(rk1), "+v"(rv0), "+v"(rv1) :: "memory"); \
  *reinterpret_cast<u32x4*>(&sK[buf][so0]) = rk0;                                          \
  *reinterpret_cast<u32x4*>(&sK[buf][so1]) = rk1;                                          \
  *reinterpret_cast<u32x4*>(&sV[buf][so0]) = rv0;                                          \
  *reinterpret_cast<u32x4*>(&sV[buf][so1]) = rv1;

  f32x16 o[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[a][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nt_all = cdiv(T, AKT), nt_half = (nt_all + 1) >> 1;
  const int t_beg = SPLIT ? half * nt_half : 0;
  const int ntiles = SPLIT ? (t_beg + nt_half < nt_all ? t_beg + nt_half : nt_all) : nt_all;      // one past this workgroup's last tile
  WIS_GLOAD(t_beg) WIS_SSTORE(0)
  // the Q fragments are complete here, and hipcc must KNOW it: its waitcnt pass otherwise carries "q loads may be outstanding"
  // into the loop and, the vmcnt counter being in-order, expresses that as vmcnt(3..0) in front of the Q.K^T MFMAs - which in the
  // loop means "wait for the four prefetch loads just issued".  A builtin wait is visible to the pass.
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) expcnt(7) lgkmcnt(15)
  __syncthreads();
  for (int kt = t_beg; kt < ntiles; ++kt) {
    const int cur = (kt - t_beg) & 1;
    if (kt + 1 < ntiles) WIS_GLOAD(kt + 1)
    // S^T[key][q] = K . Q^T  (A = K rows, B = Q rows)
    f32x16 st[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[cur][(t2 * 32 + l31) * ASTR + kk * 16 + hi * 8]);
        st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], st[t2], 0, 0, 0);
      }
    }
    // softmax bookkeeping in the log2 domain (one fma + one v_exp per score); only the last key tile can hold keys >= T;
    // the accumulator rescale is skipped while no lane of the wave saw a new maximum (the common case after a few tiles)
    constexpr float LOG2E = 1.4426950408889634f;
    float mx = -INFINITY;
    if (kt == nt_all - 1) {
      const int key_base = kt * AKT + 4 * hi;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + t2 * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= T) st[t2][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * LOG2E;             // m_run is kept pre-multiplied by log2(e)
    const float m_new = fmaxf(m_run, mx);
    const bool grew = m_new > m_run;
    float rs = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float pv = __builtin_amdgcn_exp2f(fmaf(st[t2][r], LOG2E, -m_new)); st[t2][r] = pv; rs += pv; }   // raw v_exp_f32: arguments are <= 0, a flushed denormal is an exact 0 weight (libm's exp2f wraps it in 5 range-fixup instructions)
    if (__any(grew)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[a][r] *= alpha;
    }
    l_run += rs; m_run = m_new;
    // P fragments straight from the accumulators: k-step s of 16 keys <-> S-tile s>>1, regs 8(s&1)..+7;
    // slot j of half `hi` is key 16s + 8(j>>2) + 4hi + (j&3)  (V^T is stored in exactly that order)
    f16x8 pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[s][j] = (f16)st[s >> 1][8 * (s & 1) + j];
    // O^T[dh][q] += V^T . P^T   (A = V^T rows (dh), B = P rows (q))
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&sV[cur][(dt * 32 + l31) * ASTR + 16 * s + 8 * hi]);   // (bit-swapped key order, see EpiQKV)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], o[dt], 0, 0, 0);
      }
    if (kt + 1 < ntiles) { WIS_SSTORE(cur ^ 1) }
    __syncthreads();
  }
#undef WIS_GLOAD
#undef WIS_SSTORE
  if (SPLIT) {
    // publish (write-through, relaxed agent scope), drain, ticket; the second arriver merges the first one's state into its own
    __shared__ int s_last;
    const int nqt = (int)gridDim.x >> 1;
    const size_t pair = (size_t)(b * H + h) * nqt + qt;
    float* mine = part + (pair * 2 + half) * ENC_PART_FLOATS + tid;
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine), __float_as_uint(m_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 256), __float_as_uint(l_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_store(reinterpret_cast<unsigned*>(mine + (2 + a * 16 + r) * 256), __float_as_uint(o[a][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(counters + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == 1u;
      if (last) {
        __hip_atomic_store(counters + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const float* other = part + (pair * 2 + (half ^ 1)) * ENC_PART_FLOATS + tid;
    const float m1 = other[0], l1 = other[256];
    float o1[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) o1[a][r] = other[(2 + a * 16 + r) * 256];
    const float mm = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    // (a0 x + a1 y is evaluated as the sum of two products, never as an fma chain: the same bits whichever side is `mine`)
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][r] = o[a][r] * a0 + o1[a][r] * a1;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_row < T) {
    f16* op = out + (size_t)(b * T + q_row) * d + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dh = dt * 32 + 8 * r4 + 4 * hi;
        f32x4 v = {o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv, o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv};
        st4h(op + dh, v);
      }
  }
}

// ---------------------------------------------------------------------------------------
// The same attention with a LAZY softmax reference (r4).  The loop above is bound by the vector ALU: per 64-key tile and wave 16
// MFMAs (512 cycles of matrix pipe) against ~1050 cycles of VALU issue - 32 v_exp_f32 (quarter rate: 512), 32 fma, 32 adds, 16
// v_max3, 16 conversions, and in most tiles the accumulator rescale (with 24 key tiles some one of a wave's 32 queries sees a new
// maximum in three tiles out of four).  None of that but the exponentials and the conversions is needed per tile:
//  * the scores leave the MFMA already relative to a per-query REFERENCE m_ref: the first Q.K^T MFMA of a tile takes C = -m_ref
//    (a 16-register splat kept across tiles) instead of 0, and Q arrives pre-multiplied by log2(e) / sqrt(64) (folded into the
//    query projection at load time), so p = v_exp_f32(accumulator) with no fma in front of it;
//  * the reference need not be the running maximum: any value within 2^15 below it gives the same quotient O / l in floating
//    point (P is rounded to f16 RELATIVE to its own magnitude, l and O are fp32), so it is raised only when a tile's weights
//    would leave the f16 range - which the tile's own row sum shows (sum < 2^15 => every p < 2^15): no maxima in the common path;
//  * the row sum is taken from the f16 pairs the P.V MFMA consumes, two keys per v_dot2c_f32_f16 (16 instructions, and l is the
//    sum of exactly the weights that multiply V).
// Common path per tile: 32 v_exp + 16 v_cvt_pk + 16 v_dot2c + one compare - 408 ticks by s_memtime stamps (tools/attn_lab.hip: the
// exponential costs ~8 per wave instruction here, not the 16 the estimates above assume; a lone wave's whole tile takes 1736, 512 of
// them on the matrix pipe: the loop is a serial chain per wave, and what it lacks at 158 VGPRs is a fourth wave per SIMD).  A tile whose sum reaches 2^15 (or the
// first tile of a workgroup, where nothing is known) takes the slow path: scores recomputed, true maximum, reference raised,
// accumulators rescaled - the textbook step, once or twice per workgroup instead of every tile.
// a * b that hipcc's SLP pass cannot pair with its neighbour into v_pk_mul_f32 / v_pk_add_f32 (the packed-f32
// hazard at three waves per SIMD): the product passes through an EMPTY asm, so the vectoriser's seed (a vector build, a store)
// never reaches the arithmetic.  No instruction lives inside the asm - hipcc still sees, schedules and hazard-checks the v_mul itself
// (a v_mul written IN asm consumed a v_exp result one instruction too early: the transcendental-use hazard is the compiler's to pad).
__device__ __forceinline__ float mul_scalar(float a, float b) { float t = a * b; asm("" : "+v"(t)); return t; }
__device__ __forceinline__ float add_scalar(float a, float b) { float t = a + b; asm("" : "+v"(t)); return t; }

template <bool SPLIT>
__global__ __launch_bounds__(256) void enc_attn_lazy_kernel(const f16* __restrict__ qk, const f16* __restrict__ vt,
                                                            f16* __restrict__ out, int T, int Tpad, int H, int d, float* part, unsigned* counters) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) f16 sK[2][AKT * ASTR];
  __shared__ __attribute__((aligned(16))) f16 sV[2][64 * ASTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qt = SPLIT ? (int)blockIdx.x >> 1 : (int)blockIdx.x, half = SPLIT ? (int)blockIdx.x & 1 : 0;
  const int q_row = qt * 128 + wave * 32 + l31;
  const int q_c = q_row < T ? q_row : T - 1;
  const int ld = 2 * d;

  f16x8 qf[4];
  {
    const f16* qp = qk + (size_t)(b * T + q_c) * ld + h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const f16x8*>(qp + kk * 16 + hi * 8);
  }
  const f16* kbase = qk + (size_t)b * T * ld + d + h * 64;
  const f16* vbase = vt + (size_t)(b * H + h) * 64 * Tpad;
  const int lrow0 = tid >> 3, lrow1 = (tid + 256) >> 3, lch = (tid & 7) * 8;
  const f16* vp0 = vbase + (size_t)lrow0 * Tpad + lch;
  const f16* vp1 = vbase + (size_t)lrow1 * Tpad + lch;
  const int so0 = lrow0 * ASTR + lch, so1 = lrow1 * ASTR + lch;
  // K / V tile on its way to LDS (loads and their hand-placed wait: see enc_attn_kernel).  Measured and not kept for the split-key
  // form: a second register set with tile kt+2 requested at the top of tile kt and a counted vmcnt(4) in front of the LDS stores
  // (correct, 202 VGPRs, 28.47 us per layer at one utterance against 28.46: the tile time there is not the load latency either).
  u32x4 ra[4];
#define WIS_GLOAD(kt, R)                                                                   \
  {                                                                                        \
    int key0 = (kt) * AKT + lrow0; if (key0 > T - 1) key0 = T - 1;                          \
    int key1 = (kt) * AKT + lrow1; if (key1 > T - 1) key1 = T - 1;                          \
    const f16* a0_ = kbase + (size_t)key0 * ld + lch; const f16* a1_ = kbase + (size_t)key1 * ld + lch;   \
    const f16* a2_ = vp0 + (kt) * AKT; const f16* a3_ = vp1 + (kt) * AKT;                   \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"  \
                 "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"      \
                 : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]) : "v"(a0_), "v"(a1_), "v"(a2_), "v"(a3_) : "memory"); \
  }
#define WIS_SSTORE(buf, R, WAIT)                                                           \
  asm volatile(WAIT : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) :: "memory");         \
  *reinterpret_cast<u32x4*>(&sK[buf][so0]) = R[0];                                         \
  *reinterpret_cast<u32x4*>(&sK[buf][so1]) = R[1];                                         \
  *reinterpret_cast<u32x4*>(&sV[buf][so0]) = R[2];                                         \
  *reinterpret_cast<u32x4*>(&sV[buf][so1]) = R[3];
  // S^T[key][q] - m_ref[q] = K . Q^T + C  (A = K rows, B = Q rows, C = the reference splat); keys >= T (last tile only) -> -inf
#define WIS_SCORES()                                                                       \
  _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                       \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                     \
      const f16x8 kf = *reinterpret_cast<const f16x8*>(&sK[cur][(t2 * 32 + l31) * ASTR + kk * 16 + hi * 8]); \
      st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], kk == 0 ? negm : st[t2], 0, 0, 0); \
    }                                                                                      \
  }                                                                                        \
  if (kt == nt_all - 1) {                                                                  \
    const int key_base = kt * AKT + 4 * hi;                                                \
    _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                       \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                     \
        const int key = key_base + t2 * 32 + (r & 3) + 8 * (r >> 2);                       \
        if (key >= T) st[t2][r] = -INFINITY;                                               \
      }                                                                                    \
  }
  // weights, their f16 pairs in P.V operand order (k-step s of 16 keys <-> S-tile s>>1, regs 8(s&1)..+7; slot j of half `hi` is
  // key 16s + 8(j>>2) + 4hi + (j&3): V^T is stored in exactly that order) and the row sum of the ROUNDED weights
#define WIS_WEIGHTS()                                                                      \
  rs = 0.f;                                                                                \
  _Pragma("unroll") for (int s = 0; s < 4; ++s)                                            \
    _Pragma("unroll") for (int j = 0; j < 8; j += 2) {                                     \
      const f16x2 pr = {(f16)__builtin_amdgcn_exp2f(st[s >> 1][8 * (s & 1) + j]), (f16)__builtin_amdgcn_exp2f(st[s >> 1][8 * (s & 1) + j + 1])}; \
      pf[s][j] = pr[0]; pf[s][j + 1] = pr[1];                                              \
      rs = __builtin_amdgcn_fdot2(pr, f16x2{(f16)1.f, (f16)1.f}, rs, false);               \
    }

  f32x16 o[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
  float m_run = 0.f, l_run = 0.f;      // m_run: the reference (log2 domain), NOT necessarily the running maximum

  const int nt_all = cdiv(T, AKT), nt_half = (nt_all + 1) >> 1;
  const int t_beg = SPLIT ? half * nt_half : 0;
  const int ntiles = SPLIT ? (t_beg + nt_half < nt_all ? t_beg + nt_half : nt_all) : nt_all;
  constexpr float BIG = 32768.f;
  // one key tile out of LDS buffer CUR: scores, weights, (rarely) the reference step, P.V.  (A macro, not a lambda: with the body behind a
  // lambda hipcc allocated 182 VGPRs for the unsplit instantiation - 168 with spills when held to three waves per SIMD - against 158 inline.)
  // In the reference step: on the first tile the reference becomes the tile's maximum whatever its sign (tile t_beg always holds a
  // key < T, so mx is finite; l and O are zero there, alpha = 0 keeps them so); both key halves of a query share one reference.
#define WIS_TILE(KT, CUR, FIRST)                                                           \
  {                                                                                        \
    const int kt = (KT), cur = (CUR); const bool first = (FIRST);                          \
    f32x16 st[2];                                                                          \
    f16x8 pf[4];                                                                           \
    float rs;                                                                              \
    bool slow = first;                                                                     \
    WIS_SCORES()                                                                           \
    if (!first) {                                                                          \
      WIS_WEIGHTS()                                                                        \
      slow = __any(!(rs < BIG));                                                           \
      if (slow) { WIS_SCORES() }                                                           \
    }                                                                                      \
    if (slow) {                                                                            \
      float mx = st[0][0];                                                                 \
_Pragma("unroll")                                                                          \
      for (int t2 = 0; t2 < 2; ++t2)                                                       \
_Pragma("unroll")                                                                          \
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t2][r]);                            \
      mx = fmaxf(mx, __shfl_xor(mx, 32));                                                  \
      const float delta = first ? mx : fmaxf(mx, 0.f);                                     \
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);                    \
      m_run += delta;                                                                      \
      l_run *= alpha;                                                                      \
_Pragma("unroll")                                                                          \
      for (int r = 0; r < 16; ++r) { o[0][r] = mul_scalar(o[0][r], alpha); o[1][r] = mul_scalar(o[1][r], alpha); negm[r] = -m_run; }\
_Pragma("unroll")                                                                          \
      for (int t2 = 0; t2 < 2; ++t2)                                                       \
_Pragma("unroll")                                                                          \
        for (int r = 0; r < 16; ++r) st[t2][r] -= delta;                                   \
      WIS_WEIGHTS()                                                                        \
    }                                                                                      \
    l_run += rs;                                                                           \
_Pragma("unroll")                                                                          \
    for (int dt = 0; dt < 2; ++dt)                                                         \
_Pragma("unroll")                                                                          \
      for (int s = 0; s < 4; ++s) {                                                        \
        const f16x8 vf = *reinterpret_cast<const f16x8*>(&sV[cur][(dt * 32 + l31) * ASTR + 16 * s + 8 * hi]);\
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], o[dt], 0, 0, 0);         \
      }                                                                                    \
  }
  WIS_GLOAD(t_beg, ra) WIS_SSTORE(0, ra, "s_waitcnt vmcnt(0)")
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the Q fragments are complete, and hipcc's waitcnt pass knows it (see enc_attn_kernel)
  __syncthreads();
  for (int kb = t_beg; kb < ntiles; ++kb) {      // (the tile macro declares kt / cur itself)
    const int cur0 = (kb - t_beg) & 1;
    if (kb + 1 < ntiles) WIS_GLOAD(kb + 1, ra)
    WIS_TILE(kb, cur0, kb == t_beg)
    if (kb + 1 < ntiles) { WIS_SSTORE(cur0 ^ 1, ra, "s_waitcnt vmcnt(0)") }
    __syncthreads();
  }
#undef WIS_GLOAD
#undef WIS_SSTORE
#undef WIS_SCORES
#undef WIS_WEIGHTS
#undef WIS_TILE
  if (SPLIT) {
    // (hand-off as in enc_attn_kernel; the merge formula holds for references as it does for maxima)
    __shared__ int s_last;
    const int nqt = (int)gridDim.x >> 1;
    const size_t pair = (size_t)(b * H + h) * nqt + qt;
    float* mine = part + (pair * 2 + half) * ENC_PART_FLOATS + tid;
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine), __float_as_uint(m_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 256), __float_as_uint(l_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_store(reinterpret_cast<unsigned*>(mine + (2 + a * 16 + r) * 256), __float_as_uint(o[a][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(counters + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == 1u;
      if (last) {
        __hip_atomic_store(counters + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const float* other = part + (pair * 2 + (half ^ 1)) * ENC_PART_FLOATS + tid;
    const float m1 = other[0], l1 = other[256];
    float o1[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) o1[a][r] = other[(2 + a * 16 + r) * 256];
    const float mm = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][r] = add_scalar(mul_scalar(o[a][r], a0), mul_scalar(o1[a][r], a1));      // the sum of two products: symmetric in the two states
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_row < T) {
    f16* op = out + (size_t)(b * T + q_row) * d + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dh = dt * 32 + 8 * r4 + 4 * hi;
        f32x4 v = {mul_scalar(o[dt][4 * r4], inv), mul_scalar(o[dt][4 * r4 + 1], inv), mul_scalar(o[dt][4 * r4 + 2], inv), mul_scalar(o[dt][4 * r4 + 3], inv)};
        st4h(op + dh, v);
      }
  }
}

// Which attention loop the encoder runs, fixed per process (model.hip folds log2(e) into the encoder's query projections for
// the lazy form: the two must agree).  WIS_ENC_ATTN_LAZY=0/1 is the A/B switch.
bool enc_attn_lazy() {
  static const bool on = !(getenv("WIS_ENC_ATTN_LAZY") && atoi(getenv("WIS_ENC_ATTN_LAZY")) == 0);
  return on;
}

// op-level entry only (wis_op_enc_attention keeps its contract "Q pre-scaled by 1/sqrt(64)"): the Q half of a [rows][2d] Q|K image
// times log2(e), in place on a private copy
__global__ void scale_q_log2e_kernel(f16* __restrict__ qk, int64_t rows, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  f16* p = qk + (i / d) * (2 * (int64_t)d) + i % d;
  *p = (f16)((float)*p * 1.4426950408889634f);
}
void launch_scale_q_log2e(hipStream_t st, f16* qk, int64_t rows, int d) {
  hipLaunchKernelGGL(scale_q_log2e_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, st, qk, rows, d);
}

// part / counters: ENC_PART_FLOATS floats per workgroup and one zeroed counter per (utterance, head, query tile); given and with
// at most 256 unsplit workgroups (one utterance of large-v2; round 3 split up to 600) the key range is split over two workgroups
size_t enc_attention_part_floats(int B, int T, int H) { return (size_t)B * H * cdiv(T, 128) * 2 * ENC_PART_FLOATS; }
int launch_enc_attention(hipStream_t st, const f16* qk, const f16* vt, f16* out, int B, int T, int Tpad, int H, float* part, unsigned* counters, size_t part_cap) {
  if (Tpad < cdiv(T, AKT) * AKT || Tpad % 8) { set_error("enc_attention: Tpad=%d too small for T=%d", Tpad, T); return WIS_E_ARG; }
  static const int env = getenv("WIS_ENC_ATTN_SPLIT") ? atoi(getenv("WIS_ENC_ATTN_SPLIT")) : -1;      // tuning: 0 never, 1 whenever possible
  const int wgs = cdiv(T, 128) * H * B;
  const bool split = part && counters && (size_t)wgs <= part_cap && cdiv(T, AKT) >= 4 && (env >= 0 ? env == 1 : wgs <= 256);      // never beyond the scratch it was given; split only while the unsplit grid leaves CUs empty (r4, sustained launches of the lazy loop, large-v2: one utterance = 240 workgroups 24.4 us unsplit / 23.6 split; two = 480: 35.8 / 41.9; three 49.7 / 57.6)
  if (enc_attn_lazy()) {
    if (split) hipLaunchKernelGGL((enc_attn_lazy_kernel<true>), dim3(2 * cdiv(T, 128), H, B), dim3(256), 0, st, qk, vt, out, T, Tpad, H, H * 64, part, counters);
    else hipLaunchKernelGGL((enc_attn_lazy_kernel<false>), dim3(cdiv(T, 128), H, B), dim3(256), 0, st, qk, vt, out, T, Tpad, H, H * 64, part, counters);
    return WIS_OK;
  }
  if (split) hipLaunchKernelGGL((enc_attn_kernel<true>), dim3(2 * cdiv(T, 128), H, B), dim3(256), 0, st, qk, vt, out, T, Tpad, H, H * 64, part, counters);
  else hipLaunchKernelGGL((enc_attn_kernel<false>), dim3(cdiv(T, 128), H, B), dim3(256), 0, st, qk, vt, out, T, Tpad, H, H * 64, part, counters);
  return WIS_OK;
}

}  // namespace wis
