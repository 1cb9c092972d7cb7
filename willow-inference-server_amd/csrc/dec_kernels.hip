// dec_kernels.hip — Whisper decoder step, logits processors and beam search on gfx950
// (the reference runs these inside ctranslate2 Whisper.generate,
// call site reference main.py:687-693, semantics SURVEY §3.4 + Appendix C).
//
// A decode step has M = B*beam <= 96 rows: every projection is a weight-streaming, HBM-bound
// skinny GEMM.  MI355X mapping:
//  * weights are re-packed once at load into MFMA A-fragment order ([N/16][K/32][64 lanes][8 f16])
//    so every wave-level load is ONE fully coalesced 1 KiB global_load_dwordx4 that feeds
//    v_mfma_f32_16x16x32_f16 directly (rows of the M<=16 activation block are the B operand, read
//    from LDS); 4 waves split K, partial accumulators meet in LDS, the epilogue (bias, GELU,
//    fp32 residual, KV-cache scatter) is fused.  The pre-LN LayerNorm is FOLDED into the
//    projection's weights (fold_ln_kernel): the kernel multiplies the raw rows and applies mean / rstd in its epilogue.
//  * self-attention KV cache: after every beam step the rows of each live beam are rewritten in place from its parent's
//    (kv_reorder_kernel), so the attention kernel's K/V addresses depend on nothing it has to load.
//  * cross-attention K is stored [H][dh/8][T][8] and V transposed [H][64][Tpad] so that both are MFMA A fragments read with
//    16-byte loads; the `beam` query rows of an utterance are folded into one pass over the utterance's K/V (shared per
//    utterance, never tiled per beam); the T axis is split into 256-key chunks across workgroups with an in-launch,
//    placement-independent last-arriver combine.
//  * suppress masks, log-softmax statistics and the top-2*beam candidates are computed in one
//    pass over the logits (16 chunks per row), the beam bookkeeping runs on device (one wave per
//    utterance) so the host never sees logits and a whole step replays as one HIP graph.
#include <string.h>
#include <atomic>
#include <mutex>
#include "common.hpp"
#include "kernels.hpp"

namespace wis {

// =======================================================================================
// packed image: [Npad/rows][K/32][4 k-quarters][rows][8 f16]; rows = 16 is the full MFMA A fragment ([64 lanes][8]),
// rows = 8 / 4 keep only the first rows of each fragment so that small-N matrices still spread over every CU.
__global__ void pack_gemv_kernel(const f16* __restrict__ W, f16* __restrict__ Wp, int N, int Npad, int K, int n_scale, float scale, int rows) {
  const int ksteps = K / 32;
  const size_t total = (size_t)Npad * ksteps * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx % rows);
    size_t t = idx / rows;
    const int kq = (int)(t & 3); t >>= 2;
    const int ks = (int)(t % ksteps), nt = (int)(t / ksteps);
    const int n = rows * nt + row, k = 32 * ks + 8 * kq;
    f16x8 v;
    if (n < N) {
      v = *reinterpret_cast<const f16x8*>(W + (size_t)n * K + k);
      if (n < n_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] * scale);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
    }
    *reinterpret_cast<f16x8*>(Wp + idx * 8) = v;
  }
}
// ---- 8-bit weights (compute type int8_float16; SURVEY §8(f)4, reference main.py:242): the same fragment order with one BYTE
// per weight, [Npad/16][K/32][64 lanes][8 B], stored offset-binary (q + 128) with one fp32 dequantisation scale per output row
// (CTranslate2's per-row scheme: scale = absmax / 127, q = rint(w / scale)).  The kernel widens a lane's 8 bytes to 8 exact
// f16 values with two byte permutes and two packed subtractions per dword ((0x6400 | u) is the f16 number 1024 + u), feeds
// the SAME f16 MFMA (activations stay f16, accumulation fp32) and multiplies the row scale into the epilogue.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <bool W8> struct WFrag;
template <> struct WFrag<false> {
  typedef u32x4 T;
  static __device__ __forceinline__ T zero() { return T{0u, 0u, 0u, 0u}; }
  static __device__ __forceinline__ f16x8 cvt(const T& w) { return *reinterpret_cast<const f16x8*>(&w); }
};
template <> struct WFrag<true> {
  typedef u32x2 T;
  static __device__ __forceinline__ T zero() { return T{0x80808080u, 0x80808080u}; }     // q = 0
  static __device__ __forceinline__ f16x8 cvt(const T& w) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const unsigned K64 = 0x64646464u;
    unsigned a[4] = {__builtin_amdgcn_perm(K64, w.x, 0x04010400u), __builtin_amdgcn_perm(K64, w.x, 0x04030402u),
                     __builtin_amdgcn_perm(K64, w.y, 0x04010400u), __builtin_amdgcn_perm(K64, w.y, 0x04030402u)};
    const h2 off = {(f16)1152.0f, (f16)1152.0f};
    f16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const h2 v = *reinterpret_cast<const h2*>(&a[i]) - off; r[2 * i] = v[0]; r[2 * i + 1] = v[1]; }
    return r;
  }
};
// per-row dequantisation scale (absmax / 127, times the folded query scaling) of W [N][K]; rows >= N get 0
__global__ void row_scale_kernel(const f16* __restrict__ W, float* __restrict__ scale, int N, int Npad, int K, int n_scale, float qscale) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n >= Npad) return;
  float mx = 0.f;
  if (n < N) for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf((float)W[(size_t)n * K + k]));
  mx = wave_max(mx);
  if (lane == 0) scale[n] = (n < N) ? (mx > 0.f ? mx / 127.0f : 1.0f) * (n < n_scale ? qscale : 1.0f) : 0.f;
}
__global__ void pack_gemv8_kernel(const f16* __restrict__ W, const float* __restrict__ scale, unsigned char* __restrict__ Wp, int N, int Npad, int K,
                                  int n_scale, float qscale) {
  const int ksteps = K / 32;
  const size_t total = (size_t)Npad * ksteps * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx & 15);
    size_t t = idx >> 4;
    const int kq = (int)(t & 3); t >>= 2;
    const int ks = (int)(t % ksteps), nt = (int)(t / ksteps);
    const int n = 16 * nt + row, k = 32 * ks + 8 * kq;
    unsigned char out[8];
    if (n < N) {
      const float inv = 1.0f / (scale[n] / (n < n_scale ? qscale : 1.0f));     // quantise the UNSCALED row; the folded factor lives in scale[]
      const f16x8 v = *reinterpret_cast<const f16x8*>(W + (size_t)n * K + k);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float q = rintf((float)v[j] * inv);
        q = fminf(fmaxf(q, -127.f), 127.f);
        out[j] = (unsigned char)((int)q + 128);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = 128;
    }
    *reinterpret_cast<u32x2*>(Wp + idx * 8) = *reinterpret_cast<const u32x2*>(out);
  }
}
int launch_pack_gemv8(hipStream_t st, const f16* W, unsigned char* Wp, float* scale, int N, int Npad, int K, int n_scale, float qscale) {
  if (K % 32 || Npad % 16 || Npad < N) { set_error("pack_gemv8: bad shape N=%d Npad=%d K=%d", N, Npad, K); return WIS_E_ARG; }
  hipLaunchKernelGGL(row_scale_kernel, dim3(Npad), dim3(64), 0, st, W, scale, N, Npad, K, n_scale, qscale);
  const size_t total = (size_t)Npad * (K / 32) * 4;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_gemv8_kernel, dim3(blocks), dim3(256), 0, st, W, scale, Wp, N, Npad, K, n_scale, qscale);
  return WIS_OK;
}

// ---- LayerNorm folded into the following projection (pre-LN blocks: y = W . LN(x) + b).  With W' = W o gamma (columns scaled),
//   y_n = rs * ( sum_k W'_nk x_k  -  mu * c_n ) + b'_n,     c_n = sum_k W'_nk,     b'_n = b_n + sum_k W_nk beta_k,
// so the skinny GEMM runs on the RAW activation row (cast to f16) and the row statistics (mu, rs) are only needed in its
// epilogue - the normalise-and-restage pass, its barrier and the gamma / beta loads leave the kernel's critical path.
// One wave per matrix row: rewrites W in place as f16(W * gamma), accumulates c from the ROUNDED products (what the MFMA
// will see) and adds W . beta to the bias.  Rows < n_scale carry the folded query scaling (applied to W by the packer).
__global__ void fold_ln_kernel(f16* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ bias,
                               float* __restrict__ csum, int N, int K, int n_scale, float qscale) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n >= N) return;
  float c = 0.f, wb = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = (float)W[(size_t)n * K + k];
    const f16 wg = (f16)(w * gamma[k]);
    W[(size_t)n * K + k] = wg;
    c += (float)wg; wb += w * beta[k];
  }
  c = wave_sum(c); wb = wave_sum(wb);
  const float f = n < n_scale ? qscale : 1.0f;
  if (lane == 0) { csum[n] = c * f; bias[n] += wb * f; }
}
// 8-bit variant of c: from the de-quantised values the MFMA will see (scale_n * sum_k q_nk; scale[] already carries qscale)
__global__ void csum8_kernel(const f16* __restrict__ W, const float* __restrict__ scale, float* __restrict__ csum, int N, int K, int n_scale, float qscale) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n >= N) return;
  const float inv = 1.0f / (scale[n] / (n < n_scale ? qscale : 1.0f));
  float c = 0.f;
  for (int k = lane; k < K; k += 64) { float q = rintf((float)W[(size_t)n * K + k] * inv); c += fminf(fmaxf(q, -127.f), 127.f); }
  c = wave_sum(c);
  if (lane == 0) csum[n] = c * scale[n];
}
int launch_fold_ln(hipStream_t st, f16* W, const float* gamma, const float* beta, float* bias, float* csum, int N, int K, int n_scale, float qscale) {
  hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(64), 0, st, W, gamma, beta, bias, csum, N, K, n_scale, qscale);
  return WIS_OK;
}
int launch_csum8(hipStream_t st, const f16* W, const float* scale, float* csum, int N, int K, int n_scale, float qscale) {
  hipLaunchKernelGGL(csum8_kernel, dim3(N), dim3(64), 0, st, W, scale, csum, N, K, n_scale, qscale);
  return WIS_OK;
}

// tile height of an [N][K] decoder matrix: the full 16-row MFMA fragment unless the matrix is both narrow (fewer than
// ~200 tiles) and deep (K >= 2048: FFN2), where 4-row tiles spread the long per-tile stream over every CU.  For the
// narrow d x d matrices the kernel is latency-bound and extra workgroups only add prologue work.
int gemv_rows_for(int N, int K) {
  (void)N; (void)K;
  return 16;   // measured on MI355X (round 1): the K = 5120 stream is already HBM-bound with 80 workgroups (13 MB in ~6k
               // cycles = 4.5 TB/s); 4-row tiles and a 40-deep prefetch cost ~1.5k cycles of issue per launch and gained nothing
}
int launch_pack_gemv(hipStream_t st, const f16* W, f16* Wp, int N, int Npad, int K, int n_scale, float scale, int rows) {
  if (K % 32 || Npad % rows || Npad < N || (rows != 16 && rows != 8 && rows != 4)) { set_error("pack_gemv: bad shape N=%d Npad=%d K=%d rows=%d", N, Npad, K, rows); return WIS_E_ARG; }
  const size_t total = (size_t)Npad * (K / 32) * 4;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_gemv_kernel, dim3(blocks), dim3(256), 0, st, W, Wp, N, Npad, K, n_scale, scale, rows);
  return WIS_OK;
}

// 8-byte {tag, value} granules: one relaxed agent-scope (write-through) store each, "the data is the flag" (guide G16 form R2) - the hand-off of the
// decoder cross-attention's chunk partials (dec_cross_attn_kernel SPIN) and of q / k / v from the QKV projection to the self-attention fused into its launch
typedef unsigned long long gran_t;
__device__ __forceinline__ void st_gran(gran_t* p, unsigned tag, float v) {
  __hip_atomic_store(p, ((gran_t)tag << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ gran_t ld_gran(const gran_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr unsigned CA_SPIN_LIMIT = 1u << 17;      // sweeps before the combiner gives up (~0.1 s)

// ---- self-attention FUSED into the QKV projection's launch (round 6; round-5 review item 4a) -----------------------------------------
// The one-utterance decode step ran  QKV projection -> [kernel boundary] -> dec_self_attn_kernel  per layer: a 1.6 us dispatch boundary plus a 5 us
// kernel whose only inputs from the projection are 64 floats of q and one new K / V row per (row, head).  Now the projection's launch carries
// H extra workgroups behind its n-tiles - one per head, dispatched after every producer tile of the launch - that run dec_self_attn_kernel's
// arithmetic (same lane mapping, same order: bit-identical results): each wave owns rows w and w + 4, requests the row's K / V HISTORY from the
// cache at once (it was written by earlier steps: nothing to wait for), then polls the 192 granules the projection's epilogue publishes for its
// (row, head) - q[64], k[64], v[64] of the CURRENT position, tagged with the head's epoch + 1 - and finishes.  The boundary, the kernel's own
// start-up and its history round trip are gone from the chain; what is left behind the last projection tile is one L2 round trip and the softmax.
// Hand-off rules as in dec_cross_attn_kernel's granule form: producers store and leave (no drain, no ticket); the head's consumer advances the
// epoch word at the end of the launch (tags only grow: graph replays stay valid, nothing is reset); a consumer's spin is bounded and raises the
// cross-attention's give-up flag word (the host repeats the call with both hand-offs in their kernel-boundary forms); consumers are dispatched
// behind their producers, so they can never hold a slot a producer of the same launch still needs.
// Only for decode rows that own their KV slot (rmul = 1: the ordinary step; prefill rows of one slot read each other's rows of the same launch).
__device__ __forceinline__ void sa_consume(const GemvP& p, const int h, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = lane >> 3, c = lane & 7;
  const int M = p.M, d = p.d, ctx = p.ctx;
  float* shq = reinterpret_cast<float*>(smem) + wave * 512;                  // per wave: q f32 [64] | k f16 [64] | v f16 [64] | red f32 [4][64]
  f16* shk = reinterpret_cast<f16*>(shq + 64); f16* shv = shk + 64; float* red = shq + 128;
  const unsigned tag = __hip_atomic_load(p.sa_epoch + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int hoff = h * 64 + 8 * c;
  for (int m = wave; m < M; m += 4) {
    const int ls = (m / p.sa_rpu) * p.sa_sstride + (m % p.sa_rpu);
    const int len = p.pos[m] + 1;
    const f16* krow = p.kc + (size_t)ls * ctx * d + hoff;
    const f16* vrow = p.vc + (size_t)ls * ctx * d + hoff;
    u32x4 kr[8], vr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kr[i] = *reinterpret_cast<const u32x4*>(krow + (size_t)(8 * i + pl) * d);      // history (position len - 1 is replaced below)
#pragma unroll
    for (int i = 0; i < 8; ++i) vr[i] = *reinterpret_cast<const u32x4*>(vrow + (size_t)(8 * i + pl) * d);
    // the current position's q / k / v: lane owns element `lane` of each
    const gran_t* g = p.sa_gran + ((size_t)(h * 8 + m) * 3) * 64 + lane;
    gran_t gq, gk, gv;
    for (unsigned spins = 0;; ++spins) {
      gq = ld_gran(g); gk = ld_gran(g + 64); gv = ld_gran(g + 128);
      const bool ok = ((unsigned)(gq >> 32) == tag) & ((unsigned)(gk >> 32) == tag) & ((unsigned)(gv >> 32) == tag);
      if (__ballot(!ok) == 0ull) break;
      if (spins > CA_SPIN_LIMIT) { if (lane == 0) atomicOr(p.sa_flag, 1u); break; }      // the give-up flag word of the cross-attention hand-off: one host protocol for both
      __builtin_amdgcn_s_sleep(1);
    }
    shq[lane] = __uint_as_float((unsigned)gq); shk[lane] = (f16)__uint_as_float((unsigned)gk); shv[lane] = (f16)__uint_as_float((unsigned)gv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = shq[8 * c + j];
    const u32x4 kcur = *reinterpret_cast<const u32x4*>(shk + 8 * c), vcur = *reinterpret_cast<const u32x4*>(shv + 8 * c);
    float m_run = -INFINITY, l_run = 0.f;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    int p0 = 0;
    do {
      if (p0 > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int pp = p0 + 8 * i + pl, pc = pp < len ? pp : len - 1;
          kr[i] = *reinterpret_cast<const u32x4*>(krow + (size_t)pc * d);
          vr[i] = *reinterpret_cast<const u32x4*>(vrow + (size_t)pc * d);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { const bool cur = p0 + 8 * i + pl == len - 1; kr[i] = cur ? kcur : kr[i]; vr[i] = cur ? vcur : vr[i]; }
      float sc[8]; float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pp = p0 + 8 * i + pl;
        float dot = 0.f;
        if (pp < len) {
          const f16x8 kv = *reinterpret_cast<const f16x8*>(&kr[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) dot = fmaf((float)kv[j], qv[j], dot);
        }
        dot += dpp_f<0xB1>(dot); dot += dpp_f<0x4E>(dot); dot += dpp_f<0x141>(dot);
        sc[i] = (pp < len) ? dot : -INFINITY;
        mx = fmaxf(mx, sc[i]);
      }
      mx = wave_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float pw = __expf(sc[i] - m_new);
        lsum += pw;
        if (p0 + 8 * i + pl < len) {
          const f16x8 vv = *reinterpret_cast<const f16x8*>(&vr[i]);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(pw, (float)vv[j], acc[j]);
        }
      }
      l_run = l_run * alpha + wave_sum(lsum) * 0.125f;
      m_run = m_new;
      p0 += 64;
    } while (p0 < len);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += dpp_f<0x128>(acc[j]);
    if ((lane & 8) == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(lane >> 4) * 64 + 8 * c + j] = acc[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float o = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    p.sa_out[(size_t)m * d + h * 64 + lane] = (f16)(o / l_run);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // (the LDS row is reused by this wave's next row)
  }
  __syncthreads();
  if (tid == 0) __hip_atomic_store(p.sa_epoch + h, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch's epoch
}

// =======================================================================================
// Skinny GEMM.  grid = Npad/16 workgroups of 4 waves; wave w streams the k-steps of its quarter of
// every staged K-chunk.  Dynamic LDS: xs f16 [M][KC+8] | red f32 [4][MB][64][4] | stats f32 [48][2] | sred f32 [4][8]
//
// Memory-latency structure (one dependent round trip): the activation loads (L2-hot, written by the
// previous kernel) are issued FIRST, then a 16-deep prefetch of this wave's weight fragments (HBM);
// vmcnt retires in order, so LayerNorm statistics and the f16 staging of x run from registers while
// the weight stream is in flight, and the MFMAs consume the fragments as they land, refilling the ring.
// SC (template): compile-time k-steps per wave (K / 128) for the common single-chunk shapes {3,4,6,8,10} so the prefetch and
// the MFMA loop are straight-line code (every runtime `u < S` guard would be a scalar branch around one load); SC = 0 is the
// generic ring of 16 fragments with refill.

// WIS_EP_EARLY (build flag, default 0): request the epilogue operands (bias, folded column sums, row scales, residual / KV-cache
// slot) right BEHIND the weight prefetch instead of after the last MFMA (vmcnt retires in order, so they cannot delay the
// activations or a weight fragment).  Measured on MI355X (round 2, build.py --variant, tools/run_r2c.sh): 33.12 vs 32.82 ms per
// utterance at B = 1, 97.8 vs 97.7 ms at B = 8 - no gain: the epilogue round trip is already hidden behind the cross-wave
// reduction, and the extra 15-18 live VGPRs cost as much as they save.  The late form stays the product default.
#ifndef WIS_EP_EARLY
#define WIS_EP_EARLY 0
#endif
// MODE 0: generic staging from global; 1: fast LayerNorm prologue from registers; 2: fast f16 activations from registers
// NT (r5; MB = 1, single chunk, compile-time SC only): n-tiles per workgroup.  2: waves 0-1 own tile 2 nt, waves 2-3 tile 2 nt + 1, half of
// K each (SC = K / 64 fragments per wave, all requested up front) - half as many workgroups stage the activation rows, each with twice
// the weight bytes in flight: N = 4d of the one-utterance step is 160 workgroups (one per CU) instead of 320 on 256 CUs.
template <int MB, int MODE, int SC, int RM, bool W8, int NT = 1, bool SA = false>
__device__ __forceinline__ void gemv_body(const GemvP& p, const int KC, const int nt, char* smem) {
  static_assert(NT == 1 || (NT == 2 && MB == 1 && SC > 0), "two-tile workgroups: <= 16 rows, every fragment prefetched");
  typedef typename WFrag<W8>::T WT;
  constexpr int GV_PF = SC > 0 ? SC : 16;
  // full MFMA A fragments.  4- / 8-row tiles were measured twice: packed that way (round 1) and as HALF tiles of the ordinary image (round 6: 160 workgroups
  // addressing rows 8 h .. 8 h + 7 of every fragment for the one-utterance FFN2, 1.2152 / 1.2137 vs 1.2152 / 1.2159 ms per step, session r6L) - no gain: a
  // wave's request costs the CU's address path the same ~50 cycles whether 64 or 32 of its lanes are active, and the stream is those requests
  constexpr int rows = 16;
  const int M = p.M, K = p.K;
  const int xstr = KC + 8;
  f16* xs = reinterpret_cast<f16*>(smem);
  float* red = reinterpret_cast<float*>(smem + (((size_t)M * xstr * 2 + 15) & ~(size_t)15));
  float* stats = red + 4 * MB * 64 * 4;
  float* sred = stats + 2 * MAX_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ksteps = K / 32;
  const int S = SC > 0 ? SC : KC / 128;          // k-steps per wave per chunk
  const int wt = NT == 2 ? wave >> 1 : 0;        // the wave's n-tile within the workgroup
  const int ksl0 = (NT == 2 ? (wave & 1) : wave) * S;      // first chunk-local k-step of this wave
  // fragment (tile, k-step) = 4*rows 16-byte pieces: piece (kq, row) at kq*rows + row; lanes with row >= rows stay zero
  const bool wact = (lane & 15) < rows;
  const int wstep = 4 * rows;                     // pieces per k-step
  const WT* wp4 = reinterpret_cast<const WT*>(p.Wp) + (size_t)(nt * NT + wt) * ksteps * wstep + (lane >> 4) * rows + (lane & 15);
  const WT wzero = WFrag<W8>::zero();
  const int k4n = K >> 2;                        // float4 per row
  // LayerNorm-folded path (MODE 1; host-selected, M <= 8, K <= 2048): the raw fp32 rows live in registers, thread owns float4
  // columns k4 = tid, tid + 256 of every row; they are cast to f16 and staged at once, the statistics (single shifted pass,
  // c = x[r][0]: var = E[(x-c)^2] - E[x-c]^2) are reduced AFTER the MFMA loop and meet the accumulators in the epilogue.
  // MODE 3 (r5, GV_LN16): the same on the F16 copy of the rows that whoever produces residual rows leaves next to them (12.8 KB per
  // workgroup instead of 25.6 KB): thread t < K / 8 owns the 16-byte chunk t of every row - five requests per thread in 2.5 waves where
  // the fp32 form issues ten in four -, stages it as it is, and takes the statistics from those same values (the ones the MFMAs consume).
  constexpr bool fast = MODE == 1 || MODE == 3;
  constexpr bool x16 = MODE == 3;
  constexpr int RMAX = fast ? RM : 1;        // RM in {3, 5, 8}: smallest that holds M (rows >= M are clamped duplicates)
  f32x4 xv[RMAX][2];      // (vector type, not HIP's float4 struct: hipcc split each struct load into an overlapping dwordx2 + dwordx3 pair)
  u32x4 xv16[RMAX];
  unsigned c16raw[RMAX];      // (whichever set the instantiation does not use is never materialised)
  float cshift[RMAX];
  unsigned long long* pf = (nt == 0 && tid == 0) ? p.prof : nullptr;
  if (tid == 0) tl_begin(p.prof);
  stamp(pf, 0);
  // fused self-attention (sa_consume): this tile's head and the tag its granules carry (requested with everything else)
  unsigned sa_tag = 0;
  if (SA) { const int nh = rows * nt, sec = nh / p.d; sa_tag = __hip_atomic_load(p.sa_epoch + ((nh - sec * p.d) >> 6), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u; }
  if (x16) {
    const u32x4* x8 = reinterpret_cast<const u32x4*>(p.x);
    const int k8n16 = K >> 3, t8 = tid < k8n16 ? tid : k8n16 - 1;      // (clamped address: threads >= K / 8 load a valid chunk and drop it)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { const int rr = r < M ? r : M - 1; xv16[r] = x8[(size_t)rr * k8n16 + t8]; }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { const int rr = r < M ? r : M - 1; c16raw[r] = uniform_load_issue_u32(reinterpret_cast<const f16*>(p.x) + (size_t)rr * K); }
  }
  if (fast && !x16) {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(p.x);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k4 = tid + 256 * j;
      if (k4 < k4n) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { const int rr = r < M ? r : M - 1; xv[r][j] = x4[(size_t)rr * k4n + k4]; }   // rows >= M: clamped duplicates
      }
    }
#pragma unroll
    // (the shifts are uniform over the workgroup: fetched through the SCALAR cache - as vector loads they were five of a wave's 25 requests
    // on the CU's address path, the unit the start of these kernels is bound by)
#ifdef WIS_CSHIFT_VLOAD      // (A/B build: the round-2..4 form, one vector load per row)
    for (int r = 0; r < RMAX; ++r) { const int rr = r < M ? r : M - 1; cshift[r] = reinterpret_cast<const float*>(p.x)[(size_t)rr * K]; }
#else
    for (int r = 0; r < RMAX; ++r) { const int rr = r < M ? r : M - 1; cshift[r] = uniform_load_issue_f32(reinterpret_cast<const float*>(p.x) + (size_t)rr * K); }
#endif
  }
  // f16 activations (attention / FFN hidden output of the previous kernel): same idea, up to 13 x 16 B per thread
  // (MB >= 2: up to 48 rows x 1280 columns per chunk = 30 x 16 B per thread; K larger than the chunk is walked chunk by
  // chunk, the next chunk's activations are fetched into registers while the matrix cores work on the current one)
  constexpr bool fastx = MODE == 2;
  constexpr int NXH = fastx ? (MB == 1 ? 13 : 30) : 1;
  u32x4 xh[NXH];
  const int c8 = KC >> 3, k8n = K >> 3, nx = M * c8;     // 16-byte pieces per chunk row / per full row / per chunk
  // piece index -> row without an integer division (hipcc expands `idx / c8` by a run-time divisor into ~30 instructions, thirteen
  // times per thread, in front of the weight prefetch): (idx + 0.5) * (1 / c8) truncates to the exact quotient for idx < 2^20
  const float inv_c8 = 1.0f / (float)c8;
#define WIS_ROW_OF(idx) ((int)(((float)(idx) + 0.5f) * inv_c8))
  if (fastx) {
    const u32x4* x8 = reinterpret_cast<const u32x4*>(p.x);
    if (p.x2) {                 // columns >= xsplit come from a second row-major matrix (the fused out-proj + cross-Q stage)
      const u32x4* y8 = reinterpret_cast<const u32x4*>(p.x2);
      const int s8 = p.xsplit >> 3, r8 = c8 - s8;
#pragma unroll
      for (int i = 0; i < NXH; ++i) {
        const int idx = tid + 256 * i;
        if (idx < nx) { const int row = WIS_ROW_OF(idx), col = idx - row * c8; xh[i] = col < s8 ? x8[(size_t)row * s8 + col] : y8[(size_t)row * r8 + (col - s8)]; }
      }
    } else if (MB == 1 || KC == K) {   // MB == 1: the launcher only selects this mode for a single chunk
#pragma unroll
      for (int i = 0; i < NXH; ++i) { const int idx = tid + 256 * i; if (idx < nx) xh[i] = x8[idx]; }
    } else {
#pragma unroll
      for (int i = 0; i < NXH; ++i) { const int idx = tid + 256 * i; if (idx < nx) { const int row = WIS_ROW_OF(idx); xh[i] = x8[(size_t)row * k8n + (idx - row * c8)]; } }
    }
  }
  // weight prefetch for chunk 0 (independent of x)
  WT wf[GV_PF];
  {
    const WT* wq = wp4 + (size_t)ksl0 * wstep;
#pragma unroll
    for (int u = 0; u < GV_PF; ++u) { wf[u] = wzero; if ((SC > 0 || u < S) && wact) wf[u] = __builtin_nontemporal_load(wq + (size_t)u * wstep); }
  }

  // epilogue operands (bias, folded column sums, row scales, residual, KV-cache row)
  const bool ep_act = tid < MB * NT * 64;        // (NT == 2: wave 0 finishes tile 2 nt, wave 1 tile 2 nt + 1)
  const int ep_m = NT == 2 ? (lane & 15) : (tid >> 6) * 16 + (lane & 15), ep_n = rows * (nt * NT + (NT == 2 ? (tid >> 6) & 1 : 0)) + 4 * (lane >> 4);
  const bool ep_ok = ep_act && ep_m < M && ep_n < p.N && 4 * (lane >> 4) < rows;
  float4 ep_bias = make_float4(0.f, 0.f, 0.f, 0.f), ep_cs = ep_bias, ep_res = ep_bias, ep_sc = make_float4(1.f, 1.f, 1.f, 1.f);
  int ep_slot = 0, ep_pos = 0;
  // GV_LNP (f16 rows + per-16-column (sum, M2) partials of the rows in stat_in): mean / rstd merged Chan-style about the first
  // tile's mean by the four lanes of a row, a quarter of the K / 16 pairs each - gemv_frag_body's form.  The consumer then needs
  // neither the fp32 rows (a third of the load instructions of a LayerNorm-folded launch were those) nor a pass over them.
  const bool lnp = fastx && (p.flags & GV_LNP);
  constexpr int LNQ = 20;                     // pairs per lane: K / 64 <= 20 (K <= 1280); larger K takes the loop form
  float2 lnp_pr[LNQ]; float lnp_c = 0.f;
#define WIS_EP_LOADS()                                                                                              \
  if (lnp && ep_act) {                                                                                              \
    const int mm = ep_m < M ? ep_m : M - 1, nq = K >> 6;                                                            \
    const float2* row = reinterpret_cast<const float2*>(p.stat_in) + (size_t)mm * (K >> 4);                         \
    const float2* sp = row + (size_t)(lane >> 4) * nq;                                                              \
    lnp_c = row[0].x * 0.0625f;                                                                                     \
    _Pragma("unroll") for (int i = 0; i < LNQ; ++i) lnp_pr[i] = sp[i < nq ? i : nq - 1];                            \
  }                                                                                                                 \
  if (ep_ok) {                                                                                                      \
    if (p.bias) ep_bias = *reinterpret_cast<const float4*>(p.bias + ep_n);                                          \
    if (fast || lnp) ep_cs = *reinterpret_cast<const float4*>(p.csum + ep_n);                                       \
    if (W8) ep_sc = *reinterpret_cast<const float4*>(p.wscale + ep_n);                                              \
    if (p.flags & GV_QKV) { if (ep_n >= p.d) { ep_slot = p.slot[ep_m]; ep_pos = p.pos[ep_m]; } }                    \
    else if (p.flags & GV_RESID) ep_res = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.y) + (size_t)ep_m * p.N + ep_n); \
  }
  constexpr bool ep_early = WIS_EP_EARLY && MB == 1 && SC > 0;    // single chunk, every weight fragment already requested
  if (ep_early) { WIS_EP_LOADS() }
  stamp(pf, 1);
  float sa[RMAX], sb[RMAX];
  if (x16) {
    uniform_load_wait();
#pragma unroll
    for (int r = 0; r < RMAX; ++r) uniform_load_landed(c16raw[r]);
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int rr = r < M ? r : M - 1;
      cshift[r] = (float)__builtin_bit_cast(f16, (unsigned short)(c16raw[r] & 0xFFFFu));
      sa[r] = 0.f; sb[r] = 0.f;
      if (tid < (K >> 3)) {
        *reinterpret_cast<u32x4*>(xs + (size_t)rr * xstr + tid * 8) = xv16[r];
        const f16x8 h = *reinterpret_cast<const f16x8*>(&xv16[r]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = (float)h[e] - cshift[r]; sa[r] += v; sb[r] += v * v; }
      }
    }
    stamp(pf, 2);
  } else
  if (fast) {
#ifndef WIS_CSHIFT_VLOAD
    uniform_load_wait();      // the rows' shifts (requested with the activation rows, long landed)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) uniform_load_landed(cshift[r]);
#endif
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int rr = r < M ? r : M - 1;       // clamped rows rewrite row M-1 with identical values (benign)
      sa[r] = 0.f; sb[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k4 = tid + 256 * j;
        if (k4 < k4n) {
          const f16x4 o = {(f16)xv[r][j][0], (f16)xv[r][j][1], (f16)xv[r][j][2], (f16)xv[r][j][3]};
          *reinterpret_cast<f16x4*>(xs + (size_t)rr * xstr + k4 * 4) = o;
          const float a = xv[r][j][0] - cshift[r], b = xv[r][j][1] - cshift[r], c = xv[r][j][2] - cshift[r], e = xv[r][j][3] - cshift[r];
          sa[r] += (a + b) + (c + e); sb[r] += (a * a + b * b) + (c * c + e * e);
        }
      }
    }
    stamp(pf, 2);
  } else if (fastx) {
#pragma unroll
    for (int i = 0; i < NXH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < nx) { const int row = WIS_ROW_OF(idx), k8 = idx - row * c8; *reinterpret_cast<u32x4*>(xs + (size_t)row * xstr + k8 * 8) = xh[i]; }
    }
  }

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { int r = mb * 16 + (lane & 15); xrow[mb] = (r < M ? r : M - 1) * xstr + 8 * (lane >> 4); }

  const int c8n = KC / 8;
  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    const WT* wq = wp4 + (size_t)(kc0 / 32 + ksl0) * wstep;
    const bool more = MB > 1 && kc0 + KC < K;     // multi-chunk register staging exists for the batched row counts only
    if (kc0 > 0 && SC == 0) {
#pragma unroll
      for (int u = 0; u < GV_PF; ++u) if (u < S && wact) wf[u] = __builtin_nontemporal_load(wq + (size_t)u * wstep);
    }
    if (fastx && more) {   // next chunk's activations: in flight during this chunk's MFMAs
      const u32x4* x8 = reinterpret_cast<const u32x4*>(p.x) + ((kc0 + KC) >> 3);
#pragma unroll
      for (int i = 0; i < NXH; ++i) { const int idx = tid + 256 * i; if (idx < nx) { const int row = WIS_ROW_OF(idx); xh[i] = x8[(size_t)row * k8n + (idx - row * c8)]; } }
    }
    if (!fast && !fastx) {
      // stage x[:, kc0:kc0+KC] (f16 activations)
      {
        for (int idx = tid; idx < M * c8n; idx += 256) {
          const int r = idx / c8n, c8 = idx - r * c8n, k = kc0 + c8 * 8;
          *reinterpret_cast<f16x8*>(xs + (size_t)r * xstr + c8 * 8) =
              *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.x) + (size_t)r * K + k);
        }
      }
    }
    __syncthreads();
    stamp(pf, 3);
    if (SC > 0) {
#pragma unroll
      for (int u = 0; u < GV_PF; ++u) {
        const f16x8 a = WFrag<W8>::cvt(wf[u]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const f16x8 xb = *reinterpret_cast<const f16x8*>(xs + xrow[mb] + (ksl0 + u) * 32);
          acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb, acc[mb], 0, 0, 0);
        }
        if (more && wact) wf[u] = __builtin_nontemporal_load(wq + (size_t)(KC / 32 + u) * wstep);   // same slot, next chunk
      }
    } else {
      for (int base = 0; base < S; base += GV_PF) {
#pragma unroll
        for (int u = 0; u < GV_PF; ++u) {
          if (base + u < S) {
            const f16x8 a = WFrag<W8>::cvt(wf[u]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const f16x8 xb = *reinterpret_cast<const f16x8*>(xs + xrow[mb] + (ksl0 + base + u) * 32);
              acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb, acc[mb], 0, 0, 0);
            }
            if (base + u + GV_PF < S && wact) wf[u] = __builtin_nontemporal_load(wq + (size_t)(base + u + GV_PF) * wstep);
          }
        }
      }
    }
    __syncthreads();
    if (fastx && more) {
#pragma unroll
      for (int i = 0; i < NXH; ++i) {
        const int idx = tid + 256 * i;
        if (idx < nx) { const int row = WIS_ROW_OF(idx), k8 = idx - row * c8; *reinterpret_cast<u32x4*>(xs + (size_t)row * xstr + k8 * 8) = xh[i]; }
      }
    }
  }
  stamp(pf, 4);
  // (late form: requested NOW - every weight fragment has been consumed, so they cannot hold up the in-order vmcnt queue - and
  // covered by the reductions and the barrier below; requested BEFORE the activations they delayed them and cost more than they saved)
  if (!ep_early) { WIS_EP_LOADS() }
#undef WIS_EP_LOADS
  if (fast) {   // LayerNorm statistics of the folded form: reduced here, behind the MFMAs, published with the accumulators
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const float ta = wave_sum(sa[r]), tb = wave_sum(sb[r]);
      if (lane == 0) { sred[wave * 16 + 2 * r] = ta; sred[wave * 16 + 2 * r + 1] = tb; }
    }
  }
  // cross-wave reduction; D[i = n][j = m]: lane holds m = lane&15, n = 4*(lane>>4) + r
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    *reinterpret_cast<float4*>(red + ((size_t)(wave * MB + mb) * 64 + lane) * 4) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  __syncthreads();
  stamp(pf, 5);
  if (tid < MB * NT * 64) {
    const int mb = NT == 2 ? 0 : tid >> 6, ln = tid & 63;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NT == 2) {      // the two K halves of this wave's tile
      const int w0 = 2 * (tid >> 6);
      const float4 t0 = *reinterpret_cast<const float4*>(red + ((size_t)w0 * 64 + ln) * 4), t1 = *reinterpret_cast<const float4*>(red + ((size_t)(w0 + 1) * 64 + ln) * 4);
      s.x = t0.x + t1.x; s.y = t0.y + t1.y; s.z = t0.z + t1.z; s.w = t0.w + t1.w;
    } else {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)(w * MB + mb) * 64 + ln) * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    }
    const int m = ep_m, n = ep_n;
    float4 rsum = make_float4(0.f, 0.f, 0.f, 0.f);      // the residual row values this lane stored (statistics partials below)
    float lnp_mu = 0.f, lnp_rs = 1.f;
    if (lnp) {      // whole wave: ((q0 + q1) + (q2 + q3)) of the row's four quarter sums on every lane
      const int nq = K >> 6;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < LNQ; ++i) if (i < nq) { const float dm = lnp_pr[i].x * 0.0625f - lnp_c; t1 += dm; t2 += lnp_pr[i].y + 16.0f * dm * dm; }
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      const float invK = 1.0f / (float)K, dmu = t1 * 16.0f * invK;
      lnp_mu = lnp_c + dmu;
      lnp_rs = 1.0f / sqrtf(fmaxf(t2 * invK - dmu * dmu, 0.f) + 1e-5f);
    }
    if (ep_ok) {
      if (W8) { s.x *= ep_sc.x; s.y *= ep_sc.y; s.z *= ep_sc.z; s.w *= ep_sc.w; }
      if (lnp) { s.x = lnp_rs * (s.x - lnp_mu * ep_cs.x); s.y = lnp_rs * (s.y - lnp_mu * ep_cs.y); s.z = lnp_rs * (s.z - lnp_mu * ep_cs.z); s.w = lnp_rs * (s.w - lnp_mu * ep_cs.w); }
      if (fast) {   // y = rs * (W' x - mu * c) [+ b' below]
        const int r = m < RMAX ? m : RMAX - 1;
        const float invK = 1.0f / (float)K;
        const float A = ((sred[2 * r] + sred[16 + 2 * r]) + (sred[32 + 2 * r] + sred[48 + 2 * r])) * invK;
        const float Bq = ((sred[2 * r + 1] + sred[17 + 2 * r]) + (sred[33 + 2 * r] + sred[49 + 2 * r])) * invK;
        float shift = cshift[0];                          // the row's shift x[m][0], still in registers
#pragma unroll
        for (int q = 1; q < RMAX; ++q) shift = (r == q) ? cshift[q] : shift;
        const float mu = shift + A;
        const float rs = 1.0f / sqrtf(fmaxf(Bq - A * A, 0.f) + 1e-5f);
        s.x = rs * (s.x - mu * ep_cs.x); s.y = rs * (s.y - mu * ep_cs.y); s.z = rs * (s.z - mu * ep_cs.z); s.w = rs * (s.w - mu * ep_cs.w);
      }
      s.x += ep_bias.x; s.y += ep_bias.y; s.z += ep_bias.z; s.w += ep_bias.w;
      if (p.flags & GV_QKV) {
        const int d = p.d;
        if (SA) {      // q / k / v of the current position to the head's consumer workgroup of this launch: four granules per lane
          const int sec = n / d, hn = n - sec * d;
          gran_t* gp = p.sa_gran + (((size_t)((hn >> 6) * 8 + m) * 3) + sec) * 64 + (hn & 63);
          st_gran(gp, sa_tag, s.x); st_gran(gp + 1, sa_tag, s.y); st_gran(gp + 2, sa_tag, s.z); st_gran(gp + 3, sa_tag, s.w);
        }
        if (n < d) {
          if (!SA) *reinterpret_cast<float4*>(p.q + (size_t)m * d + n) = s;
        } else {
          const bool isk = n < 2 * d;
          f16* dst = (isk ? p.kc : p.vc) + ((size_t)ep_slot * p.ctx + ep_pos) * d + (n - (isk ? d : 2 * d));
          const f16x4 o = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(dst) = o;
        }
      } else {
        if (p.flags & GV_GELU) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
        const size_t o = (size_t)m * p.N + n;
        if (p.flags & GV_RESID) {
          const float4 r = make_float4(ep_res.x + s.x, ep_res.y + s.y, ep_res.z + s.z, ep_res.w + s.w);
          rsum = r;
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = r;
          if (p.y16) { const f16x4 h = {(f16)r.x, (f16)r.y, (f16)r.z, (f16)r.w}; *reinterpret_cast<f16x4*>(p.y16 + o) = h; }   // f16 copy: next layer's folded cross-Q input
        } else if (p.flags & GV_OUT_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = s;
        } else {
          const f16x4 h = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + o) = h;
        }
      }
    }
    // LayerNorm partials of the rows this launch produces (GV_RESID with stat_out: the out-projection in front of the folded cross-attention
    // query): per (row, 16-column tile) the pair (sum, M2 about the tile mean), as gemv_frag_kernel's residual epilogue leaves them - the
    // four lanes of a row (kq = 0..3) hold four columns each; whole waves take part in the shuffles.  The consumer merges d / 16 pairs per
    // row instead of re-reading and re-summing the whole rows in every one of its workgroups.
    if ((p.flags & GV_RESID) && p.stat_out) {
      float t1 = ep_ok ? (rsum.x + rsum.y) + (rsum.z + rsum.w) : 0.f;
      t1 += __shfl_xor(t1, 16); t1 += __shfl_xor(t1, 32);
      float t2 = 0.f;
      if (ep_ok) { const float ml = t1 * 0.0625f, a_ = rsum.x - ml, b_ = rsum.y - ml, c_ = rsum.z - ml, e_ = rsum.w - ml; t2 = (a_ * a_ + b_ * b_) + (c_ * c_ + e_ * e_); }
      t2 += __shfl_xor(t2, 16); t2 += __shfl_xor(t2, 32);
      const int tile = nt * NT + (NT == 2 ? (tid >> 6) & 1 : 0);
      if ((lane >> 4) == 0 && m < M && rows * tile < p.N) *reinterpret_cast<float2*>(p.stat_out + ((size_t)m * (p.N >> 4) + tile) * 2) = make_float2(t1, t2);
    }
  }
  stamp(pf, 6);
  if (tid == 0) tl_end(p.prof);
#undef WIS_ROW_OF
}
// Kernel arguments: what the kernel needs BEFORE its first vector load (activation / weight base addresses, the shape) leads the
// argument list as plain scalars, the rest follows as the struct.  With -amdgpu-kernarg-preload-count (build.py) the command
// processor delivers the leading 16 dwords in SGPRs at wave launch, so the activation and weight requests go out without waiting
// for a scalar load of the kernarg segment (an HBM round trip: the step's 1.6 GB weight stream leaves nothing of it in the
// caches between replays); hipcc preloads only scalar / pointer arguments, never a by-value struct (kernarg_preload_length 0).
#define WIS_GV_LEAD(q) (q).x, (q).x2, (q).Wp, (q).M, (q).N, (q).K, (q).xsplit
#define WIS_GV_LEAD_DECL(s) const void* s##x, const void* s##x2, const f16* s##Wp, int s##M, int s##N, int s##K, int s##xsplit
#define WIS_GV_LEAD_APPLY(q, s) (q).x = s##x; (q).x2 = s##x2; (q).Wp = s##Wp; (q).M = s##M; (q).N = s##N; (q).K = s##K; (q).xsplit = s##xsplit
template <int MB, int MODE, int SC, int RM, bool W8, int NT = 1, bool SA = false>
__global__ __launch_bounds__(256) void gemv_kernel(WIS_GV_LEAD_DECL(l_), int KC, GemvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WIS_GV_LEAD_APPLY(p, l_);
  if (SA && (int)blockIdx.x >= p.sa_first) { sa_consume(p, (int)blockIdx.x - p.sa_first, smem); return; }      // the heads' self-attention workgroups, behind the n-tiles
  gemv_body<MB, MODE, SC, RM, W8, NT, SA>(p, KC, blockIdx.x, smem);
}
// Two skinny GEMMs in ONE launch (workgroups [0, nA) run problem A, the rest problem B; both f16-activation, single-chunk,
// <= 16 rows): the decoder's attention output projection together with the cross-attention query projection folded THROUGH it
// (model.hip fused_out_cq): one dependent stage instead of two.
// (r6, WIS_CA_PREFETCH=1) L2 prefetch riders: the launch leaves 96 of 256 CUs idle, the cross-attention that follows it starts with a 7.7 MB K / V stream
// from HBM.  Workgroup nA + nB + i of this launch touches exactly what workgroup i of the cross-attention will load (key chunk c = i % chunks of head
// h = i / chunks: 32 KB of K fragments, 32 KB of V^T rows) and leaves; with the round-robin placement of linear workgroup ids over the 8 XCDs both land on
// XCD i % 8, i.e. in the L2 the consumer reads through.
struct DualPf { const f16* k; const f16* v; int T, Tpad, chunks, CL, n; };
template <int SCA, int SCB>
__global__ __launch_bounds__(256) void gemv_dual_kernel(int nA, int M, const void* ax, const f16* aWp, const void* bx, const void* bx2, const f16* bWp, int bxsplit,
                                                        GemvP pa, GemvP pb, DualPf pf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (pf.n && (int)blockIdx.x >= pf.n) {
    const int i = (int)blockIdx.x - pf.n, c = i % pf.chunks, h = i / pf.chunks, tid = threadIdx.x;
    const int klo = c * pf.CL;
    const f16* kb = pf.k + (size_t)h * 8 * pf.T * 8;
    const f16* vb = pf.v + (size_t)h * 64 * pf.Tpad;
    u32x4 sink = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // K: 8 planes x CL keys x 16 B
      int key = klo + tid; if (key > pf.T - 1) key = pf.T - 1;
      const u32x4 v = *reinterpret_cast<const u32x4*>(kb + ((size_t)u * pf.T + key) * 8);
      sink[0] ^= v[0]; sink[1] ^= v[1]; sink[2] ^= v[2]; sink[3] ^= v[3];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {      // V^T: 64 rows x CL keys x 2 B = 64 x 32 pieces of 16 B
      const int piece = tid + 256 * u, row = piece >> 5, col = piece & 31;
      const u32x4 v = *reinterpret_cast<const u32x4*>(vb + (size_t)row * pf.Tpad + klo + 8 * col);
      sink[0] ^= v[0]; sink[1] ^= v[1]; sink[2] ^= v[2]; sink[3] ^= v[3];
    }
    asm volatile("" :: "v"(sink[0]), "v"(sink[1]), "v"(sink[2]), "v"(sink[3]));
    return;
  }
  // (K of either problem is the instantiation's: SC k-steps of 32 per wave, four waves)
  if ((int)blockIdx.x < nA) { pa.x = ax; pa.x2 = nullptr; pa.Wp = aWp; pa.M = M; pa.K = SCA * 128; gemv_body<1, 2, SCA, 1, false>(pa, SCA * 128, blockIdx.x, smem); }
  else { pb.x = bx; pb.x2 = bx2; pb.Wp = bWp; pb.M = M; pb.K = SCB * 128; pb.xsplit = bxsplit; gemv_body<1, 2, SCB, 1, false>(pb, SCB * 128, (int)blockIdx.x - nA, smem); }
}
// per-device launch state of the skinny GEMM (several replica worker threads may launch on different GPUs at once): the dynamic-LDS
// ceiling of each device, queried once, and the lock under which the per-instantiation attribute is raised
static std::mutex g_gemv_dev_mu;
static std::atomic<size_t> g_gemv_lds_max[64];
static size_t gemv_lds_limit(int dev) {
  size_t v = g_gemv_lds_max[dev & 63].load(std::memory_order_acquire);
  if (!v) {
    int a = 0;
    if (hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || a < 65536) a = 65536;
    v = (size_t)a;
    g_gemv_lds_max[dev & 63].store(v, std::memory_order_release);
  }
  return v;
}
int launch_gemv_dual(hipStream_t st, const GemvP& pa, const GemvP& pb, const f16* pf_k, const f16* pf_v, int pf_T, int pf_Tpad, int pf_chunks, int pf_H) {
  if (pa.M != pb.M || pa.M < 1 || pa.M > 16 || pa.x2 || pa.wscale || pb.wscale || (pa.flags & (GV_LN | GV_QKV)) || (pb.flags & (GV_LN | GV_QKV))) { set_error("gemv_dual: unsupported pair"); return WIS_E_UNSUPPORTED; }
  if (pa.M * (pa.K / 8) > 13 * 256 || pb.M * (pb.K / 8) > 13 * 256 || pa.K % 128 || pb.K % 128) { set_error("gemv_dual: rows do not fit the register staging (M=%d K=%d/%d)", pa.M, pa.K, pb.K); return WIS_E_UNSUPPORTED; }
  const int sa = pa.K / 128, sb = pb.K / 128;
  const size_t aux = (size_t)4 * 64 * 16 + MAX_ROWS * 8 + 4 * 16 * 4 + 16;
  const int Kmax = pa.K > pb.K ? pa.K : pb.K;
  const size_t lds = (((size_t)pa.M * (Kmax + 8) * 2 + 15) & ~(size_t)15) + aux;
  if (lds > 65536) { set_error("gemv_dual: LDS"); return WIS_E_UNSUPPORTED; }
  const int nA = cdiv(pa.N, 16), nB = cdiv(pb.N, 16);
  DualPf pf; memset(&pf, 0, sizeof(pf));
  if (pf_k && pf_v && pf_chunks > 0) {
    const int CL = cdiv(cdiv(pf_T, pf_chunks), 32) * 32, used = cdiv(pf_T, CL);
    if (CL == 256 && (nA + nB) % 8 == 0) { pf.k = pf_k; pf.v = pf_v; pf.T = pf_T; pf.Tpad = pf_Tpad; pf.chunks = used; pf.CL = CL; pf.n = nA + nB; }
  }
  dim3 grid(nA + nB + (pf.n ? pf.chunks * pf_H : 0)), block(256);
  GemvP a = pa, b = pb; a.rows = 16; b.rows = 16;
  if (sa == 10 && sb == 20) hipLaunchKernelGGL((gemv_dual_kernel<10, 20>), grid, block, lds, st, nA, a.M, a.x, a.Wp, b.x, b.x2, b.Wp, b.xsplit, a, b, pf);
  else if (sa == 8 && sb == 16) hipLaunchKernelGGL((gemv_dual_kernel<8, 16>), grid, block, lds, st, nA, a.M, a.x, a.Wp, b.x, b.x2, b.Wp, b.xsplit, a, b, pf);
  else if (sa == 6 && sb == 12) hipLaunchKernelGGL((gemv_dual_kernel<6, 12>), grid, block, lds, st, nA, a.M, a.x, a.Wp, b.x, b.x2, b.Wp, b.xsplit, a, b, pf);
  else if (sa == 4 && sb == 8) hipLaunchKernelGGL((gemv_dual_kernel<4, 8>), grid, block, lds, st, nA, a.M, a.x, a.Wp, b.x, b.x2, b.Wp, b.xsplit, a, b, pf);
  else if (sa == 3 && sb == 6) hipLaunchKernelGGL((gemv_dual_kernel<3, 6>), grid, block, lds, st, nA, a.M, a.x, a.Wp, b.x, b.x2, b.Wp, b.xsplit, a, b, pf);
  else { set_error("gemv_dual: K=%d/%d not instantiated", pa.K, pb.K); return WIS_E_UNSUPPORTED; }
  return WIS_OK;
}

int launch_gemv(hipStream_t st, const GemvP& p) {
  if (p.M < 1 || p.M > 48 || p.K % 128 || p.N % 4) { set_error("gemv: M=%d N=%d K=%d unsupported (the LDS-staged form holds <= 48 rows; more rows take launch_gemv_frag)", p.M, p.N, p.K); return WIS_E_UNSUPPORTED; }
  const int MB = cdiv(p.M, 16);
  // largest K-chunk (multiple of 128 dividing K) whose f16 image of M rows fits the LDS: 64 KiB for <= 16 rows (several
  // workgroups per CU), the whole 160 KiB CU array (minus slack) for the batched-decode row counts
  int cur_dev = 0;
  if (hipGetDevice(&cur_dev) != hipSuccess) cur_dev = 0;
  const size_t lds_dev_max = gemv_lds_limit(cur_dev);
  // (two half-size chunks so that two workgroups fit a CU were measured slower for N = 4d: 24.3 vs 19.0 us at 40 rows)
  const size_t lds_cap = MB == 1 ? 65536 : (lds_dev_max > 155648 ? 155648 : lds_dev_max);
  int KC = p.K;
  const size_t aux = (size_t)4 * MB * 64 * 16 + MAX_ROWS * 8 + 4 * 16 * 4 + 16;   // red + stats + sred (+ alignment)
  while ((size_t)p.M * (KC + 8) * 2 + aux > lds_cap) {
    int next = 0;
    for (int c = KC - 128; c >= 128; c -= 128) if (p.K % c == 0) { next = c; break; }
    if (!next) { set_error("gemv: cannot chunk K=%d for M=%d", p.K, p.M); return WIS_E_UNSUPPORTED; }
    KC = next;
  }
  const size_t lds = (((size_t)p.M * (KC + 8) * 2 + 15) & ~(size_t)15) + aux;
  const int rows = p.rows ? p.rows : 16;
  GemvP pp = p; pp.rows = rows;
  const int npad = cdiv(p.N, rows) * rows;
  dim3 grid(npad / rows), block(256);
  int mode = 0;
  if (p.flags & GV_LN) {
    // folded LayerNorm: raw fp32 rows in registers - at most 8 rows of at most 2048 columns; callers split larger row counts
    // into layernorm_kernel (no affine) + the f16-activation path (model.hip: launch_ln_gemv)
    if (p.M > 8 || p.K > 2048 || KC != p.K || !p.csum) { set_error("gemv: fused LayerNorm needs M <= 8, K <= 2048 and the folded column sums (M=%d K=%d)", p.M, p.K); return WIS_E_UNSUPPORTED; }
  }
  if (p.flags & GV_LNP) {
    if ((p.flags & GV_LN) || p.M > 16 || p.K > 1280 || p.K % 64 || KC != p.K || !p.csum || !p.stat_in || p.M * (p.K / 8) > 13 * 256) {
      set_error("gemv: the LayerNorm fold from partials needs f16 rows, column sums, partials, M <= 16, K <= 1280 (M=%d K=%d)", p.M, p.K); return WIS_E_UNSUPPORTED; }
  }
  if (p.flags & GV_LN16) {
    if ((p.flags & (GV_LN | GV_LNP)) || p.M > 8 || p.K > 2048 || p.K % 8 || KC != p.K || !p.csum || MB != 1) {
      set_error("gemv: the LayerNorm fold on f16 rows needs column sums, M <= 8, K <= 2048 (M=%d K=%d)", p.M, p.K); return WIS_E_UNSUPPORTED; }
  }
  if (MB == 1) {
    if (KC == p.K) {
      if (p.flags & GV_LN) mode = 1;
      else if (p.flags & GV_LN16) mode = 3;
      else if (!(p.flags & GV_LN) && p.M * (p.K / 8) <= 13 * 256) mode = 2;
    }
  } else if (!(p.flags & GV_LN) && p.M * (KC / 8) <= 30 * 256) mode = 2;       // register-staged f16 chunks (single or multi chunk)
  const int sck = KC / 128;
  const int sc = ((KC == p.K || mode == 2) && (sck == 3 || sck == 4 || sck == 6 || sck == 8 || sck == 10)) ? sck : 0;
#define WIS_GV1(MBv, MODEv, SCv, RMv, W8v) do { \
    if (lds > 65536) {      /* the attribute is per device and per instantiation: set once each, under the per-device state's lock */ \
      static std::atomic<unsigned long long> big_ok{0};      /* bit = device */ \
      if (!((big_ok.load(std::memory_order_acquire) >> (cur_dev & 63)) & 1ull)) { \
        std::lock_guard<std::mutex> lk(g_gemv_dev_mu); \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<MBv, MODEv, SCv, RMv, W8v>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap) != hipSuccess) { \
          set_error("gemv: cannot raise the dynamic LDS limit to %zu bytes", lds_cap); return WIS_E_HIP; } \
        big_ok.fetch_or(1ull << (cur_dev & 63), std::memory_order_release); } } \
    hipLaunchKernelGGL((gemv_kernel<MBv, MODEv, SCv, RMv, W8v>), grid, block, lds, st, WIS_GV_LEAD(pp), KC, pp); } while (0)
#define WIS_GV(MBv, MODEv, SCv, RMv) do { if (p.wscale) WIS_GV1(MBv, MODEv, SCv, RMv, true); else WIS_GV1(MBv, MODEv, SCv, RMv, false); } while (0)
#define WIS_GV_SC(MBv, MODEv, RMv) do { switch (sc) { case 3: WIS_GV(MBv, MODEv, 3, RMv); break; case 4: WIS_GV(MBv, MODEv, 4, RMv); break; case 6: WIS_GV(MBv, MODEv, 6, RMv); break; \
                                                      case 8: WIS_GV(MBv, MODEv, 8, RMv); break; case 10: WIS_GV(MBv, MODEv, 10, RMv); break; default: WIS_GV(MBv, MODEv, 0, RMv); } } while (0)
  // two n-tiles per workgroup (gemv_body NT = 2): the LayerNorm-folded projections of the one-utterance step at K = 1280, 4-5 rows, f16 weights;
  // WIS_GV_NT2 bit 0 = N >= 4096 without the KV-cache epilogue (FFN1: 320 tiles on 256 CUs left a quarter of the CUs with two workgroups;
  // 160 two-tile workgroups: decode step 1.268 -> 1.250 ms), bit 1 = the QKV projection (240 tiles had a CU each already: 120 fat
  // workgroups measured SLOWER, 1.292 ms - off).  Default 1.
  static const int nt2_mask = getenv("WIS_GV_NT2") ? atoi(getenv("WIS_GV_NT2")) : 1;
  if (MB == 1 && mode == 1 && sc == 10 && p.M > 3 && p.M <= 5 && !p.wscale && rows == 16 && p.N % 32 == 0 && !(p.flags & GV_RESID) &&
      (((nt2_mask & 1) && !(p.flags & GV_QKV) && p.N >= 4096) || ((nt2_mask & 2) && (p.flags & GV_QKV)))) {
    hipLaunchKernelGGL((gemv_kernel<1, 1, 20, 5, false, 2>), dim3(p.N / 32), block, lds, st, WIS_GV_LEAD(pp), KC, pp);
    return WIS_OK;
  }
  if (p.sa_gran) {      // QKV projection + the heads' self-attention workgroups in one launch (sa_consume)
    if (!(MB == 1 && mode == 1 && sc > 0 && !p.wscale && (p.flags & GV_QKV) && p.M <= 8 && p.d % 64 == 0 && p.N == 3 * p.d && p.sa_epoch && p.sa_flag && p.sa_out)) {
      set_error("gemv: the fused self-attention needs the LayerNorm-folded f16 QKV projection of <= 8 rows (M=%d N=%d K=%d)", p.M, p.N, p.K); return WIS_E_UNSUPPORTED; }
    pp.sa_first = (int)grid.x;
    const dim3 gsa(grid.x + p.d / 64);
    const size_t lds_sa = lds > 4 * 512 * 4 ? lds : 4 * 512 * 4;      // the consumer workgroups' exchange rows: 2 KiB per wave
#define WIS_GV_SA(SCv, RMv) hipLaunchKernelGGL((gemv_kernel<1, 1, SCv, RMv, false, 1, true>), gsa, block, lds_sa, st, WIS_GV_LEAD(pp), KC, pp)
#define WIS_GV_SA_RM(SCv) do { if (p.M <= 3) WIS_GV_SA(SCv, 3); else if (p.M <= 5) WIS_GV_SA(SCv, 5); else WIS_GV_SA(SCv, 8); } while (0)
    switch (sc) { case 3: WIS_GV_SA_RM(3); break; case 4: WIS_GV_SA_RM(4); break; case 6: WIS_GV_SA_RM(6); break; case 8: WIS_GV_SA_RM(8); break; default: WIS_GV_SA_RM(10); }
#undef WIS_GV_SA_RM
#undef WIS_GV_SA
    return WIS_OK;
  }
  if (MB == 1) {
    if (mode == 1) { if (p.M <= 3) WIS_GV_SC(1, 1, 3); else if (p.M <= 5) WIS_GV_SC(1, 1, 5); else WIS_GV_SC(1, 1, 8); }
    else if (mode == 3) { if (p.M <= 3) WIS_GV_SC(1, 3, 3); else if (p.M <= 5) WIS_GV_SC(1, 3, 5); else WIS_GV_SC(1, 3, 8); }
    else if (mode == 2) {
      // K = 4d of the one-utterance step (FFN2: 80 workgroups x 160 KiB of weights): EVERY fragment requested up front (40 per wave:
      // 160 VGPRs, one workgroup per CU anyway) instead of a ring of 16 - the ring held 64 KiB in flight per CU, and 80 CUs x 64 KiB
      // over the memory latency is what the stream ran at (WIS_FFN2_RING=1: the ring, for A/B)
      static const bool ffn2_ring = getenv("WIS_FFN2_RING") && atoi(getenv("WIS_FFN2_RING")) != 0;
      if (sck == 40 && KC == p.K && !p.wscale && !ffn2_ring) WIS_GV1(1, 2, 40, 1, false);
      else WIS_GV_SC(1, 2, 1);
    } else WIS_GV_SC(1, 0, 1);
  }
  else if (MB == 2) { if (mode == 2) WIS_GV_SC(2, 2, 1); else WIS_GV(2, 0, 0, 1); }
  else { if (mode == 2) WIS_GV_SC(3, 2, 1); else WIS_GV(3, 0, 0, 1); }
#undef WIS_GV_SC
#undef WIS_GV
#undef WIS_GV1
  return WIS_OK;
}

// =======================================================================================
// Batched decode rows (8 < M <= 96): skinny GEMM on activation FRAGMENT images.  grid = Npad/16 workgroups of 4 waves; wave w owns
// the k-steps [w*S, (w+1)*S), S = K/128.  Per k-step a wave issues ONE 1 KiB weight fragment load (HBM, non-temporal) and MB
// 1 KiB activation fragment loads (L2: the image is 40-120 KiB and every workgroup reads it), PF k-steps ahead of the MFMAs -
// no LDS staging of the activations, no staging barrier, 12 KiB of LDS (the cross-wave reduction) so several workgroups share a CU
// (the LDS-staged form held 103 KiB per workgroup: one workgroup per CU, two dispatch rounds for the 320-tile FFN1).
// LayerNorm (GV_LN): the projection is stored folded (W o gamma, b + W.beta, column sums c); mean / rstd of a row come from the
// per-16-column partial sums the producing residual epilogue left in stat_in (summed in a fixed order: deterministic):
//   y = rs * (W'x - mu * c) + b'.
// Residual epilogues (GV_RESID) write the fp32 rows in place, their f16 fragment image for the next projection and the partials.
// A lane's quarter of a row's LayerNorm partials at K = 1280: 20 (sum, M2) pairs = 160 contiguous bytes, taken as TEN 16-byte loads requested together (they were 20
// eight-byte loads - in the two-tile kernel issued four at a time, five dependent rounds in the kernel's tail; a wave request costs the CU's address path the same
// whatever it carries); merged in pair order: the sums come out bit for bit as before.
__device__ __forceinline__ void ln_partials20(const float2* sp, float c, float& s1, float& s2) {
  const float4* sp4 = reinterpret_cast<const float4*>(sp);
  float4 v[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) v[i] = sp4[i];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const float dm0 = v[i].x * 0.0625f - c; s1 += dm0; s2 += v[i].y + 16.0f * dm0 * dm0;
    const float dm1 = v[i].z * 0.0625f - c; s1 += dm1; s2 += v[i].w + 16.0f * dm1 * dm1;
  }
}
// MS (r6): M split - the workgroup computes MB row blocks (from p.mb0) of an image of p.mbi: a projection with few n-tiles (d x d: 80) is run by
// (n-tile, row-block group) workgroups, each pulling ITS rows' activation fragments through its CU instead of the whole image (the phase stamps put 40-50 %
// of such a wave's life into getting the image's requests accepted: profiles/r06_phase_cycles.md); no cross-workgroup reduction - rows are independent.
// The row-block groups of an n-tile read the same weight fragments (ids a multiple of 8 apart: dispatched to the same XCD, whose L2 serves the repeats), so
// the weights are requested with ordinary loads there, not non-temporal ones.
template <int MB, int PF, bool W8, bool MS = false>
__device__ __forceinline__ void gemv_frag_body(const GemvP& p, const int nt, const int KS, const int ksi) {
  typedef typename WFrag<W8>::T WT;
  constexpr int EPN = (MB + 3) / 4;      // row blocks a wave finishes in the epilogue: wave w owns blocks w, w + 4 (up to 96 rows = 6 blocks)
  __shared__ __attribute__((aligned(16))) float red[4 * MB * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // K slices (KS > 1: the K = 4d projection - 80 workgroups would each pull the whole activation image and a 160 KiB weight
  // panel through one CU's ~55 GB/s; see GemvP::ksplit): slice ksi covers the k-steps [ksi, ksi + 1) * ksteps / KS, a quarter per wave
  const int M = p.M, K = p.K, ksteps = K >> 5, S = ksteps / (4 * KS);
  // (wks / wk0: the matrix is a k-step window of a wider packed image - the two halves of the folded cross-Q matrix [W'q | W'q Wo])
  const int wks = p.wks ? p.wks : ksteps;
  // phase stamps of ONE wave (tap builds only; stamp() is empty otherwise): workgroup 0 / slice 0 / thread 0 -
  // 0 start, 1 first PF k-steps requested, 2 stream consumed (MFMA loop), 3 epilogue operands requested + partial sums in LDS + barrier,
  // 4 K-split merge passed (ticket), 5 epilogue done
  unsigned long long* pf = (nt == 0 && ksi == 0 && tid == 0 && (!MS || p.mb0 == 0)) ? p.prof : nullptr;
  stamp(pf, 0);
  const WT* wq = reinterpret_cast<const WT*>(p.Wp) + ((size_t)nt * wks + p.wk0 + (size_t)(ksi * 4 + wave) * S) * 64 + lane;
  const int MBI = MS ? p.mbi : MB, mb0 = MS ? p.mb0 : 0;      // row blocks of the image, this workgroup's first
  const u32x4* xq = reinterpret_cast<const u32x4*>(p.x) + ((size_t)(ksi * 4 + wave) * S * MBI + mb0) * 64 + lane;
  WT a[PF]; u32x4 b[PF][MB];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    if (u < S) {
      a[u] = MS ? wq[(size_t)u * 64] : __builtin_nontemporal_load(wq + (size_t)u * 64);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) b[u][mb] = xq[(size_t)(u * MBI + (MS && mb0 + mb >= MBI ? 0 : mb)) * 64];      // (a ragged last group: a valid block, its sums discarded)
    }
  }
  stamp(pf, 1);
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < S; base += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (base + u < S) {
        const f16x8 av = WFrag<W8>::cvt(a[u]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, *reinterpret_cast<const f16x8*>(&b[u][mb]), acc[mb], 0, 0, 0);
        const int nx = base + u + PF;
        if (nx < S) {
          a[u] = MS ? wq[(size_t)nx * 64] : __builtin_nontemporal_load(wq + (size_t)nx * 64);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) b[u][mb] = xq[(size_t)(nx * MBI + (MS && mb0 + mb >= MBI ? 0 : mb)) * 64];
        }
      }
    }
  }
  stamp(pf, 2);
  // epilogue operands: requested behind the whole stream, covered by the reduction barrier
  const int l15 = lane & 15, kq = lane >> 4;
  const int ep_n = 16 * nt + 4 * kq;
  const bool ln = p.flags & GV_LN;
  const int nq = K >> 6;                                      // (K/16) partial pairs per row, a quarter per lane
  // LayerNorm statistics from the row's per-16-column partials (sum, M2 about the tile's own mean - see the residual epilogue
  // below): pairs merge like Welford / Chan states, in ONE pass about a shift c = the mean of the row's first tile:
  //   mu = c + mean_t(m_t - c),   M2 = sum_t (M2_t + 16 (m_t - c)^2) - K (mu - c)^2
  // so a row whose mean is large against its spread loses nothing to the E[x^2] - mu^2 cancellation (|mu - c| is of the order of
  // the row's own spread; the <= 8-row kernel shifts by x[r][0] for the same reason), and both sums are ready before the reduction
  // barrier (a second pass about the exact mean put a dependent lane exchange + 20 more reads behind it).
  bool ep_act[EPN], ep_ok[EPN]; int ep_m[EPN];
  float4 ep_bias = make_float4(0.f, 0.f, 0.f, 0.f), ep_cs = ep_bias, ep_sc = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 ep_res[EPN]; int ep_slot[EPN], ep_pos[EPN]; float s1[EPN], s2[EPN], sc0[EPN];
  if (ep_n < p.N) {
    if (p.bias) ep_bias = *reinterpret_cast<const float4*>(p.bias + ep_n);
    if (ln) ep_cs = *reinterpret_cast<const float4*>(p.csum + ep_n);
    if (W8) ep_sc = *reinterpret_cast<const float4*>(p.wscale + ep_n);
  }
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    const int mb = wave + 4 * e;
    ep_act[e] = mb < MB && (!MS || mb0 + mb < MBI); ep_m[e] = (mb0 + mb) * 16 + l15; ep_ok[e] = ep_act[e] && ep_m[e] < M && ep_n < p.N;
    ep_res[e] = make_float4(0.f, 0.f, 0.f, 0.f); ep_slot[e] = 0; ep_pos[e] = 0; s1[e] = 0.f; s2[e] = 0.f; sc0[e] = 0.f;
    if (ep_act[e]) {
      const int mm = ep_m[e] < M ? ep_m[e] : M - 1;
      if (ln) {       // this lane sums a quarter of the row's partials (rows >= M: a clamped duplicate, discarded)
        const float2* row = reinterpret_cast<const float2*>(p.stat_in) + (size_t)mm * (K >> 4);
        const float2* sp = row + (size_t)kq * nq;
        const float c = row[0].x * 0.0625f;
        sc0[e] = c;
        if (PF == 10) {      // K = 1280: the lane's 20 pairs, all requested at once
          ln_partials20(sp, c, s1[e], s2[e]);
        } else {
#pragma unroll 4
          for (int i = 0; i < nq; ++i) { const float2 v = sp[i]; const float dm = v.x * 0.0625f - c; s1[e] += dm; s2[e] += v.y + 16.0f * dm * dm; }
        }
      }
      if (ep_ok[e]) {
        if (p.flags & GV_QKV) { if (ep_n >= p.d) { ep_slot[e] = p.slot[ep_m[e]]; ep_pos[e] = p.pos[ep_m[e]]; } }
        else if (p.flags & GV_RESID) ep_res[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.y) + (size_t)ep_m[e] * p.N + ep_n);
      }
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    *reinterpret_cast<float4*>(red + ((size_t)(wave * MB + mb) * 64 + lane) * 4) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  __syncthreads();
  stamp(pf, 3);
  if (KS > 1) {
    // publish this slice's sums (write-through, agent scope: the other slices of the n-tile run on other XCDs), take a ticket; the
    // last arriver re-arms the ticket, drops its stale L1 lines and goes on to add the slices in index order (the order never
    // depends on who arrived last: bit-reproducible) - the hand-off of the cross-attention's ticket form
#pragma unroll
    for (int e = 0; e < EPN; ++e) {
      if (!ep_act[e]) continue;
      const int ep_mb = wave + 4 * e;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)(w * MB + ep_mb) * 64 + lane) * 4);
        s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
      }
      float* dst = p.kpart + ((((size_t)nt * KS + ksi) * MB + ep_mb) * 64 + lane) * 4;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(s) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(p.kcnt + nt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)(KS - 1);
      if (last) {
        __hip_atomic_store(p.kcnt + nt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_last = last;
    }
    __syncthreads();
    stamp(pf, 4);
    if (!s_last) return;
  }
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    if (!ep_act[e]) { if (e == EPN - 1) stamp(pf, 5); continue; }                                   // whole waves: a row block belongs to one wave
    const int ep_mb = wave + 4 * e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KS > 1) {
      for (int k = 0; k < KS; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(p.kpart + ((((size_t)nt * KS + k) * MB + ep_mb) * 64 + lane) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    } else {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)(w * MB + ep_mb) * 64 + lane) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
    if (W8) { s.x *= ep_sc.x; s.y *= ep_sc.y; s.z *= ep_sc.z; s.w *= ep_sc.w; }
    if (ln) {     // the four lanes of a row (kq = 0..3) hold a quarter of its sums each: ((q0 + q1) + (q2 + q3)) on every lane
      float t1 = s1[e], t2 = s2[e];
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      const float invK = 1.0f / (float)K;
      const float dmu = t1 * 16.0f * invK, mu = sc0[e] + dmu;      // K / 16 tiles
      const float rs = 1.0f / sqrtf(fmaxf(t2 * invK - dmu * dmu, 0.f) + 1e-5f);
      s.x = rs * (s.x - mu * ep_cs.x); s.y = rs * (s.y - mu * ep_cs.y); s.z = rs * (s.z - mu * ep_cs.z); s.w = rs * (s.w - mu * ep_cs.w);
    }
    s.x += ep_bias.x; s.y += ep_bias.y; s.z += ep_bias.z; s.w += ep_bias.w;
    const int m = ep_m[e], n = ep_n;
    if (p.flags & GV_QKV) {
      if (ep_ok[e]) {
        const int d = p.d;
        if (n < d) {
          *reinterpret_cast<float4*>(p.q + (size_t)m * d + n) = s;
        } else {
          const bool isk = n < 2 * d;
          f16* dst = (isk ? p.kc : p.vc) + ((size_t)ep_slot[e] * p.ctx + ep_pos[e]) * d + (n - (isk ? d : 2 * d));
          const f16x4 o = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(dst) = o;
        }
      }
      continue;
    }
    if (p.flags & GV_GELU) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
    if (p.flags & GV_RESID) {
      const float4 r = make_float4(ep_res[e].x + s.x, ep_res[e].y + s.y, ep_res[e].z + s.z, ep_res[e].w + s.w);
      float t1 = 0.f, t2 = 0.f;
      if (ep_ok[e]) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.N + n) = r;
        if (p.y_xf) { const f16x4 h = {(f16)r.x, (f16)r.y, (f16)r.z, (f16)r.w}; *reinterpret_cast<f16x4*>(p.y_xf + xf_index(m, n, p.ymb)) = h; }
        t1 = (r.x + r.y) + (r.z + r.w);
      }
      if (p.stat_out) {      // partials of this workgroup's 16 columns of row m: (sum, M2 about the tile mean); all four kq lanes take part in the shuffles
        t1 += __shfl_xor(t1, 16);
        t1 += __shfl_xor(t1, 32);
        if (ep_ok[e]) { const float ml = t1 * 0.0625f, a_ = r.x - ml, b_ = r.y - ml, c_ = r.z - ml, e_ = r.w - ml; t2 = (a_ * a_ + b_ * b_) + (c_ * c_ + e_ * e_); }
        t2 += __shfl_xor(t2, 16);
        t2 += __shfl_xor(t2, 32);
        if (kq == 0 && m < M && 16 * nt < p.N) *reinterpret_cast<float2*>(p.stat_out + ((size_t)m * (p.N >> 4) + nt) * 2) = make_float2(t1, t2);
      }
      continue;
    }
    if (!ep_ok[e]) continue;
    if (p.flags & GV_OUT_F32) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.N + n) = s;
    } else {
      const f16x4 h = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
      f16* yo = reinterpret_cast<f16*>(p.y);
      *reinterpret_cast<f16x4*>(yo + (p.ymb ? xf_index(m, n, p.ymb) : (size_t)m * p.N + n)) = h;
    }
  }
  stamp(pf, 6);
}

template <int MB, int PF, bool W8>
__global__ __launch_bounds__(256) void gemv_frag_kernel(const void* l_x, const f16* l_Wp, int l_M, int l_N, int l_K, int l_wks, int l_wk0, int KS, GemvP p) {
  p.x = l_x; p.Wp = l_Wp; p.M = l_M; p.N = l_N; p.K = l_K; p.wks = l_wks; p.wk0 = l_wk0;      // (KS = gridDim.y, passed: the hidden grid-size arguments are a kernarg load too)
  gemv_frag_body<MB, PF, W8>(p, blockIdx.x, KS, blockIdx.y);
}
template <int MB, int PF, bool W8>      // the M-split form: grid (n-tiles, 1, row-block groups), MB row blocks per workgroup of an image of mbi
__global__ __launch_bounds__(256) void gemv_frag_ms_kernel(const void* l_x, const f16* l_Wp, int l_M, int l_N, int l_K, int l_wks, int l_wk0, int mbi, GemvP p) {
  p.x = l_x; p.Wp = l_Wp; p.M = l_M; p.N = l_N; p.K = l_K; p.wks = l_wks; p.wk0 = l_wk0; p.mbi = mbi; p.mb0 = blockIdx.z * MB;
  gemv_frag_body<MB, PF, W8, true>(p, blockIdx.x, 1, 0);
}
// Up to three skinny GEMMs of one row count in ONE launch (f16 weights, no K split): workgroups [0, n0) run problem 0, the next n1
// problem 1, the rest problem 2 - a dependent stage less per decoder layer at 9-96 rows (model.hip dec_forward_frag): the self-attention
// output projection together with the two halves of the cross-attention query folded THROUGH it (q_raw = W'q x0 + (W'q Wo) a + W'q bo:
// one half reads the layer input's fragment image, the other the attention output's; the cross-attention kernel adds the halves and
// applies the LayerNorm statistics of the rows the out-projection produces in this same launch).
struct GemvP3 { GemvP p[3]; int n0, n1; };
// (r6: what a workgroup's first requests need - which problem it is, that problem's activation and weight bases, the common shape and the k-step window -
// leads the argument list as 14 dwords of scalars, so the command processor preloads it into SGPRs (build.py: -amdgpu-kernarg-preload-count); the by-value
// struct alone had a preload length of 0: every workgroup began with a kernarg fetch)
template <int MB, int PF>
__global__ __launch_bounds__(256) void gemv_frag3_kernel(int n0, int n1, const void* x0, const void* x1, const void* x2, const f16* w0, const f16* w1, const f16* w2, GemvP3 ps) {
  const int b = blockIdx.x;
  const int sel = b < n0 ? 0 : (b < n0 + n1 ? 1 : 2);
  const int nt = sel == 0 ? b : (sel == 1 ? b - n0 : b - n0 - n1);
  GemvP p = ps.p[sel];
  p.x = sel == 0 ? x0 : (sel == 1 ? x1 : x2);
  p.Wp = sel == 0 ? w0 : (sel == 1 ? w1 : w2);
  gemv_frag_body<MB, PF, false>(p, nt, 1, 0);
}

// The same skinny GEMM with TWO n-tiles (32 output columns) per workgroup, for the LayerNorm-folded projections that have more
// n-tiles than the chip has CUs (FFN1: 320, vocabulary: 3242): a workgroup's ingest is its weight panel PLUS the whole activation
// image (41 + 123 KB at 8 utterances), a CU takes in ~55 GB/s, and with 320 one-tile workgroups 64 CUs get two of them - the
// kernel's time is those CUs' 328 KB.  Two tiles per workgroup: 160 workgroups x (82 + 123) KB, one round.  Every activation
// fragment feeds two MFMAs.  No residual form, no K split (those projections have 80 n-tiles).
template <int MB, int PF, bool W8>
__global__ __launch_bounds__(256) void gemv_frag2_kernel(WIS_GV_LEAD_DECL(l_), GemvP p) {
  WIS_GV_LEAD_APPLY(p, l_);
  typedef typename WFrag<W8>::T WT;
  constexpr int EPN = (MB + 3) / 4;
  __shared__ __attribute__((aligned(16))) float red[4 * MB * 2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt0 = 2 * blockIdx.x;
  const int M = p.M, K = p.K, ksteps = K >> 5, S = ksteps >> 2;
  const WT* wq0 = reinterpret_cast<const WT*>(p.Wp) + ((size_t)nt0 * ksteps + (size_t)wave * S) * 64 + lane;
  const WT* wq1 = wq0 + (size_t)ksteps * 64;
  const u32x4* xq = reinterpret_cast<const u32x4*>(p.x) + (size_t)wave * S * MB * 64 + lane;
  unsigned long long* pf = (blockIdx.x == 0 && tid == 0) ? p.prof : nullptr;      // (phase stamps as in gemv_frag_body; tap builds only)
  stamp(pf, 0);
  WT a0[PF], a1[PF]; u32x4 b[PF][MB];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    if (u < S) {
      a0[u] = __builtin_nontemporal_load(wq0 + (size_t)u * 64);
      a1[u] = __builtin_nontemporal_load(wq1 + (size_t)u * 64);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) b[u][mb] = xq[(size_t)(u * MB + mb) * 64];
    }
  }
  stamp(pf, 1);
  f32x4 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int base = 0; base < S; base += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (base + u < S) {
        const f16x8 av0 = WFrag<W8>::cvt(a0[u]), av1 = WFrag<W8>::cvt(a1[u]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const f16x8 xb = *reinterpret_cast<const f16x8*>(&b[u][mb]);
          acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av0, xb, acc[mb][0], 0, 0, 0);
          acc[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av1, xb, acc[mb][1], 0, 0, 0);
        }
        const int nx = base + u + PF;
        if (nx < S) {
          a0[u] = __builtin_nontemporal_load(wq0 + (size_t)nx * 64);
          a1[u] = __builtin_nontemporal_load(wq1 + (size_t)nx * 64);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) b[u][mb] = xq[(size_t)(nx * MB + mb) * 64];
        }
      }
    }
  }
  stamp(pf, 2);
  // epilogue operands (requested behind the whole stream, covered by the reduction barrier): per n-tile the bias / column sums /
  // row scales, per owned row block the LayerNorm statistics (one pass about the first tile's mean, as in gemv_frag_kernel)
  const int l15 = lane & 15, kq = lane >> 4;
  const int nq = K >> 6;
  float4 ep_bias[2], ep_cs[2], ep_sc[2];
  int ep_n[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    ep_n[nb] = 16 * (nt0 + nb) + 4 * kq;
    ep_bias[nb] = make_float4(0.f, 0.f, 0.f, 0.f); ep_cs[nb] = ep_bias[nb]; ep_sc[nb] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ep_n[nb] < p.N) {
      if (p.bias) ep_bias[nb] = *reinterpret_cast<const float4*>(p.bias + ep_n[nb]);
      ep_cs[nb] = *reinterpret_cast<const float4*>(p.csum + ep_n[nb]);
      if (W8) ep_sc[nb] = *reinterpret_cast<const float4*>(p.wscale + ep_n[nb]);
    }
  }
  bool ep_act[EPN]; int ep_m[EPN], ep_slot[EPN], ep_pos[EPN]; float s1[EPN], s2[EPN], sc0[EPN];
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    const int mb = wave + 4 * e;
    ep_act[e] = mb < MB; ep_m[e] = mb * 16 + l15; ep_slot[e] = 0; ep_pos[e] = 0; s1[e] = 0.f; s2[e] = 0.f; sc0[e] = 0.f;
    if (ep_act[e]) {
      const int mm = ep_m[e] < M ? ep_m[e] : M - 1;
      const float2* row = reinterpret_cast<const float2*>(p.stat_in) + (size_t)mm * (K >> 4);
      const float2* sp = row + (size_t)kq * nq;
      const float c = row[0].x * 0.0625f;
      sc0[e] = c;
      if (nq == 20) ln_partials20(sp, c, s1[e], s2[e]);      // (uniform: K = 1280)
      else {
#pragma unroll 4
        for (int i = 0; i < nq; ++i) { const float2 v = sp[i]; const float dm = v.x * 0.0625f - c; s1[e] += dm; s2[e] += v.y + 16.0f * dm * dm; }
      }
      if ((p.flags & GV_QKV) && ep_m[e] < M) { ep_slot[e] = p.slot[ep_m[e]]; ep_pos[e] = p.pos[ep_m[e]]; }
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
      *reinterpret_cast<float4*>(red + ((size_t)((wave * MB + mb) * 2 + nb) * 64 + lane) * 4) = make_float4(acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]);
  __syncthreads();
  stamp(pf, 3);
  stamp(pf, 4);
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    if (!ep_act[e]) continue;                                   // whole waves: a row block belongs to one wave
    const int ep_mb = wave + 4 * e, m = ep_m[e];
    float t1 = s1[e], t2 = s2[e];
    t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
    t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
    const float invK = 1.0f / (float)K;
    const float dmu = t1 * 16.0f * invK, mu = sc0[e] + dmu;
    const float rs = 1.0f / sqrtf(fmaxf(t2 * invK - dmu * dmu, 0.f) + 1e-5f);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)((w * MB + ep_mb) * 2 + nb) * 64 + lane) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      if (W8) { s.x *= ep_sc[nb].x; s.y *= ep_sc[nb].y; s.z *= ep_sc[nb].z; s.w *= ep_sc[nb].w; }
      s.x = rs * (s.x - mu * ep_cs[nb].x) + ep_bias[nb].x; s.y = rs * (s.y - mu * ep_cs[nb].y) + ep_bias[nb].y;
      s.z = rs * (s.z - mu * ep_cs[nb].z) + ep_bias[nb].z; s.w = rs * (s.w - mu * ep_cs[nb].w) + ep_bias[nb].w;
      const int n = ep_n[nb];
      if (m >= M || n >= p.N) continue;
      if (p.flags & GV_QKV) {
        const int d = p.d;
        if (n < d) {
          *reinterpret_cast<float4*>(p.q + (size_t)m * d + n) = s;
        } else {
          const bool isk = n < 2 * d;
          f16* dst = (isk ? p.kc : p.vc) + ((size_t)ep_slot[e] * p.ctx + ep_pos[e]) * d + (n - (isk ? d : 2 * d));
          const f16x4 o = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(dst) = o;
        }
        continue;
      }
      if (p.flags & GV_GELU) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
      if (p.flags & GV_OUT_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.N + n) = s;
      } else {
        const f16x4 h = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
        f16* yo = reinterpret_cast<f16*>(p.y);
        *reinterpret_cast<f16x4*>(yo + (p.ymb ? xf_index(m, n, p.ymb) : (size_t)m * p.N + n)) = h;
      }
    }
  }
  stamp(pf, 5);
  stamp(pf, 6);
}

int launch_gemv_frag3(hipStream_t st, const GemvP* p, int n) {
  if (n < 2 || n > 3) { set_error("gemv_frag3: %d problems", n); return WIS_E_ARG; }
  GemvP3 ps; memset(&ps, 0, sizeof(ps));
  int nts[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const GemvP& g = p[i];
    if (g.M != p[0].M || g.xmb != p[0].xmb || g.K != p[0].K || g.M < 1 || g.M > MAX_ROWS || g.K % 128 || g.N % 16 || g.xmb != cdiv(g.M, 16) || g.wscale || g.ksplit > 1 ||
        (g.flags & (GV_LN | GV_QKV))) { set_error("gemv_frag3: problem %d unsupported (M=%d N=%d K=%d flags=%d)", i, g.M, g.N, g.K, g.flags); return WIS_E_UNSUPPORTED; }
    ps.p[i] = g; nts[i] = g.N / 16;
  }
  if (n == 2) ps.p[2] = ps.p[1];
  ps.n0 = nts[0]; ps.n1 = nts[1];
  dim3 grid(nts[0] + nts[1] + nts[2]), block(256);
  const bool s10 = p[0].K == 1280;
#define WIS_GF3_ARGS ps.n0, ps.n1, ps.p[0].x, ps.p[1].x, ps.p[2].x, ps.p[0].Wp, ps.p[1].Wp, ps.p[2].Wp, ps
#define WIS_GF3(MBv, PFA, PFB) do { if (s10) hipLaunchKernelGGL((gemv_frag3_kernel<MBv, PFA>), grid, block, 0, st, WIS_GF3_ARGS); else hipLaunchKernelGGL((gemv_frag3_kernel<MBv, PFB>), grid, block, 0, st, WIS_GF3_ARGS); } while (0)
  switch (p[0].xmb) {
    case 1: WIS_GF3(1, 10, 8); break;
    case 2: WIS_GF3(2, 10, 8); break;
    case 3: WIS_GF3(3, 10, 8); break;
    case 4: WIS_GF3(4, 6, 6); break;
    case 5: WIS_GF3(5, 6, 6); break;
    case 6: WIS_GF3(6, 6, 6); break;
    default: set_error("gemv_frag3: %d row blocks unsupported", p[0].xmb); return WIS_E_UNSUPPORTED;
  }
#undef WIS_GF3
#undef WIS_GF3_ARGS
  return WIS_OK;
}

int launch_gemv_frag(hipStream_t st, const GemvP& p) {
  if (p.M < 1 || p.M > MAX_ROWS || p.K % 128 || p.N % 4 || p.xmb != cdiv(p.M, 16)) { set_error("gemv_frag: M=%d N=%d K=%d xmb=%d unsupported", p.M, p.N, p.K, p.xmb); return WIS_E_UNSUPPORTED; }
  if ((p.flags & GV_LN) && (!p.csum || !p.stat_in)) { set_error("gemv_frag: the folded LayerNorm needs column sums and row partials"); return WIS_E_ARG; }
  if ((p.flags & GV_RESID) && (p.N % 16)) { set_error("gemv_frag: residual rows need N %% 16 == 0"); return WIS_E_UNSUPPORTED; }
  const int npad = cdiv(p.N, 16) * 16;
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  if (ks > 1 && (!p.kpart || !p.kcnt || (p.K / 32) % (4 * ks) || (p.flags & GV_LN))) { set_error("gemv_frag: K split %d unsupported (K=%d)", ks, p.K); return WIS_E_UNSUPPORTED; }
  // two n-tiles per workgroup where there are more n-tiles than CUs (LayerNorm-folded projections: FFN1, vocabulary): -1.7 ... -3.5 % of
  // the decode time at 8 / 16 utterances.  ON since round 4 (WIS_FRAG_NB=1 keeps one tile per workgroup: A/B switch).  Round 3 kept it
  // off for a sporadic corruption of its five-row-block instantiation (168 VGPRs: three waves per SIMD) - features 12 and 14 of a tile
  // wrong for a whole row block.  Cause, narrowed down in round 4 (tools/frag2_lab.hip, tools/frag_stress.hip): hipcc's SLP pass had
  // turned the epilogue's f32 arithmetic into packed-f32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32); the
  // LOW halves of those packed results (float4 components x and z = features 12 and 14 of the 4-feature group a lane owns) come
  // out wrong in lanes 48-63 when three such waves share a SIMD - 14-22 of every 40 launches, 10 808 of 96 000 with four launches
  // in flight, on EVERY box the SLP build was run on (round 4 wrote "some chips": its clean runs were of other builds; corrected in
  // round 5); the same source built with -fno-slp-vectorize (scalar v_add / v_fma: same 168 registers, same
  // occupancy) is clean (0 of 240 000 launches), as is any form with two waves per SIMD.  The whole library is therefore compiled
  // without SLP vectorisation (build.py HIP_FLAGS), tools/isa_lint.py fails the build on any MFMA kernel that contains packed-f32
  // arithmetic, and tests/test_gpu_stress.py bit-compares 10^4 launches of every shipped instantiation on four streams with the
  // idle-GPU launch.
  static const int env_nb = getenv("WIS_FRAG_NB") ? atoi(getenv("WIS_FRAG_NB")) : 2;
  if (env_nb == 2 && (p.flags & GV_LN) && !(p.flags & GV_RESID) && ks == 1 && (npad / 16) % 2 == 0 && npad / 16 > 256) {
    dim3 g2(npad / 32), blk(256);
#define WIS_GF2(MBv, PFv) do { if (p.wscale) hipLaunchKernelGGL((gemv_frag2_kernel<MBv, PFv, true>), g2, blk, 0, st, WIS_GV_LEAD(p), p); \
                               else hipLaunchKernelGGL((gemv_frag2_kernel<MBv, PFv, false>), g2, blk, 0, st, WIS_GV_LEAD(p), p); } while (0)
    switch (p.xmb) {
      case 1: WIS_GF2(1, 6); break;
      case 2: WIS_GF2(2, 6); break;
      case 3: WIS_GF2(3, 6); break;
      case 4: WIS_GF2(4, 4); break;
      case 5: WIS_GF2(5, 4); break;
      case 6: WIS_GF2(6, 4); break;
      default: set_error("gemv_frag: %d row blocks unsupported", p.xmb); return WIS_E_UNSUPPORTED;
    }
#undef WIS_GF2
    return WIS_OK;
  }
  // M split (r6): few n-tiles (<= 85: the d x d projections), no K split, 2-6 row blocks in equal groups: (n-tile, group) workgroups of 1 or 2 row blocks
  // (WIS_FRAG_MSPLIT=0: one workgroup per n-tile, A/B switch)
  static const bool env_ms = !(getenv("WIS_FRAG_MSPLIT") && atoi(getenv("WIS_FRAG_MSPLIT")) == 0);
  // ... INSTEAD of a K split where one was asked for (the K = 4d projection, 80 n-tiles: two K slices were 160 workgroups of 80 KB of weights + HALF the
  // image each and a ticket merge; three row-block groups are 240 workgroups of 160 KB + a THIRD of the image, no merge: 1.996 -> 1.936 ms per step at 8 utterances)
  const int ms_groups = p.xmb <= 1 ? 0 : (p.xmb <= 3 ? p.xmb : (p.xmb + 1) / 2);      // groups of one row block up to 48 rows, of two beyond (the last may be ragged)
  if (env_ms && ms_groups && npad / 16 <= 85) {
    dim3 gm(npad / 16, 1, ms_groups), blk(256);
    // (a 20- / 12-deep ring for the K = 5120 streams - 160 KB requested per CU instead of 64 - measured 1.930 / 1.937 vs 1.937 / 1.938 ms per step at 8 utterances
    // and no better at 12 / 16, session r6K: what a CU takes in is capped near 64 GB/s whatever is asked for; the eight-deep ring stays)
    const bool s10m = p.K == 1280;
#define WIS_GFM(MBv) do { \
    if (p.wscale) { if (s10m) hipLaunchKernelGGL((gemv_frag_ms_kernel<MBv, 10, true>), gm, blk, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, p.xmb, p); else hipLaunchKernelGGL((gemv_frag_ms_kernel<MBv, 8, true>), gm, blk, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, p.xmb, p); } \
    else { if (s10m) hipLaunchKernelGGL((gemv_frag_ms_kernel<MBv, 10, false>), gm, blk, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, p.xmb, p); else hipLaunchKernelGGL((gemv_frag_ms_kernel<MBv, 8, false>), gm, blk, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, p.xmb, p); } } while (0)
    if (p.xmb <= 3) WIS_GFM(1); else WIS_GFM(2);
#undef WIS_GFM
    return WIS_OK;
  }
  dim3 grid(npad / 16, ks), block(256);
  const bool s10 = p.K / ks == 1280;     // ten k-steps per wave: the whole stream of a wave is requested up front
  // up to three row blocks: the whole wave stream (ten k-steps) or eight k-steps in flight; four to six row blocks (49-96 rows):
  // a six-deep ring, so that weight + activation fragments stay inside the register file ((MB + 1) x 4 VGPRs per k-step)
#define WIS_GF(MBv, PFA, PFB) do { \
    if (p.wscale) { if (s10) hipLaunchKernelGGL((gemv_frag_kernel<MBv, PFA, true>), grid, block, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, (int)grid.y, p); else hipLaunchKernelGGL((gemv_frag_kernel<MBv, PFB, true>), grid, block, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, (int)grid.y, p); } \
    else { if (s10) hipLaunchKernelGGL((gemv_frag_kernel<MBv, PFA, false>), grid, block, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, (int)grid.y, p); else hipLaunchKernelGGL((gemv_frag_kernel<MBv, PFB, false>), grid, block, 0, st, p.x, p.Wp, p.M, p.N, p.K, p.wks, p.wk0, (int)grid.y, p); } } while (0)
  // (a twelve-deep ring for the K = 4d slices at 33-48 rows, 218 VGPRs: 2.138 vs 2.134 ms per step at 8 utterances - that kernel is bound by
  // its activation fragments from L2, not by bytes in flight; the one-utterance FFN2, bound by exactly that, takes launch_gemv's SC = 40 form)
  switch (p.xmb) {
    case 1: WIS_GF(1, 10, 8); break;
    case 2: WIS_GF(2, 10, 8); break;
    case 3: WIS_GF(3, 10, 8); break;
    case 4: WIS_GF(4, 6, 6); break;
    case 5: WIS_GF(5, 6, 6); break;
    case 6: WIS_GF(6, 6, 6); break;
    default: set_error("gemv_frag: %d row blocks unsupported", p.xmb); return WIS_E_UNSUPPORTED;
  }
#undef WIS_GF
  return WIS_OK;
}

// rows (fp32 or f16, row-major) -> fragment image + per-16-column partial sums.  grid M, block 256: a thread owns 4 consecutive
// columns; the partials of a 16-column tile are the sums over 4 adjacent threads, in the same (a+b)+(c+d) order everywhere.
__device__ __forceinline__ void xf_emit4(float4 v, int m, int col, f16* xf, float* stat, int K, int MB) {
  if (xf) { const f16x4 h = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w}; *reinterpret_cast<f16x4*>(xf + xf_index(m, col, MB)) = h; }
  if (stat) {      // (sum, M2 about the tile mean) of the 16-column tile: the pair format gemv_frag_kernel merges
    float t1 = (v.x + v.y) + (v.z + v.w);
    t1 += dpp_f<0xB1>(t1);       // quad: lanes 4j .. 4j+3 = the four column quads of one 16-column tile
    t1 += dpp_f<0x4E>(t1);
    const float ml = t1 * 0.0625f, a = v.x - ml, b = v.y - ml, c = v.z - ml, e = v.w - ml;
    float t2 = (a * a + b * b) + (c * c + e * e);
    t2 += dpp_f<0xB1>(t2);
    t2 += dpp_f<0x4E>(t2);
    if ((threadIdx.x & 3) == 0) *reinterpret_cast<float2*>(stat + ((size_t)m * (K >> 4) + (col >> 4)) * 2) = make_float2(t1, t2);
  }
}
__global__ __launch_bounds__(256) void xf_pack_kernel(const void* __restrict__ x, int x_f16, f16* __restrict__ xf, float* __restrict__ stat, int K, int MB) {
  const int m = blockIdx.x;
  for (int c4 = threadIdx.x; c4 < (K >> 2); c4 += 256) {     // K % 16 == 0: whole quads of threads stay together
    float4 v;
    if (x_f16) { const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(x) + (size_t)m * K + 4 * c4); v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
    else v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)m * K + 4 * c4);
    xf_emit4(v, m, 4 * c4, xf, stat, K, MB);
  }
}
int launch_xf_pack(hipStream_t st, const void* x, int x_f16, f16* xf, float* stat, int M, int K, int MB) {
  if (K % 16 || MB < cdiv(M, 16)) { set_error("xf_pack: K=%d MB=%d unsupported", K, MB); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(xf_pack_kernel, dim3(M), dim3(256), 0, st, x, x_f16, xf, stat, K, MB);
  return WIS_OK;
}
__global__ __launch_bounds__(256) void dec_embed_xf_kernel(const f16* __restrict__ emb, const f16* __restrict__ pos_emb, const int* __restrict__ tok,
                                                           const int* __restrict__ pos, float* __restrict__ x, f16* __restrict__ xf, float* __restrict__ stat, int d, int MB) {
  const int m = blockIdx.x;
  const f16* e = emb + (size_t)tok[m] * d;
  const f16* pe = pos_emb + (size_t)pos[m] * d;
  for (int c4 = threadIdx.x; c4 < (d >> 2); c4 += 256) {
    const f16x4 a = *reinterpret_cast<const f16x4*>(e + 4 * c4), b = *reinterpret_cast<const f16x4*>(pe + 4 * c4);
    const float4 v = make_float4((float)a[0] + (float)b[0], (float)a[1] + (float)b[1], (float)a[2] + (float)b[2], (float)a[3] + (float)b[3]);
    *reinterpret_cast<float4*>(x + (size_t)m * d + 4 * c4) = v;
    xf_emit4(v, m, 4 * c4, xf, stat, d, MB);
  }
}
int launch_dec_embed_xf(hipStream_t st, const f16* emb, const f16* pos_emb, const int* tok, const int* pos, float* x, f16* xf, float* stat, int M, int d, int MB) {
  hipLaunchKernelGGL(dec_embed_xf_kernel, dim3(M), dim3(256), 0, st, emb, pos_emb, tok, pos, x, xf, stat, d, MB);
  return WIS_OK;
}

// =======================================================================================
// x[m] = E[tok[m]] + pos_emb[pos[m]]   (no embedding scale; learned positions) -> fp32
__global__ void dec_embed_kernel(const f16* __restrict__ emb, const f16* __restrict__ pos_emb, const int* __restrict__ tok,
                                 const int* __restrict__ pos, float* __restrict__ x, f16* __restrict__ xh, int d) {
  const int m = blockIdx.x;
  const f16* e = emb + (size_t)tok[m] * d;
  const f16* pe = pos_emb + (size_t)pos[m] * d;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    const float v = (float)e[i] + (float)pe[i];
    x[(size_t)m * d + i] = v;
    if (xh) xh[(size_t)m * d + i] = (f16)v;          // f16 copy of the layer input (fused out-proj + cross-Q stage)
  }
}
// the same with the rows' LayerNorm partials (GV_LNP consumers): thread = one 16-column tile of row m
__global__ void dec_embed_stat_kernel(const f16* __restrict__ emb, const f16* __restrict__ pos_emb, const int* __restrict__ tok,
                                      const int* __restrict__ pos, float* __restrict__ x, f16* __restrict__ xh, float* __restrict__ stat, int d) {
  const int m = blockIdx.x, t = threadIdx.x, nt = d >> 4;
  if (t >= nt) return;
  const f16x8* e = reinterpret_cast<const f16x8*>(emb + (size_t)tok[m] * d + 16 * t);
  const f16x8* pe = reinterpret_cast<const f16x8*>(pos_emb + (size_t)pos[m] * d + 16 * t);
  const f16x8 e0 = e[0], e1 = e[1], p0 = pe[0], p1 = pe[1];
  float v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = (float)e0[i] + (float)p0[i]; v[8 + i] = (float)e1[i] + (float)p1[i]; }
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s1 += v[i];
  const float ml = s1 * 0.0625f;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const float a = v[i] - ml; s2 += a * a; }
  float4* xo = reinterpret_cast<float4*>(x + (size_t)m * d + 16 * t);
#pragma unroll
  for (int i = 0; i < 4; ++i) xo[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  f16x8 h0, h1;
#pragma unroll
  for (int i = 0; i < 8; ++i) { h0[i] = (f16)v[i]; h1[i] = (f16)v[8 + i]; }
  f16x8* ho = reinterpret_cast<f16x8*>(xh + (size_t)m * d + 16 * t);
  ho[0] = h0; ho[1] = h1;
  *reinterpret_cast<float2*>(stat + ((size_t)m * nt + t) * 2) = make_float2(s1, s2);
}
int launch_dec_embed(hipStream_t st, const f16* emb, const f16* pos_emb, const int* tok, const int* pos, float* x, int M, int d, f16* xh, float* stat) {
  if (stat && xh && d % 16 == 0 && d / 16 <= 256) hipLaunchKernelGGL(dec_embed_stat_kernel, dim3(M), dim3(((d / 16 + 63) / 64) * 64), 0, st, emb, pos_emb, tok, pos, x, xh, stat, d);
  else hipLaunchKernelGGL(dec_embed_kernel, dim3(M), dim3(256), 0, st, emb, pos_emb, tok, pos, x, xh, d);
  return WIS_OK;
}

// =======================================================================================
// causal self-attention of one new token per row over its cached history.  grid (M, H), block 64.
// lane = (pl = position slot 0..7, c = 16-byte chunk 0..7 of the 64-wide head): one wave instruction moves 8 cache
// rows x 128 B; K and V of the first 64 positions are all issued up front (one round trip), the 8-lane dot
// products and the position reductions are DPP ops, longer histories continue with an online softmax.
// Logical slot of row m = (m / rpu) * sstride + (m % rpu) * rmul (decode rows own their slot, prefill rows share
// the utterance's first slot).
// (argument order: everything the first round of loads needs - q, the caches, the row table, the shape - sits in the first 14
// dwords, which -amdgpu-kernarg-preload-count delivers in SGPRs at wave launch; `out` and the taps are only read at the end)
// TREE (draft verification of a beam search, model.hip verify_beam_draft): the rows of the pass are nodes of a beam tree - row m's history is not
// one slot but a path: positions < w0 in slot anc[m * aw], position w0 + t in slot anc[m * aw + t] (each node's K / V sits in the slot of the beam
// that produced it).  One dependent load more than the plain form, on a path that runs once per 16 steps.
// NB (r6): 8-position blocks requested up front and per pass of the loop: 8 (64 positions), or 2 / 4 when the caller KNOWS the histories are short (model.hip picks the
// step graph by the step index: every row of a decode pass has the same length) - the kernel always asks for 8 NB positions of K and V per (row, head), whatever the
// length (the addresses must not wait for it), so at 8 utterances its 800 waves put 13 600 wave requests through the CUs' address paths for histories of ~20 positions;
// any NB is correct for any length (the loop goes on in steps of 8 NB positions)
template <bool TREE, int NB>
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const float* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc,
                                                           const int* __restrict__ pos, int d, int ctx, int rpu, int sstride, int rmul,
                                                           f16* __restrict__ out, unsigned long long* prof, int out_mb,
                                                           const int* __restrict__ anc, int w0, int aw, const int* __restrict__ base) {
  __shared__ float red[4][64];
  const int m = blockIdx.x, h = blockIdx.y, lane = threadIdx.x, pl = lane >> 3, c = lane & 7;
  unsigned long long* pf = (m == 0 && h == 0 && lane == 0) ? prof : nullptr;
  if (lane == 0) tl_begin(prof);
  stamp(pf, 0);
  // the row's history lives in ITS OWN slot (kv_reorder_kernel made it so after the last beam step): the K / V addresses of the
  // first 64 positions depend on nothing that has to be loaded, so q, the row's length and all of K and V travel in ONE round
  // trip (positions >= len are fetched from valid memory and masked below)
  // (TREE: positions before the window live in slot base[m] - or, without a base table, in the slot of the row's window-step-0 ancestor)
  const int ls = TREE ? (base ? base[m] : anc[(size_t)m * aw]) : (m / rpu) * sstride + (m % rpu) * rmul;
  const int len = pos[m] + 1;
  // slot that holds position p of this row's history
  auto slot_of = [&](int p) -> int {
    if (!TREE) return ls;
    int t = p - w0; t = t < aw - 1 ? t : aw - 1;
    return t < 0 ? ls : anc[(size_t)m * aw + t];
  };
  const float4 q0 = *reinterpret_cast<const float4*>(q + (size_t)m * d + h * 64 + 8 * c);
  const float4 q1 = *reinterpret_cast<const float4*>(q + (size_t)m * d + h * 64 + 8 * c + 4);
  const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
  const int hoff = h * 64 + 8 * c;
  const f16* krow = kc + (size_t)ls * ctx * d + hoff;
  const f16* vrow = vc + (size_t)ls * ctx * d + hoff;
  u32x4 kr[NB], vr[NB];
  if (!TREE) {
#pragma unroll
    for (int i = 0; i < NB; ++i) kr[i] = *reinterpret_cast<const u32x4*>(krow + (size_t)(8 * i + pl) * d);      // ctx >= 64: in bounds
#pragma unroll
    for (int i = 0; i < NB; ++i) vr[i] = *reinterpret_cast<const u32x4*>(vrow + (size_t)(8 * i + pl) * d);
  } else {
    int sl[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) sl[i] = slot_of(8 * i + pl);
#pragma unroll
    for (int i = 0; i < NB; ++i) kr[i] = *reinterpret_cast<const u32x4*>(kc + ((size_t)sl[i] * ctx + 8 * i + pl) * d + hoff);
#pragma unroll
    for (int i = 0; i < NB; ++i) vr[i] = *reinterpret_cast<const u32x4*>(vc + ((size_t)sl[i] * ctx + 8 * i + pl) * d + hoff);
  }
  stamp(pf, 1);
  float m_run = -INFINITY, l_run = 0.f;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // (do-while: a `for (p0 = 0; p0 < len; ...)` head makes hipcc wait for the row length - a scalar load, one full round trip -
  // BEFORE it issues the q / K / V loads above; the first block is unconditional, len >= 1)
  int p0 = 0;
  do {
    if (p0 > 0) {   // histories beyond 8 NB positions: next block (not prefetched).  Unconditional loads from a clamped position (masked
                    // below): guarded per position, hipcc serialised the sixteen loads into eight wait-for-the-last-pair round trips
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p = p0 + 8 * i + pl, pc = p < len ? p : len - 1;
        if (!TREE) {
          kr[i] = *reinterpret_cast<const u32x4*>(krow + (size_t)pc * d);
          vr[i] = *reinterpret_cast<const u32x4*>(vrow + (size_t)pc * d);
        } else {
          const int sl = slot_of(pc);
          kr[i] = *reinterpret_cast<const u32x4*>(kc + ((size_t)sl * ctx + pc) * d + hoff);
          vr[i] = *reinterpret_cast<const u32x4*>(vc + ((size_t)sl * ctx + pc) * d + hoff);
        }
      }
    }
    float sc[NB]; float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = p0 + 8 * i + pl;
      float dot = 0.f;
      if (p < len) {
        const f16x8 kv = *reinterpret_cast<const f16x8*>(&kr[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) dot = fmaf((float)kv[j], qv[j], dot);
      }
      dot += dpp_f<0xB1>(dot); dot += dpp_f<0x4E>(dot); dot += dpp_f<0x141>(dot);   // sum over the 8 chunk lanes
      sc[i] = (p < len) ? dot : -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
    mx = wave_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float pw = __expf(sc[i] - m_new);     // 0 for masked positions
      lsum += pw;
      if (p0 + 8 * i + pl < len) {
        const f16x8 vv = *reinterpret_cast<const f16x8*>(&vr[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(pw, (float)vv[j], acc[j]);
      }
    }
    l_run = l_run * alpha + wave_sum(lsum) * 0.125f;   // every position is replicated on its 8 chunk lanes
    m_run = m_new;
    p0 += 8 * NB;
  } while (p0 < len);
  stamp(pf, 2);
  // reduce acc over the 8 position slots: xor 8 inside the 16-lane row by DPP, the four rows through LDS
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += dpp_f<0x128>(acc[j]);     // row_ror:8
  if ((lane & 8) == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[lane >> 4][8 * c + j] = acc[j];
  }
  __syncthreads();
  stamp(pf, 3);
  const float o = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  out[out_mb ? xf_index(m, h * 64 + lane, out_mb) : (size_t)m * d + h * 64 + lane] = (f16)(o / l_run);
  stamp(pf, 4);
  if (lane == 0) tl_end(prof);
}
int launch_dec_self_attn(hipStream_t st, const float* q, const f16* kc, const f16* vc, const int* pos, f16* out,
                         int M, int H, int d, int ctx, int rpu, int sstride, int rmul, unsigned long long* prof, int out_mb,
                         const int* anc, int w0, int aw, const int* base, int nb) {
  if (ctx > 512 || ctx < 64) { set_error("dec_self_attn: ctx=%d outside [64, 512]", ctx); return WIS_E_UNSUPPORTED; }
  if (anc) {
    if (aw < 1) { set_error("dec_self_attn: ancestor table of width %d", aw); return WIS_E_ARG; }
    hipLaunchKernelGGL((dec_self_attn_kernel<true, 8>), dim3(M, H), dim3(64), 0, st, q, kc, vc, pos, d, ctx, rpu, sstride, rmul, out, prof, out_mb, anc, w0, aw, base);
  } else if (nb == 2)
    hipLaunchKernelGGL((dec_self_attn_kernel<false, 2>), dim3(M, H), dim3(64), 0, st, q, kc, vc, pos, d, ctx, rpu, sstride, rmul, out, prof, out_mb, anc, w0, aw, base);
  else if (nb == 4)
    hipLaunchKernelGGL((dec_self_attn_kernel<false, 4>), dim3(M, H), dim3(64), 0, st, q, kc, vc, pos, d, ctx, rpu, sstride, rmul, out, prof, out_mb, anc, w0, aw, base);
  else
    hipLaunchKernelGGL((dec_self_attn_kernel<false, 8>), dim3(M, H), dim3(64), 0, st, q, kc, vc, pos, d, ctx, rpu, sstride, rmul, out, prof, out_mb, anc, w0, aw, base);
  return WIS_OK;
}

// =======================================================================================
// cross-attention of the R (<= 16) query rows of one utterance over a chunk of <= 256 encoder keys, on the
// matrix cores.  grid (chunks, H, B), block 256 = 4 waves.
//   scores  S^T[key][r] = K[key][:] . Q[r][:]      v_mfma_f32_16x16x32_f16, A = K (16 keys x 32 dh, 16-byte loads from
//                                                   the [H][dh/8][T][8] image), B = Q^T (registers); wave w owns key tiles w, w+4, ..
//   output  O^T[dh][r]  = V^T[dh][:] . P[r][:]      A = V^T (16 dh x 32 keys, 16-byte loads from the [H][64][Tpad] image),
//                                                   B = P^T (f16 in LDS); wave w owns dh 16w..16w+15
// Q, then every K and V fragment, are issued up front (one memory round trip; vmcnt retires in order, so the tiny Q load
// is first).  No cross-lane shuffles: the MFMA output layout puts a query row on a lane column.  Chunk partials
// (o[64], max, sum) are published with write-through (sc1) relaxed agent-scope stores and combined by the last-arriving
// workgroup with sc1 loads: placement independent, no fences (guide §6 G16).
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
constexpr int CA_PSTR = 264;   // f16 row pitch of the P image (256 keys + 8: 16-byte aligned, bank-skewed)

// TPW = key tiles per wave (CL / 64): 2 for 128-key chunks, 4 for 256-key chunks; loads are unconditional with clamped
// addresses (masked by `kl < n` below), so the issue phase is straight-line code.
// CM = the most chunks the combine handles (its partial buffers are register arrays of that size): 6 for the default 256-key
// chunking, 16 otherwise.  With [16] arrays at every chunking the kernel needed 218 VGPRs - two workgroups per CU, so the 960
// workgroups of an 8-utterance batch ran in two rounds; with CM = 6 it needs 88 (five per CU).
// FOLD: the folded-query prologue (B = 1-sized row counts only; its own instantiation so that the batched kernel keeps its
// register budget).
// SPIN (small grids: B * H <= 192, <= 6 chunks, <= 8 rows): the chunk partials travel as 8-byte {tag, value} GRANULES, one write-through
// store each (guide G16 form R2: "the data is the flag") - a producing workgroup stores and leaves: no drain, no barrier, no ticket.
// The workgroup of the LAST chunk of every (utterance, head) is its combiner: it keeps its own partial in LDS and re-reads the other
// chunks' granules (relaxed agent-scope loads) until every tag equals this launch's epoch, then combines in the fixed chunk order
// (the same arithmetic, in the same order, as the ticket form: bit-identical results).  The epoch of an (utterance, head) is a word
// in device memory that its combiner advances at the end of the launch (launches on a stream are ordered, so the next launch
// reads the new value; tags only grow, so a slot's old contents can never match) - nothing is reset between launches, and graph
// replays stay valid.  Progress: producers never wait; at most B * H <= 192 workgroups spin, fewer than the chip's 256 CUs, so a
// producer always finds a slot.  The spin is bounded: on exhaustion the combiner raises a flag (checked by wis_generate) instead of
// hanging.  What it removes from the hand-off: the store drain, the returning ticket atomic and the agent-scope acquire (1.7 us).

// FOLD: 0 = q is the finished query; 1 = folded query, statistics reduced from the rows themselves (xres = x1 fp32 [B*R][d]: the one-utterance
// step); 2 = folded query of the BATCHED step: q_raw arrives as two halves (q + q2: W'q x0 + W'q bo and (W'q Wo) a, model.hip
// dec_forward_frag) and the statistics come from the per-16-column (sum, M2) partials the out-projection's residual epilogue left
// (xres = [B*R][d/16][2]; merged like gemv_frag_kernel merges them) - 80 pairs per row instead of 1280 floats; 3 = the same with V requested
// up front (small grids: the one-utterance step); 4 (r5) = 2 with the query's operands handed out through LDS AND V up front (see STG below).
template <int TPW, int CM, int FOLD, bool SPIN>
// (argument order as in dec_self_attn_kernel: the first 14 dwords - q, K, V^T, the folded query's rows / second half, the epoch block and
// the packed shape - are preloaded into SGPRs, so the kernel's first round of requests (q, the statistics' rows, the K fragments) does
// not wait for a kernarg fetch; what needs the later arguments - column sums / bias of the folded query, V^T's pitch - is requested behind them)
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const float* __restrict__ q, const f16* __restrict__ kx, const f16* __restrict__ vt,
                                                             const float* __restrict__ xres, const float* __restrict__ q2, unsigned* epoch, int RHCC, int dT,
                                                             int TpadF, f16* __restrict__ out, float* part, unsigned* counters,
                                                             unsigned long long* prof, int out_mb,
                                                             const float* __restrict__ qcs, const float* __restrict__ qb, gran_t* gran) {
  // rows | heads << 8 | chunk length << 14 | chunks << 24, d | T << 16: six pointers + two words = the 14 preloaded dwords
  const int R = RHCC & 0xFF, H = (RHCC >> 8) & 0x3F, CL = (RHCC >> 14) & 0x3FF, C = (RHCC >> 24) & 0xFF, d = dT & 0xFFFF, T = (dT >> 16) & 0xFFFF;
  __shared__ float ssc[16][257];
  __shared__ __attribute__((aligned(16))) f16 sp16[16 * CA_PSTR];
  __shared__ float smax[16][16];
  __shared__ float ssumw[4][16];
  __shared__ float smx[16], ssum[16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  // bit 30 of the V^T pitch argument: the row groups b = 0 .. B-1 are rows of ONE utterance (draft verification: up to 96 tree rows
  // against utterance 0's K / V); queries, outputs, partials and tickets stay per group
  const int Tpad = TpadF & 0x3FFFFFFF, bkv = (TpadF >> 30) ? 0 : b;
  const int klo = c * CL, n = (klo + CL <= T) ? CL : T - klo;   // 1 <= n <= 256, klo % 32 == 0
  // (tap builds: workgroup (0, 0, 0) stamps entries 0-5 of its row, the LAST chunk's workgroup - the combiner of the granule form - entries 7-13)
  unsigned long long* pf = (h == 0 && b == 0 && tid == 0 && prof) ? (c == 0 ? prof : (c == C - 1 ? prof + 7 : nullptr)) : nullptr;
  if (tid == 0) tl_begin(prof);
  stamp(pf, 0);

  // ---- loads: Q (B operand, lane = (row r, k-quarter)), K fragments, V^T fragments
  const int rq = l15 < R ? l15 : R - 1;
  const float* qp = q + (size_t)(b * R + rq) * d + h * 64 + 8 * kq;
  // STG (FOLD 4, the batched fold with everything requested up front): the query rows (both halves), the folded query's column sums and
  // its bias - (2 R + 2) x 64 floats that every wave would fetch for itself into 64 VGPRs, sixteen requests per wave - are fetched once
  // per wave as one 16-byte piece per lane and array (clamped, unconditional) and handed out through LDS behind the prologue's barrier.
  // What that buys is REGISTERS: the operands no longer sit in VGPRs beside the K fragments while the requests are in flight, so the V
  // fragments can be requested up front as well and the kernel still fits four workgroups per CU.
  constexpr bool STG = FOLD == 4;
  float4 qa0_, qa1_, qb0_, qb1_, stq, stq2, stc;
  qa0_ = qa1_ = qb0_ = qb1_ = stq = stq2 = stc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!STG) {
    qa0_ = *reinterpret_cast<const float4*>(qp); qa1_ = *reinterpret_cast<const float4*>(qp + 4);
    qb0_ = *reinterpret_cast<const float4*>(qp + 32); qb1_ = *reinterpret_cast<const float4*>(qp + 36);
  } else {
    const int sr = (tid >> 4) < R ? (tid >> 4) : R - 1;      // thread t < 16 R owns piece (t & 15) of row t >> 4
    const size_t so = (size_t)(b * R + sr) * d + h * 64 + 4 * (tid & 15);
    stq = *reinterpret_cast<const float4*>(q + so);
    if (q2) stq2 = *reinterpret_cast<const float4*>(q2 + so);      // (uniform)
  }
  unsigned ep_now = 0;
  if (SPIN) ep_now = __hip_atomic_load(epoch + 1 + b * H + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // requested with everything else (word 0 is the flag)
  // Folded query (model.hip fused_out_cq): `q` holds q_raw = W'x0 + (W'Wo) a + W'bo of the LayerNorm-folded cross-attention query
  // projection, computed one stage early from the layer input and the self-attention output; the LayerNorm statistics belong
  // to the rows the out-projection has produced SINCE (xres = x1, fp32 [B*R][d]).  Every workgroup reduces its utterance's R
  // rows itself (R x d floats from L2, requested together with K and V) and finishes q = rs (q_raw - mu c) + b'.
  // wave w reduces rows w and w + 4 of the utterance (R <= 8): 5 float4 per lane and row cover d <= 1280
  constexpr bool PSTAT = FOLD >= 2;
  constexpr int NXS = FOLD == 1 ? 5 : 1;
  float4 xs4[2][NXS]; float4 cs0, cs1, cs2, cs3, bq0, bq1, bq2, bq3;
  float4 qc0, qc1, qc2, qc3; float2 pt[2][2];
  const int d4 = d >> 2, ntile = d >> 4;
  if (PSTAT) {
    qc0 = qc1 = qc2 = qc3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!STG && q2) {      // (uniform: the batched step's second half of q_raw; the one-utterance step's dual launch leaves the whole of it in q)
      const float* qp2 = q2 + (size_t)(b * R + rq) * d + h * 64 + 8 * kq;
      qc0 = *reinterpret_cast<const float4*>(qp2); qc1 = *reinterpret_cast<const float4*>(qp2 + 4); qc2 = *reinterpret_cast<const float4*>(qp2 + 32); qc3 = *reinterpret_cast<const float4*>(qp2 + 36);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {      // wave w merges the partials of rows w and w + 4: lane = tile (and tile + 64; d <= 2048)
      const int r = wave + 4 * j;
      const float2* sp = reinterpret_cast<const float2*>(xres) + (size_t)(b * R + (r < R ? r : R - 1)) * ntile;
      pt[j][0] = sp[lane < ntile ? lane : ntile - 1];
      pt[j][1] = sp[lane + 64 < ntile ? lane + 64 : ntile - 1];
    }
  }
  if (FOLD == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave + 4 * j;
      const float4* xr = reinterpret_cast<const float4*>(xres) + (size_t)(b * R + (r < R ? r : R - 1)) * d4;
#pragma unroll
      for (int i = 0; i < NXS; ++i) { const int c4 = lane + 64 * i; xs4[j][i] = xr[c4 < d4 ? c4 : d4 - 1]; }      // unconditional (clamped; masked in the sums): no exec-masked branch between the loads
    }
  }
  // (K fragments: addressed from preloaded arguments only, so they go out before the first wait for a scalar kernarg load)
  const f16* kb = kx + (size_t)(bkv * H + h) * 8 * T * 8;
  u32x4 kf[TPW][2];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int key = klo + 16 * (wave + 4 * i) + l15; if (key > T - 1) key = T - 1;      // clamped; masked below
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[i][ks] = *reinterpret_cast<const u32x4*>(kb + ((size_t)(kq + 4 * ks) * T + key) * 8);
  }
  if (STG) {      // threads 0-15 of every wave: the head's 64 column sums, 16-31: its bias
    stc = *reinterpret_cast<const float4*>(((tid & 16) ? qb : qcs) + h * 64 + 4 * (tid & 15));
  }
  if (FOLD && !STG) {
    const float* cp = qcs + h * 64 + 8 * kq; const float* bp = qb + h * 64 + 8 * kq;
    cs0 = *reinterpret_cast<const float4*>(cp); cs1 = *reinterpret_cast<const float4*>(cp + 4); cs2 = *reinterpret_cast<const float4*>(cp + 32); cs3 = *reinterpret_cast<const float4*>(cp + 36);
    bq0 = *reinterpret_cast<const float4*>(bp); bq1 = *reinterpret_cast<const float4*>(bp + 4); bq2 = *reinterpret_cast<const float4*>(bp + 32); bq3 = *reinterpret_cast<const float4*>(bp + 36);
  }
  constexpr int NSTEP = 2 * TPW;                      // 32-key P.V steps per chunk; V^T is zero padded up to Tpad >= chunks * CL
  const f16* vb = vt + ((size_t)(bkv * H + h) * 64 + 16 * wave + l15) * Tpad + klo + 8 * kq;
  u32x4 vf[NSTEP];
  // (batched fold: the V fragments are requested BEHIND the query prologue - its column sums, biases, second q half and partials
  // are dead by then, so the kernel stays near the 88 registers of the plain form (five workgroups per CU: the 960 workgroups of an
  // 8-utterance batch in one round) instead of 153 (three per CU); V is not needed before the softmax)
  // FOLD 2 / 3: statistics from row partials; 2 requests V BEHIND the prologue (large grids: registers decide how many workgroups a CU
  // holds), 3 with everything else (the one-utterance step's 120 workgroups have CUs to spare)
  constexpr bool VLATE = FOLD == 2;
  if (!VLATE) {
#pragma unroll
    for (int sidx = 0; sidx < NSTEP; ++sidx) vf[sidx] = *reinterpret_cast<const u32x4*>(vb + 32 * sidx);
  }
  stamp(pf, 1);

  float4 qa0 = qa0_, qa1 = qa1_, qb0 = qb0_, qb1 = qb1_;
  if (FOLD) {
    // mean / rstd of the rows: a DPP wave reduction per row (fixed order: bit-reproducible), published through LDS behind a RAW
    // barrier - `__syncthreads()` would first wait for every outstanding load, i.e. for the K / V fragments this prologue is
    // supposed to run in the shadow of (measured with three `__syncthreads()` here: cross-attention 9.0 -> 12.1 us)
    __shared__ float srow[16][2];
    __builtin_amdgcn_sched_barrier(0);      // every load of the kernel is requested above this line: the statistics wait for their rows only
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave + 4 * j;
      float a1 = 0.f, a2 = 0.f;
      if (PSTAT) {
        // Chan merge of the (sum, M2 about the tile mean) pairs about c0 = the first tile's mean (gemv_frag_kernel's one-pass form):
        //   mu = c0 + mean_t (m_t - c0),   var = (sum_t (M2_t + 16 (m_t - c0)^2)) / d - (mu - c0)^2
        const float c0 = readlane_f(pt[j][0].x, 0) * 0.0625f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool in = lane + 64 * i < ntile;
          const float dm = in ? pt[j][i].x * 0.0625f - c0 : 0.f;
          a1 += dm; a2 += in ? pt[j][i].y + 16.0f * dm * dm : 0.f;
        }
        a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0 && r < R) { const float invd = 1.0f / (float)d, dmu = a1 * 16.0f * invd; srow[r][0] = c0 + dmu; srow[r][1] = 1.0f / sqrtf(fmaxf(a2 * invd - dmu * dmu, 0.f) + 1e-5f); }
      } else {
      // ONE pass, shifted by the row's first element: no E[x^2] - mu^2 cancellation for rows with a large common offset, and both
      // wave reductions issue together (a second pass over the registers about the exact mean measured +2.3 us per launch)
      const float c0 = readlane_f(xs4[j][0].x, 0);
#pragma unroll
      for (int i = 0; i < NXS; ++i) {
        const float4 v = xs4[j][i];
        const bool in = lane + 64 * i < d4;
        const float a = in ? v.x - c0 : 0.f, b = in ? v.y - c0 : 0.f, c = in ? v.z - c0 : 0.f, e = in ? v.w - c0 : 0.f;
        a1 += (a + b) + (c + e); a2 += (a * a + b * b) + (c * c + e * e);
      }
      a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (lane == 0 && r < R) { const float ms = a1 / (float)d; srow[r][0] = c0 + ms; srow[r][1] = 1.0f / sqrtf(fmaxf(a2 / (float)d - ms * ms, 0.f) + 1e-5f); }
      }
    }
    constexpr int SQP = 68;      // row pitch of the staged operands (floats): 16-byte aligned, rows four banks apart
    __shared__ __attribute__((aligned(16))) float sq[STG ? 10 * SQP : 4];      // rows 0..7: q_raw (both halves added), 8: column sums, 9: bias
    if (STG) {
      if (tid < 16 * R) *reinterpret_cast<float4*>(&sq[(tid >> 4) * SQP + 4 * (tid & 15)]) = make_float4(stq.x + stq2.x, stq.y + stq2.y, stq.z + stq2.z, stq.w + stq2.w);
      if (tid < 32) *reinterpret_cast<float4*>(&sq[(8 + (tid >> 4)) * SQP + 4 * (tid & 15)]) = stc;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float mu = srow[rq][0], rs = srow[rq][1];
    if (STG) {
      const float* sqr = &sq[rq * SQP + 8 * kq]; const float* scs = &sq[8 * SQP + 8 * kq]; const float* sbq = &sq[9 * SQP + 8 * kq];
      qa0 = *reinterpret_cast<const float4*>(sqr); qa1 = *reinterpret_cast<const float4*>(sqr + 4); qb0 = *reinterpret_cast<const float4*>(sqr + 32); qb1 = *reinterpret_cast<const float4*>(sqr + 36);
      cs0 = *reinterpret_cast<const float4*>(scs); cs1 = *reinterpret_cast<const float4*>(scs + 4); cs2 = *reinterpret_cast<const float4*>(scs + 32); cs3 = *reinterpret_cast<const float4*>(scs + 36);
      bq0 = *reinterpret_cast<const float4*>(sbq); bq1 = *reinterpret_cast<const float4*>(sbq + 4); bq2 = *reinterpret_cast<const float4*>(sbq + 32); bq3 = *reinterpret_cast<const float4*>(sbq + 36);
    }
    if (PSTAT && !STG) {      // the two halves of q_raw (when there are two; the staged form added them on the way into LDS)
      qa0 = make_float4(qa0.x + qc0.x, qa0.y + qc0.y, qa0.z + qc0.z, qa0.w + qc0.w); qa1 = make_float4(qa1.x + qc1.x, qa1.y + qc1.y, qa1.z + qc1.z, qa1.w + qc1.w);
      qb0 = make_float4(qb0.x + qc2.x, qb0.y + qc2.y, qb0.z + qc2.z, qb0.w + qc2.w); qb1 = make_float4(qb1.x + qc3.x, qb1.y + qc3.y, qb1.z + qc3.z, qb1.w + qc3.w);
    }
    qa0 = make_float4(rs * (qa0.x - mu * cs0.x) + bq0.x, rs * (qa0.y - mu * cs0.y) + bq0.y, rs * (qa0.z - mu * cs0.z) + bq0.z, rs * (qa0.w - mu * cs0.w) + bq0.w);
    qa1 = make_float4(rs * (qa1.x - mu * cs1.x) + bq1.x, rs * (qa1.y - mu * cs1.y) + bq1.y, rs * (qa1.z - mu * cs1.z) + bq1.z, rs * (qa1.w - mu * cs1.w) + bq1.w);
    qb0 = make_float4(rs * (qb0.x - mu * cs2.x) + bq2.x, rs * (qb0.y - mu * cs2.y) + bq2.y, rs * (qb0.z - mu * cs2.z) + bq2.z, rs * (qb0.w - mu * cs2.w) + bq2.w);
    qb1 = make_float4(rs * (qb1.x - mu * cs3.x) + bq3.x, rs * (qb1.y - mu * cs3.y) + bq3.y, rs * (qb1.z - mu * cs3.z) + bq3.z, rs * (qb1.w - mu * cs3.w) + bq3.w);
  }
  if (VLATE) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sidx = 0; sidx < NSTEP; ++sidx) vf[sidx] = *reinterpret_cast<const u32x4*>(vb + 32 * sidx);
  }
  f16x8 qf0, qf1;
  qf0[0] = (f16)qa0.x; qf0[1] = (f16)qa0.y; qf0[2] = (f16)qa0.z; qf0[3] = (f16)qa0.w; qf0[4] = (f16)qa1.x; qf0[5] = (f16)qa1.y; qf0[6] = (f16)qa1.z; qf0[7] = (f16)qa1.w;
  qf1[0] = (f16)qb0.x; qf1[1] = (f16)qb0.y; qf1[2] = (f16)qb0.z; qf1[3] = (f16)qb0.w; qf1[4] = (f16)qb1.x; qf1[5] = (f16)qb1.y; qf1[6] = (f16)qb1.z; qf1[7] = (f16)qb1.w;

  // ---- scores: D[i = key][j = r]: lane holds r = l15, keys 16t + 4kq + reg
  float lmax = -INFINITY;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave + 4 * i;
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&kf[i][0]), qf0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&kf[i][1]), qf1, acc, 0, 0, 0);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int kl = 16 * t + 4 * kq + r4;
        const float v = (kl < n) ? acc[r4] : -INFINITY;
        ssc[l15][kl] = v;
        lmax = fmaxf(lmax, v);
      }
    }
  }
  smax[wave * 4 + kq][l15] = lmax;
  __syncthreads();
  stamp(pf, 2);
  if (tid < 16) {
    float mxv = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) mxv = fmaxf(mxv, smax[i][tid]);
    smx[tid] = mxv;
  }
  __syncthreads();
  // ---- P = exp(S - max) as f16 (thread = key), row sums
  {
    const int kl = tid;
    const bool valid = kl < 64 * TPW;
    float e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      e[r] = 0.f;
      if (r < R) {
        if (valid) e[r] = __expf(ssc[r][kl] - smx[r]);
        sp16[r * CA_PSTR + kl] = (f16)e[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) if (r < R) { const float v = wave_sum(e[r]); if (lane == 0) ssumw[wave][r] = v; }
  }
  __syncthreads();
  stamp(pf, 3);
  if (tid < 16) ssum[tid] = (ssumw[0][tid] + ssumw[1][tid]) + (ssumw[2][tid] + ssumw[3][tid]);

  // ---- O^T[dh][r]: wave w owns dh 16w .. 16w+15; D[i = dh][j = r]: lane holds r = l15, dh = 16w + 4kq + reg
  f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sidx = 0; sidx < NSTEP; ++sidx) {
    {
      const f16x8 pb = *reinterpret_cast<const f16x8*>(&sp16[rq * CA_PSTR + 32 * sidx + 8 * kq]);
      oacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&vf[sidx]), pb, oacc, 0, 0, 0);
    }
  }
  __syncthreads();   // ssum visible
  stamp(pf, 4);
  const int dh0 = 16 * wave + 4 * kq;
  if (C == 1) {
    if (l15 < R) {
      const float inv = 1.0f / ssum[l15];
      const f16x4 o = {(f16)(oacc[0] * inv), (f16)(oacc[1] * inv), (f16)(oacc[2] * inv), (f16)(oacc[3] * inv)};
      *reinterpret_cast<f16x4*>(out + (out_mb ? xf_index(b * R + l15, h * 64 + dh0, out_mb) : (size_t)(b * R + l15) * d + h * 64 + dh0)) = o;
    }
    return;
  }
  if (SPIN) {
    const unsigned tag = ep_now + 1u;
    // slot of (utterance-head, chunk c, row r): fixed strides (6 chunks x 8 rows) - the same words for every R / C / batch size, so a tag
    // in a slot was always written under THIS (utterance, head)'s epoch and can never alias another one's (advisor, round 3)
    gran_t* gbase = gran + (size_t)(b * H + h) * 6 * 8 * 66;
    if (c != C - 1) {      // producer: one granule per value, no drain, no ticket
      if (l15 < R) {
        gran_t* gp = gbase + ((size_t)c * 8 + l15) * 66;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) st_gran(gp + dh0 + r4, tag, oacc[r4]);
        if (wave == 0 && kq == 0) { st_gran(gp + 64, tag, smx[l15]); st_gran(gp + 65, tag, ssum[l15]); }
      }
      stamp(pf, 5);
      if (tid == 0) tl_end(prof);
      return;
    }
    // combiner: own partial through LDS (the score buffer is free), the other chunks' from their granules
    if (l15 < R) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) ssc[l15][dh0 + r4] = oacc[r4];
      if (wave == 0 && kq == 0) { ssc[l15][64] = smx[l15]; ssc[l15][65] = ssum[l15]; }
    }
    __syncthreads();
    stamp(pf, 5);
    for (int item = tid; item < R * 32; item += 256) {
      const int r = item >> 5, dp = item & 31;
      float2 ml[CM], ov[CM];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int cc = 0; cc < CM - 1; ++cc) {
          if (cc < C - 1) {
            const gran_t* gp = gbase + ((size_t)cc * 8 + r) * 66;
            const gran_t g0 = ld_gran(gp + 64), g1 = ld_gran(gp + 65), g2 = ld_gran(gp + 2 * dp), g3 = ld_gran(gp + 2 * dp + 1);
            ok = ok & ((unsigned)(g0 >> 32) == tag) & ((unsigned)(g1 >> 32) == tag) & ((unsigned)(g2 >> 32) == tag) & ((unsigned)(g3 >> 32) == tag);
            ml[cc] = make_float2(__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1));
            ov[cc] = make_float2(__uint_as_float((unsigned)g2), __uint_as_float((unsigned)g3));
          }
        }
        if (ok) break;
        if (++spins > CA_SPIN_LIMIT) { atomicOr(epoch, 1u); break; }      // word 0: the give-up flag
        __builtin_amdgcn_s_sleep(1);
      }
      float M_ = ssc[r][64];
#pragma unroll
      for (int cc = 0; cc < CM - 1; ++cc) if (cc < C - 1) M_ = fmaxf(M_, ml[cc].x);
      float L = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
      for (int cc = 0; cc < CM - 1; ++cc) {
        if (cc < C - 1) {
          const float w = __expf(ml[cc].x - M_);
          L = fmaf(ml[cc].y, w, L); O0 = fmaf(ov[cc].x, w, O0); O1 = fmaf(ov[cc].y, w, O1);
        }
      }
      {      // own chunk last: the fixed chunk order 0 .. C-1 of the ticket form
        const float w = __expf(ssc[r][64] - M_);
        L = fmaf(ssc[r][65], w, L); O0 = fmaf(ssc[r][2 * dp], w, O0); O1 = fmaf(ssc[r][2 * dp + 1], w, O1);
      }
      const float inv = 1.0f / L;
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      const f16x2 o2 = {(f16)(O0 * inv), (f16)(O1 * inv)};
      *reinterpret_cast<f16x2*>(out + (out_mb ? xf_index(b * R + r, h * 64 + 2 * dp, out_mb) : (size_t)(b * R + r) * d + h * 64 + 2 * dp)) = o2;
    }
    if (tid == 0) __hip_atomic_store(epoch + 1 + b * H + h, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // next launch's epoch
    stamp(pf, 6);
    if (tid == 0) tl_end(prof);
    return;
  }
  // ---- split-T: publish the partial with write-through stores; the last-arriving workgroup combines
  float* pbase = part + ((size_t)(b * H + h) * C) * R * 66;
  if (l15 < R) {
    float* pp = pbase + ((size_t)c * R + l15) * 66;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) st_sc1(pp + dh0 + r4, oacc[r4]);
    if (wave == 0 && kq == 0) { st_sc1(pp + 64, smx[l15]); st_sc1(pp + 65, ssum[l15]); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  stamp(pf, 5);
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counters + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev == (unsigned)(C - 1));
    if (last) {
      __hip_atomic_store(counters + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                          // drop this CU's stale L1 lines
    }
    s_last = last;
  }
  __syncthreads();
  stamp(pf, 6);
  if (tid == 0) tl_end(prof);
  if (!s_last) return;
  // one round: thread = (row r, dh pair); plain (pipelined) loads, made safe by the agent-scope acquire that lane 0
  // executed after winning the ticket (guide §6 G16 consumer form: relaxed ticket -> ONE acquire -> barrier -> plain loads;
  // 8-byte sc1 buffer loads were measured 4x slower than this on MI355X)
  // (R <= 16 rows x 32 dh pairs = up to 512 items over the 256 threads: the merged prefill pass of a long prompt has more than 8
  // rows per utterance - the op-level test at R = 16 found rows 8..15 uncombined when this was a single `if (tid < R * 32)`)
  for (int item = tid; item < R * 32; item += 256) {
    const int r = item >> 5, dp = item & 31;
    float2 ml[CM], ov[CM];
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) {
      if (cc < C) {
        const float* pp = pbase + ((size_t)cc * R + r) * 66;
        ml[cc] = *reinterpret_cast<const float2*>(pp + 64);
        ov[cc] = *reinterpret_cast<const float2*>(pp + 2 * dp);
      }
    }
    float M_ = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) if (cc < C) M_ = fmaxf(M_, ml[cc].x);
    float L = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) {
      if (cc < C) {
        const float w = __expf(ml[cc].x - M_);
        L = fmaf(ml[cc].y, w, L); O0 = fmaf(ov[cc].x, w, O0); O1 = fmaf(ov[cc].y, w, O1);
      }
    }
    const float inv = 1.0f / L;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 o2 = {(f16)(O0 * inv), (f16)(O1 * inv)};
    *reinterpret_cast<f16x2*>(out + (out_mb ? xf_index(b * R + r, h * 64 + 2 * dp, out_mb) : (size_t)(b * R + r) * d + h * 64 + 2 * dp)) = o2;
  }
  stamp(pf, 7);
  if (tid == 0) tl_end(prof);
}

// =======================================================================================
// Cross-attention of the batched step with ROLES (round 6; the folded query from row partials, 256-key chunks, <= 8 rows, d <= 1280).
// What the phase stamps of dec_cross_attn_kernel<4, 6, 4> showed at 8 utterances (profiles/r06_phase_cycles.md): a wave spends half of its
// life getting its 17 requests ACCEPTED (960 workgroups ask for 61 MB at once; the CU's address path takes them at the rate HBM
// answers), and scores, softmax and P.V - 6 us - start only behind that: a wave that is waiting for the memory pipeline to take its V
// requests cannot multiply the K fragments that have landed.  Here the waves of a workgroup have roles:
//   waves 0-1 (K waves)  request the query's operands, the row partials and the K fragments of THEIR 128 keys, finish the folded query,
//                        take scores, maximum, exponentials and row sums of their half chunk in registers (the MFMA output layout keeps a
//                        query row in a lane column: two cross-row shuffles, no score buffer), exchange the two row maxima through LDS flag
//                        words (K wave with K wave: no workgroup barrier) and leave P (f16) + (max, sum) in LDS behind a flag word;
//   waves 2-3 (V waves)  request V^T (32 head dimensions each) once the K waves' requests are in (one raw barrier), so K leads V through
//                        the memory pipeline, then accumulate O = V P half chunk by half chunk as the flags come up and publish the chunk
//                        partial as before (two granules per 16-byte store).
// The K waves never wait for a V wave (no workgroup barrier between the first one and the tail), so scores and softmax run while V is
// still streaming in.  Arithmetic: P and O are dec_cross_attn_kernel's bit for bit (exponentials about the chunk's maximum, P.V steps in key
// order); the row sum is added in another order (last-bit differences); partial format, combine and its order are the same.
// NT: K / V fragments by non-temporal loads - small grids only (every fragment is read once by one workgroup; measured -0.4 % of the one-utterance
// step, +1.6 % of the step at 8 utterances: session r6D)
#define WIS_CA_LD(p) (NT ? __builtin_nontemporal_load(p) : *(p))
// two granules in one 16-byte write-through store (the lane's neighbouring values): each half is a naturally aligned 8-byte unit of one 16-byte
// aligned store of one lane, which the memory pipeline carries in one piece (the reader still takes 8-byte granules); a quarter of the write
// transactions of the one-granule form
__device__ __forceinline__ void st_gran2(gran_t* p, unsigned tag, float v0, float v1) {
  const u32x4 g = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(g) : "memory");
}
// (Measured and not kept, round 6: the folded query finished ONCE per workgroup by the first V wave - which has nothing to ask for before the barrier - instead of by
// every K wave for itself, 81 instead of 96 wave requests per workgroup: 1.222 / 1.219 vs 1.204 / 1.205 ms per step at one utterance - the K waves then wait for a
// wave that started later - and 1.863 / 1.874 / 1.881 / 1.888 vs 1.880 / 1.885 / 1.883 / 1.886 at eight: nothing; sessions r6R / r6S.)
// NKW = K waves = V waves per workgroup: 2 (256 threads, four workgroups per CU: the batched step's 960 workgroups in one round) or 4 (512 threads: the
// one-utterance step's 120 workgroups have a CU each - twice the waves put the workgroup's 160 requests into the CU's address path in half the time, and
// every K wave has half the scores and exponentials to take: its path - requests, query, scores, softmax - is the critical one there)
// all-reduce over the four 16-lane rows of a lane column (lanes l, l + 16, l + 32, l + 48) with gfx950's row-swap moves: v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second (both = v: one register then holds the lower half's values twice, the other the upper half's),
// v_permlane16_swap the odd rows of the first with the even rows of the second - two VALU moves instead of two ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ float xrow_max(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xrow_sum(float v) {      // (both lanes of a pair add the same two operands in the same order: every lane of the column holds the same bits)
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <bool SPIN, bool NT, int NKW>
__global__ __launch_bounds__(128 * NKW) void dec_cross_attn_rs_kernel(const float* __restrict__ q, const f16* __restrict__ kx, const f16* __restrict__ vt,
                                                                const float* __restrict__ xres, const float* __restrict__ q2, unsigned* epoch, int RHCC, int dT,
                                                                int TpadF, f16* __restrict__ out, float* part, unsigned* counters,
                                                                unsigned long long* prof, int out_mb,
                                                                const float* __restrict__ qcs, const float* __restrict__ qb, gran_t* gran) {
  constexpr int CM = 6, SQP = 68, NPT = 10, KT = 16 / NKW, KEYS = 256 / NKW, NB = 4 / NKW, VS = 8 / NKW, NTHR = 128 * NKW;      // key tiles / keys per K wave, 16-dh blocks / 32-key steps per flag per V wave
  // (CM: <= 6 chunks; SQP: staged operand pitch; NPT: row partials per lane, 8 lanes per row, d <= 1280)
  const int R = RHCC & 0xFF, H = (RHCC >> 8) & 0x3F, C = (RHCC >> 24) & 0xFF, d = dT & 0xFFFF, T = (dT >> 16) & 0xFFFF;
  __shared__ __attribute__((aligned(16))) f16 sp16[16 * CA_PSTR];
  __shared__ __attribute__((aligned(16))) float sq[NKW][10 * SQP];      // per K wave: rows 0..7 q_raw (both halves added), 8 column sums, 9 bias
  __shared__ float srow[NKW][8][2];
  __shared__ float sml[NKW][16][2];                                    // (chunk max, sum) of the K waves' parts
  __shared__ float own[16][66];                                        // the combiner's own partial
  __shared__ float smaxw[NKW][16];                                       // the K waves' row maxima (exchanged behind flagm)
  __shared__ int flag[NKW], flagm[NKW];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int Tpad = TpadF & 0x3FFFFFFF, bkv = (TpadF >> 30) ? 0 : b;
  const int klo = c * 256, n = (klo + 256 <= T) ? 256 : T - klo;
  const bool kwave = wave < NKW; const int kw = wave & (NKW - 1);
  const int rq = l15 < R ? l15 : R - 1;
  // (tap builds: workgroup (0, 0, 0): its first K wave stamps entries 0-5 of the row, its first V wave entries 7-13)
  unsigned long long* pf = (c == 0 && h == 0 && b == 0 && lane == 0 && prof) ? (wave == 0 ? prof : (wave == NKW ? prof + 7 : nullptr)) : nullptr;
  if (tid == 0) tl_begin(prof);
  stamp(pf, 0);
  unsigned ep_now = 0;
  if (SPIN) ep_now = __hip_atomic_load(epoch + 1 + b * H + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float o_[NB][4]; float Mc = 0.f, Lc = 0.f;      // V waves: the chunk's partial (NB 16-dh blocks of lane column r = l15)
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) o_[i][r4] = 0.f;

  if (kwave) {
    // ---- requests: the query's operands (one 16-byte piece per lane and array), the row partials (lane = (row, eighth)), the K fragments
    float4 stq[2], stq2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sr0 = (lane >> 4) + 4 * u, sr = sr0 < R ? sr0 : R - 1;
      const size_t so = (size_t)(b * R + sr) * d + h * 64 + 4 * (lane & 15);
      stq[u] = *reinterpret_cast<const float4*>(q + so);
      stq2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q2) stq2[u] = *reinterpret_cast<const float4*>(q2 + so);      // (uniform)
    }
    const float4 stc = *reinterpret_cast<const float4*>(((lane & 16) ? qb : qcs) + h * 64 + 4 * (lane & 15));
    const int ntile = d >> 4, l8 = lane & 7, prow = (lane >> 3) < R ? (lane >> 3) : R - 1;
    const float2* sp = reinterpret_cast<const float2*>(xres) + (size_t)(b * R + prow) * ntile;
    float2 pt[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) { const int t = l8 + 8 * i; pt[i] = sp[t < ntile ? t : ntile - 1]; }
    const f16* kb = kx + (size_t)(bkv * H + h) * 8 * T * 8;
    u32x4 kf[KT][2];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      int key = klo + KEYS * kw + 16 * i + l15; if (key > T - 1) key = T - 1;      // clamped; masked below
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) kf[i][ks] = WIS_CA_LD(reinterpret_cast<const u32x4*>(kb + ((size_t)(kq + 4 * ks) * T + key) * 8));
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");      // the K requests are in: the V waves may ask for theirs (raw: no wait for any load)
    __builtin_amdgcn_sched_barrier(0);           // (and nothing that waits for a load moves above it: hipcc had hoisted the first statistics' wait)
    stamp(pf, 1);
    // ---- LayerNorm statistics of the R rows (Chan merge of the per-16-column (sum, M2) pairs about the first tile's mean), in every K wave
    const float c0 = __shfl(pt[0].x, lane & ~7) * 0.0625f;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const bool in = l8 + 8 * i < ntile;
      const float dm = in ? pt[i].x * 0.0625f - c0 : 0.f;
      a1 += dm; a2 += in ? pt[i].y + 16.0f * dm * dm : 0.f;
    }
    a1 += dpp_f<0xB1>(a1); a2 += dpp_f<0xB1>(a2);
    a1 += dpp_f<0x4E>(a1); a2 += dpp_f<0x4E>(a2);
    a1 += dpp_f<0x141>(a1); a2 += dpp_f<0x141>(a2);      // the 8 lanes of a row hold its sums
    float* sqw = sq[kw];
    if (l8 == 0) {
      const float invd = 1.0f / (float)d, dmu = a1 * 16.0f * invd;
      srow[kw][lane >> 3][0] = c0 + dmu; srow[kw][lane >> 3][1] = 1.0f / sqrtf(fmaxf(a2 * invd - dmu * dmu, 0.f) + 1e-5f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      *reinterpret_cast<float4*>(&sqw[((lane >> 4) + 4 * u) * SQP + 4 * (lane & 15)]) = make_float4(stq[u].x + stq2[u].x, stq[u].y + stq2[u].y, stq[u].z + stq2[u].z, stq[u].w + stq2[u].w);
    if (lane < 32) *reinterpret_cast<float4*>(&sqw[(8 + (lane >> 4)) * SQP + 4 * (lane & 15)]) = stc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private staging: the wave's own LDS writes have landed
    const float mu = srow[kw][rq][0], rs = srow[kw][rq][1];
    const float* sqr = &sqw[rq * SQP + 8 * kq]; const float* scs = &sqw[8 * SQP + 8 * kq]; const float* sbq = &sqw[9 * SQP + 8 * kq];
    f16x8 qf0, qf1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qf0[e] = (f16)(rs * (sqr[e] - mu * scs[e]) + sbq[e]);
      qf1[e] = (f16)(rs * (sqr[32 + e] - mu * scs[32 + e]) + sbq[32 + e]);
    }
    stamp(pf, 2);
    // ---- scores of this wave's KEYS keys: D[key][r], lane holds r = l15 and keys 16 i + 4 kq + reg; maximum and sums across the four lane rows
    f32x4 acc[KT];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&kf[i][0]), qf0, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&kf[i][1]), qf1, a, 0, 0, 0);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int kl = KEYS * kw + 16 * i + 4 * kq + r4;
        a[r4] = (kl < n) ? a[r4] : -INFINITY;
        lmax = fmaxf(lmax, a[r4]);
      }
      acc[i] = a;
    }
    stamp(pf, 3);
    lmax = xrow_max(lmax);
    // the chunk's maximum = the larger of the two K waves' (the exponentials are taken about the CHUNK's maximum, as dec_cross_attn_kernel takes them:
    // P and O come out bit for bit as there, only the row sum is added in another order): the one exchange between the two K waves, through LDS
    // behind a flag word each - no workgroup barrier, the V waves are not involved
    if (kq == 0) smaxw[kw][l15] = lmax;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&flagm[kw], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int w = 0; w < NKW; ++w)
      while (__hip_atomic_load(&flagm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float mref = smaxw[0][l15];                               // finite: the chunk holds a key
#pragma unroll
    for (int w = 1; w < NKW; ++w) mref = fmaxf(mref, smaxw[w][l15]);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      f16x4 ph;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) { const float e = __expf(acc[i][r4] - mref); lsum += e; ph[r4] = (f16)e; }
      *reinterpret_cast<f16x4*>(&sp16[l15 * CA_PSTR + KEYS * kw + 16 * i + 4 * kq]) = ph;
    }
    lsum = xrow_sum(lsum);
    if (kq == 0) { sml[kw][l15][0] = mref; sml[kw][l15][1] = lsum; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&flag[kw], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    stamp(pf, 4);
  } else {
    if (lane < 2 * NKW && wave == NKW) __hip_atomic_store(lane < NKW ? &flag[lane] : &flagm[lane - NKW], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    stamp(pf, 0);
    // ---- V^T fragments of this wave's 16 NB head dimensions, in key order (the first K wave's keys first)
    const f16* vb = vt + ((size_t)(bkv * H + h) * 64 + 16 * NB * kw + l15) * Tpad + klo + 8 * kq;
    u32x4 vf[NB][8];
#pragma unroll
    for (int j = 0; j < NKW; ++j)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int s4 = 0; s4 < VS; ++s4) vf[blk][VS * j + s4] = WIS_CA_LD(reinterpret_cast<const u32x4*>(vb + (size_t)blk * 16 * Tpad + 32 * (VS * j + s4)));
    __builtin_amdgcn_sched_barrier(0);
    stamp(pf, 1);
    f32x4 oacc[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) oacc[blk] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKW; ++j) {      // the 32-key steps in key order, as dec_cross_attn_kernel accumulates them; K wave j's keys as soon as its P is there
      while (__hip_atomic_load(&flag[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s4 = 0; s4 < VS; ++s4) {
        const f16x8 pb = *reinterpret_cast<const f16x8*>(&sp16[rq * CA_PSTR + 32 * (VS * j + s4) + 8 * kq]);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) oacc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&vf[blk][VS * j + s4]), pb, oacc[blk], 0, 0, 0);
      }
      if (j == 0 || j == NKW - 1) stamp(pf, j == 0 ? 2 : 3);
    }
    Mc = sml[0][rq][0];
    if (NKW == 2) Lc = sml[0][rq][1] + sml[1][rq][1];
    else Lc = (sml[0][rq][1] + sml[1][rq][1]) + (sml[2 % NKW][rq][1] + sml[3 % NKW][rq][1]);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) o_[blk][r4] = oacc[blk][r4];
    stamp(pf, 4);
  }

  // ---- the chunk's partial: (O[64], M, L) per row, published / combined as in dec_cross_attn_kernel (V wave kw owns dh 16 NB kw .. 16 NB (kw + 1) - 1)
  const int dhb = 16 * NB * kw + 4 * kq;      // + 16 blk
  const bool pub = !kwave && l15 < R;
  if (C == 1) {
    if (pub) {
      const float inv = 1.0f / Lc;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const f16x4 o = {(f16)(o_[blk][0] * inv), (f16)(o_[blk][1] * inv), (f16)(o_[blk][2] * inv), (f16)(o_[blk][3] * inv)};
        *reinterpret_cast<f16x4*>(out + (out_mb ? xf_index(b * R + l15, h * 64 + dhb + 16 * blk, out_mb) : (size_t)(b * R + l15) * d + h * 64 + dhb + 16 * blk)) = o;
      }
    }
    return;
  }
  if (SPIN) {
    const unsigned tag = ep_now + 1u;
    gran_t* gbase = gran + (size_t)(b * H + h) * 6 * 8 * 66;
    if (c != C - 1) {
      if (pub) {
        gran_t* gp = gbase + ((size_t)c * 8 + l15) * 66;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) { st_gran2(gp + dhb + 16 * blk, tag, o_[blk][0], o_[blk][1]); st_gran2(gp + dhb + 16 * blk + 2, tag, o_[blk][2], o_[blk][3]); }
        if (wave == NKW && kq == 0) st_gran2(gp + 64, tag, Mc, Lc);
      }
      stamp(pf, 5);
      if (tid == 0) tl_end(prof);
      return;
    }
    if (pub) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) own[l15][dhb + 16 * blk + r4] = o_[blk][r4];
      if (wave == NKW && kq == 0) { own[l15][64] = Mc; own[l15][65] = Lc; }
    }
    __syncthreads();
    stamp(pf, 5);
    for (int item = tid; item < R * 32; item += NTHR) {
      const int r = item >> 5, dp = item & 31;
      float2 ml[CM], ov[CM];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int cc = 0; cc < CM - 1; ++cc) {
          if (cc < C - 1) {
            const gran_t* gp = gbase + ((size_t)cc * 8 + r) * 66;
            const gran_t g0 = ld_gran(gp + 64), g1 = ld_gran(gp + 65), g2 = ld_gran(gp + 2 * dp), g3 = ld_gran(gp + 2 * dp + 1);
            ok = ok & ((unsigned)(g0 >> 32) == tag) & ((unsigned)(g1 >> 32) == tag) & ((unsigned)(g2 >> 32) == tag) & ((unsigned)(g3 >> 32) == tag);
            ml[cc] = make_float2(__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1));
            ov[cc] = make_float2(__uint_as_float((unsigned)g2), __uint_as_float((unsigned)g3));
          }
        }
        if (ok) break;
        if (++spins > CA_SPIN_LIMIT) { atomicOr(epoch, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      float M_ = own[r][64];
#pragma unroll
      for (int cc = 0; cc < CM - 1; ++cc) if (cc < C - 1) M_ = fmaxf(M_, ml[cc].x);
      float L = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
      for (int cc = 0; cc < CM - 1; ++cc) {
        if (cc < C - 1) {
          const float w = __expf(ml[cc].x - M_);
          L = fmaf(ml[cc].y, w, L); O0 = fmaf(ov[cc].x, w, O0); O1 = fmaf(ov[cc].y, w, O1);
        }
      }
      {
        const float w = __expf(own[r][64] - M_);
        L = fmaf(own[r][65], w, L); O0 = fmaf(own[r][2 * dp], w, O0); O1 = fmaf(own[r][2 * dp + 1], w, O1);
      }
      const float inv = 1.0f / L;
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      const f16x2 o2 = {(f16)(O0 * inv), (f16)(O1 * inv)};
      *reinterpret_cast<f16x2*>(out + (out_mb ? xf_index(b * R + r, h * 64 + 2 * dp, out_mb) : (size_t)(b * R + r) * d + h * 64 + 2 * dp)) = o2;
    }
    if (tid == 0) __hip_atomic_store(epoch + 1 + b * H + h, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamp(pf, 6);
    if (tid == 0) tl_end(prof);
    return;
  }
  // ---- ticket form (large grids)
  float* pbase = part + ((size_t)(b * H + h) * C) * R * 66;
  if (pub) {
    float* pp = pbase + ((size_t)c * R + l15) * 66;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) st_sc1(pp + dhb + 16 * blk + r4, o_[blk][r4]);
    if (wave == NKW && kq == 0) { st_sc1(pp + 64, Mc); st_sc1(pp + 65, Lc); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stamp(pf, 5);
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counters + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev == (unsigned)(C - 1));
    if (last) {
      __hip_atomic_store(counters + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  if (tid == 0) tl_end(prof);
  if (!s_last) return;
  for (int item = tid; item < R * 32; item += NTHR) {
    const int r = item >> 5, dp = item & 31;
    float2 ml[CM], ov[CM];
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) {
      if (cc < C) {
        const float* pp = pbase + ((size_t)cc * R + r) * 66;
        ml[cc] = *reinterpret_cast<const float2*>(pp + 64);
        ov[cc] = *reinterpret_cast<const float2*>(pp + 2 * dp);
      }
    }
    float M_ = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) if (cc < C) M_ = fmaxf(M_, ml[cc].x);
    float L = 0.f, O0 = 0.f, O1 = 0.f;
#pragma unroll
    for (int cc = 0; cc < CM; ++cc) {
      if (cc < C) {
        const float w = __expf(ml[cc].x - M_);
        L = fmaf(ml[cc].y, w, L); O0 = fmaf(ov[cc].x, w, O0); O1 = fmaf(ov[cc].y, w, O1);
      }
    }
    const float inv = 1.0f / L;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 o2 = {(f16)(O0 * inv), (f16)(O1 * inv)};
    *reinterpret_cast<f16x2*>(out + (out_mb ? xf_index(b * R + r, h * 64 + 2 * dp, out_mb) : (size_t)(b * R + r) * d + h * 64 + 2 * dp)) = o2;
  }
  stamp(pf, 6);
}

int launch_dec_cross_attn(hipStream_t st, const float* q, const f16* kx, const f16* vt, f16* out, float* part, unsigned* counters,
                          int B, int R, int H, int d, int T, int Tpad, int chunks, unsigned long long* prof, int out_mb,
                          const float* xres, const float* qcs, const float* qb, unsigned long long* gran, unsigned* epoch, const float* q2, int xres_is_stat, int kv_shared) {
  if (xres && (!qcs || !qb || R > 8 || d > (xres_is_stat ? 2048 : 1280))) { set_error("dec_cross_attn: folded query needs column sums, bias, R <= 8 and d <= 1280 (2048 from partials)"); return WIS_E_ARG; }
  if (xres_is_stat && !xres) { set_error("dec_cross_attn: the fold from partials needs the row partials"); return WIS_E_ARG; }
  if (R < 1 || R > 16 || chunks < 1 || chunks > 16) { set_error("dec_cross_attn: R=%d chunks=%d unsupported", R, chunks); return WIS_E_UNSUPPORTED; }
  if (H > 63 || d > 65535 || T > 65535) { set_error("dec_cross_attn: H=%d d=%d T=%d beyond the packed shape arguments", H, d, T); return WIS_E_UNSUPPORTED; }
  const int CL = cdiv(cdiv(T, chunks), 32) * 32;     // chunk length: multiple of 32 keys (16-byte aligned V^T fragments)
  if ((CL != 128 && CL != 256) || CL * chunks > Tpad) { set_error("dec_cross_attn: %d chunks of %d keys unsupported (Tpad %d)", chunks, CL, Tpad); return WIS_E_UNSUPPORTED; }
  const int used = cdiv(T, CL);                      // chunks that actually hold keys
  static const int env_spin = getenv("WIS_CA_SPIN") ? atoi(getenv("WIS_CA_SPIN")) : 1;      // 0: always the ticket form (A/B switch)
  // granule hand-off: small grids only (fewer spinning combiners than CUs), the default 256-key chunking, <= 8 rows per utterance
  const bool spin = env_spin && gran && epoch && B * H <= CA_SPIN_MAX_BH && CL == 256 && used >= 2 && used <= 6 && R <= 8;
  static const int lds_pad = getenv("WIS_CA_LDS_PAD") ? atoi(getenv("WIS_CA_LDS_PAD")) : 0;      // experiment: dynamic LDS nobody uses = fewer workgroups per CU
  const int shm = (long)B * H * used > 512 ? lds_pad : 0;
#define WIS_CA(TPWv, CMv, FOLDv, SPINv) hipLaunchKernelGGL((dec_cross_attn_kernel<TPWv, CMv, FOLDv, SPINv>), dim3(used, H, B), dim3(256), shm, st, q, kx, vt, xres, q2, epoch, \
                                                  (R | (H << 8) | (CL << 14) | (used << 24)), (d | (T << 16)), (Tpad | (kv_shared ? (1 << 30) : 0)), out, part, counters, prof, out_mb, qcs, qb, gran)
  if (xres_is_stat) {
    const bool small = (long)B * H * used <= 256;      // at most one workgroup per CU: V is requested up front (FOLD 3)
    // the role-split kernel (K waves / V waves): WIS_CA_RS=0 off (A/B switch), 1 the batched step's grids only, 2 (default) the one-utterance step's too
    static const int ca_rs = getenv("WIS_CA_RS") ? atoi(getenv("WIS_CA_RS")) : 2;
    if (ca_rs && (!small || ca_rs >= 2) && CL == 256 && used <= 6 && R <= 8 && d <= 1280) {
#define WIS_CA_RS(SPINv, NTv, NKWv) hipLaunchKernelGGL((dec_cross_attn_rs_kernel<SPINv, NTv, NKWv>), dim3(used, H, B), dim3(128 * NKWv), 0, st, q, kx, vt, xres, q2, epoch, \
                                                        (R | (H << 8) | (CL << 14) | (used << 24)), (d | (T << 16)), (Tpad | (kv_shared ? (1 << 30) : 0)), out, part, counters, prof, out_mb, qcs, qb, gran)
      static const bool rs8 = getenv("WIS_CA_RS8") && atoi(getenv("WIS_CA_RS8")) != 0;      // small grids with 4 + 4 waves per workgroup: measured 0.8 % SLOWER per step than 2 + 2 (session r6G); A/B switch, off
      if (spin) { if (small && rs8) WIS_CA_RS(true, true, 4); else if (small) WIS_CA_RS(true, true, 2); else WIS_CA_RS(true, false, 2); }
      else { if (small && rs8) WIS_CA_RS(false, true, 4); else if (small) WIS_CA_RS(false, true, 2); else WIS_CA_RS(false, false, 2); }
#undef WIS_CA_RS
      return WIS_OK;
    }
    // FOLD 4 (large grids, <= 8 rows): the folded query's operands through LDS, V requested up front (WIS_CA_FOLD4=0: FOLD 2, V behind the prologue)
    static const bool fold4 = !(getenv("WIS_CA_FOLD4") && atoi(getenv("WIS_CA_FOLD4")) == 0);
    const bool f4 = fold4 && !small && R <= 8;
    if (spin) { if (small) WIS_CA(4, 6, 3, true); else if (f4) WIS_CA(4, 6, 4, true); else WIS_CA(4, 6, 2, true); }
    else if (CL <= 128) WIS_CA(2, 16, 2, false);
    else if (used <= 6) { if (small) WIS_CA(4, 6, 3, false); else if (f4) WIS_CA(4, 6, 4, false); else WIS_CA(4, 6, 2, false); } else WIS_CA(4, 16, 2, false);
  }
  else if (spin) { if (xres) WIS_CA(4, 6, 1, true); else WIS_CA(4, 6, 0, true); }
  else if (xres) { if (CL <= 128) WIS_CA(2, 16, 1, false); else if (used <= 6) WIS_CA(4, 6, 1, false); else WIS_CA(4, 16, 1, false); }
  else { if (CL <= 128) WIS_CA(2, 16, 0, false); else if (used <= 6) WIS_CA(4, 6, 0, false); else WIS_CA(4, 16, 0, false); }
#undef WIS_CA
  return WIS_OK;
}

// =======================================================================================
// logits processors + log-softmax statistics + per-sub-chunk top-n_cand.  grid (STAT_SUB / 4, M), block 256: every WAVE owns one
// of the STAT_SUB = 64 sub-chunks of a logits row (811 values for the multilingual vocabulary, 13 per lane) and works alone - the
// maximum, the sum of exponentials and the n_cand selection rounds are DPP wave reductions, no workgroup barrier anywhere
// (round 1 ran ten block-wide arg-best rounds with a barrier each: 21 us per step).
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }
// the same order without short-circuit control flow (hipcc turns `||` / `&&` on lane-varying operands into exec-mask branches;
// inside unrolled selection loops that is one branch per element)
__device__ __forceinline__ bool better_b(float av, int ai, float bv, int bi) { return (av > bv) | ((av == bv) & (ai < bi)); }
__device__ __forceinline__ void take_better(float& v, int& i, float ov, int oi) { const bool t = better_b(ov, oi, v, i); v = t ? ov : v; i = t ? oi : i; }

__global__ __launch_bounds__(256) void logit_stats_kernel(const float* __restrict__ logits, const float* __restrict__ bias_all,
                                                          const float* __restrict__ bias_begin, const int* __restrict__ step_u,
                                                          float* __restrict__ st_max, float* __restrict__ st_sum,
                                                          float* __restrict__ st_val, int* __restrict__ st_idx, SampleCfg cfg,
                                                          int lr_b, int lr_j, int lr_off, unsigned long long* prof, const int* __restrict__ rowmap) {
  constexpr int PT = 16;   // values per lane: supports n_vocab <= 64 * 64 * 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sc = blockIdx.x * 4 + wave, m = blockIdx.y, b = m / cfg.beam;
  unsigned long long* pf = (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) ? prof : nullptr;
  stamp(pf, 0);
  const int step = step_u[b];
  const int SL = cdiv(cfg.n_vocab, STAT_SUB);
  const int lo = sc * SL, hi = (lo + SL < cfg.n_vocab) ? lo + SL : cfg.n_vocab;
  // logits row of (utterance b, beam j): decode steps b*beam + j; the merged prefill+first step samples every beam from the
  // utterance's last prompt row (beams > 0 carry cum = -inf there)
  // (rowmap: draft verification of a beam search - live beam j's logits sit in the row of the draft node it was matched to)
  const int jb = m - b * cfg.beam;
  const float* row = logits + (size_t)(b * lr_b + (rowmap ? rowmap[jb] : jb) * lr_j + lr_off) * cfg.n_vocab_pad;
  const bool first = (step == 0) && cfg.suppress_blank;
  const bool mask_eot = cfg.fixed_new > 0 && step < cfg.fixed_new;
  const bool force_eot = cfg.fixed_new > 0 && step >= cfg.fixed_new;

  // straight-line loads: every stream of every value is requested before anything is waited for (clamped addresses, masks applied
  // by select - a guarded `if (nidx < hi) { load }` per value compiled into sixteen branches with a wait each: 15k cycles)
  float vals[PT]; u64 keys[PT];
  float lv[PT], la[PT], lb[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nidx = lo + lane + 64 * i;
    const int cidx = nidx < hi ? nidx : hi - 1;
    lv[i] = row[cidx]; lb[i] = bias_begin[cidx]; la[i] = bias_all ? bias_all[cidx] : 0.f;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nidx = lo + lane + 64 * i;
    float v = lv[i] + la[i];
    v = first ? v + lb[i] : v;
    const bool dead = (nidx >= hi) | (mask_eot & (nidx == cfg.eot)) | (force_eot & (nidx != cfg.eot));
    v = dead ? -INFINITY : v;
    vals[i] = v; keys[i] = nidx < hi ? sel_key(v, nidx) : 0ull; mx = fmaxf(mx, v);
  }
  stamp(pf, 1);
  mx = wave_max(mx);
  float sum = 0.f;
  if (mx > -INFINITY) {
#pragma unroll
    for (int i = 0; i < PT; ++i) sum += __expf(vals[i] - mx);   // exp(-inf) = 0 for masked / out of range
  }
  sum = wave_sum(sum);
  if (lane == 0) { st_max[m * STAT_SUB + sc] = mx; st_sum[m * STAT_SUB + sc] = sum; }
  stamp(pf, 2);
  // top-n_cand of the sub-chunk in (value desc, index asc) order: n_cand rounds of a wave arg-best over the lanes' best
  // not-yet-picked value
  // n_cand rounds: lane-local best key strictly below the previous pick (independent tests, a 4-level max tree - a scan that
  // threads one running best through all 16 values is a 16-deep dependent chain, and a lone wave per SIMD has nothing to hide a
  // dependent VALU latency behind), then a wave maximum of the keys
  u64 prev = ~0ull;
  for (int rnd = 0; rnd < cfg.n_cand; ++rnd) {
    u64 t[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) t[i] = keys[i] < prev ? keys[i] : 0ull;
#pragma unroll
    for (int w = PT / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int i = 0; i < w; ++i) t[i] = key_max(t[i], t[i + w]);
    const u64 best = wave_max_key(t[0]);
    if (lane == 0) {     // an exhausted sub-chunk (every value picked or out of range) reports (-inf, INT_MAX)
      st_val[((size_t)m * STAT_SUB + sc) * cfg.n_cand + rnd] = best ? key_value(best) : -INFINITY;
      st_idx[((size_t)m * STAT_SUB + sc) * cfg.n_cand + rnd] = best ? key_index(best) : 0x7fffffff;
    }
    prev = best;
  }
  stamp(pf, 3);
}
int launch_logit_stats(hipStream_t st, const float* logits, const float* bias_all, const float* bias_begin, const int* step_u,
                       float* st_max, float* st_sum, float* st_val, int* st_idx, int B, const SampleCfg& cfg, int lr_b, int lr_j, int lr_off,
                       unsigned long long* prof, const int* rowmap) {
  if (cdiv(cfg.n_vocab, STAT_SUB) > 16 * 64) { set_error("logit_stats: vocab too large"); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(logit_stats_kernel, dim3(STAT_SUB / 4, B * cfg.beam), dim3(256), 0, st, logits, bias_all, bias_begin, step_u,
                     st_max, st_sum, st_val, st_idx, cfg, lr_b, lr_j, lr_off, prof, rowmap);
  return WIS_OK;
}

// Greedy pick of teacher-forced rows (wis_generate_draft: the final decode of a streamed recording verifies the last interim hypothesis
// in a few multi-row passes instead of one pass per token).  One wave per logits row over what logit_stats_kernel left for it: the
// log-softmax normaliser from the 64 sub-chunk (max, sum-exp) pairs and the best candidate of the 64 sub-chunk winners, in
// beam_step_kernel's arithmetic and order (score = logit - lse, ties to the lower token id) - the token beam_step would take at k = 1.
__global__ __launch_bounds__(64) void greedy_pick_kernel(const float* __restrict__ st_max, const float* __restrict__ st_sum, const float* __restrict__ st_val,
                                                         const int* __restrict__ st_idx, int n_cand, int V, int* __restrict__ tok_out, float* __restrict__ lp_out) {
  const int m = blockIdx.x, lane = threadIdx.x;
  const float smx = st_max[m * STAT_SUB + lane], ssm = st_sum[m * STAT_SUB + lane];
  const float v = st_val[((size_t)m * STAT_SUB + lane) * n_cand];
  const int ix = st_idx[((size_t)m * STAT_SUB + lane) * n_cand];
  const float M_ = wave_max(smx);
  const float S = wave_sum(smx > -INFINITY ? ssm * __expf(smx - M_) : 0.f);
  const float lse = M_ + logf(S);
  const int tk = ix > V - 1 ? V - 1 : ix;
  const u64 best = wave_max_key(sel_key(v > -INFINITY ? v - lse : -INFINITY, tk));
  if (lane == 0) { tok_out[m] = key_index(best); lp_out[m] = key_value(best); }
}
int launch_greedy_pick(hipStream_t st, const float* st_max, const float* st_sum, const float* st_val, const int* st_idx, int rows, const SampleCfg& cfg, int* tok_out, float* lp_out) {
  hipLaunchKernelGGL(greedy_pick_kernel, dim3(rows), dim3(64), 0, st, st_max, st_sum, st_val, st_idx, cfg.n_cand, cfg.n_vocab, tok_out, lp_out);
  return WIS_OK;
}

// =======================================================================================
// beam search bookkeeping (CTranslate2 4.1.0 BeamSearch::search semantics).  grid B, block 256 (four waves per
// utterance).  Everything the step needs from memory - the candidate pool (k rows x 64 sub-chunks x n_cand), the row statistics,
// the cumulative scores, the step / done / hypothesis counters and the token histories - is requested up front in one round
// trip; the pool lives in registers (<= 32 entries per thread), the n_cand selection rounds are wave arg-bests joined through a
// double-buffered 4-entry LDS exchange (one barrier per round).
// PSL = pool entries per thread: 16 covers beam <= 5 (5 x 64 x 10 = 3200 entries), 32 the largest beam (8 x 64 x 16)
// End of a beam step for one workgroup (thread 0, after the workgroup's global writes are fenced): count it; the workgroup that
// ends the step as the LAST of the grid publishes the step to the host-mapped progress block - one 8-byte system-scope release
// store that the host polls between graph launches (model.hip generate_impl), so a search that ends on EOT is noticed without a
// stream round trip.  Results of finished utterances were written to the same block (and fenced at system scope) before their
// workgroup's count, so whoever reads "done = B" in the record finds them there.
__device__ __forceinline__ void publish_step(const BeamState& bs) {
  __threadfence();
  const unsigned B = gridDim.x;
  const unsigned n = atomicAdd(bs.tick, 1u) + 1u;
  if (n % B != 0u) return;
  __threadfence();
  const unsigned steps = n / B;
  const unsigned done = (unsigned)__hip_atomic_load(bs.all_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned gen = __hip_atomic_load(bs.tick + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned gu = bs.giveup ? (__hip_atomic_load(bs.giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1u : 0u) : 0u;
  const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  if (steps == 1u) __hip_atomic_store(bs.host + HP_STAMP0, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(bs.host + HP_STAMP, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (done >= B && atomicCAS(bs.tick + 2, 0u, steps) == 0u) {
    __hip_atomic_store(bs.host + HP_DONE_STEP, (unsigned long long)steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(bs.host + HP_DONE_STAMP, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long rec = ((unsigned long long)(gen & 0xFFFFu) << 48) | ((unsigned long long)(steps & 0xFFFFu) << 32) | ((unsigned long long)gu << 16) | (done & 0xFFFFu);
  __hip_atomic_store(bs.host + HP_REC, rec, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int PSL>
__global__ __launch_bounds__(256) void beam_step_kernel(const float* __restrict__ st_max, const float* __restrict__ st_sum,
                                                        const float* __restrict__ st_val, const int* __restrict__ st_idx,
                                                        BeamState bs, RowMeta rm, int P, int ctx, SampleCfg cfg, unsigned long long* prof) {
  constexpr int HPT = (MAX_R * 256 + 255) / 256;                    // history tokens per thread (8)
  __shared__ float lse[MAX_R], s_cum[MAX_R];
  __shared__ float cand_v[MAX_CAND]; __shared__ int cand_word[MAX_CAND]; __shared__ int cand_org[MAX_CAND];
  __shared__ int nb_src[MAX_R]; __shared__ int nb_tok[MAX_R]; __shared__ float nb_cum[MAX_R];
  __shared__ int hyp_src[MAX_R]; __shared__ int hyp_slot[MAX_R]; __shared__ int hyp_n[MAX_R]; __shared__ int n_newhyp;
  __shared__ int s_finished;
  __shared__ int sh_alive[MAX_R * 256];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* pf = (b == 0 && tid == 0) ? prof : nullptr;
  stamp(pf, 0);
  const int k = cfg.beam, NC = cfg.n_cand, V = cfg.n_vocab;
  const int r0 = b * k, per_row = STAT_SUB * NC, total = k * per_row;
  // ---- one round trip: pool, statistics, scores, counters, histories
  float pv_[PSL]; int pi_[PSL];
#pragma unroll
  for (int u = 0; u < PSL; ++u) {
    const int e = tid + 256 * u, ce = e < total ? e : total - 1;       // clamped address, masked by select: straight-line loads
    const float v = st_val[(size_t)r0 * per_row + ce]; const int ix = st_idx[(size_t)r0 * per_row + ce];     // rows r0 .. r0+k-1 are contiguous
    pv_[u] = e < total ? v : -INFINITY; pi_[u] = e < total ? ix : 0x7fffffff;
  }
  float smx[2], ssm[2];                      // wave w reduces rows w and w + 4; lane = sub-chunk
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2) {
    const int j = wave + 4 * j2, mrow = r0 + (j < k ? j : 0);
    smx[j2] = st_max[mrow * STAT_SUB + lane]; ssm[j2] = st_sum[mrow * STAT_SUB + lane];
  }
  const float my_cum = tid < k ? bs.cum[r0 + tid] : 0.f;
  int hist_tok[HPT];
#pragma unroll
  for (int u = 0; u < HPT; ++u) {            // token histories of the k beams, whole rows (nothing here waits for the step counter)
    const int i = tid + 256 * u, j = i >> 8, t = i & 255;
    const int cj = j < k ? j : k - 1, ct = t < cfg.max_new ? t : cfg.max_new - 1;
    hist_tok[u] = bs.alive[(size_t)(r0 + cj) * cfg.max_new + ct];
  }
  const int done_b = bs.done[b];
  const int step = bs.step_u[b];
  const int nh0 = bs.n_hyp[b];
  if (done_b) { if (tid == 0) publish_step(bs); return; }

  stamp(pf, 1);
#pragma unroll
  for (int u = 0; u < HPT; ++u) { const int i = tid + 256 * u; if (i < MAX_R * 256) sh_alive[i] = hist_tok[u]; }
  if (tid < k) s_cum[tid] = my_cum;
  // log-softmax normaliser per live row
#pragma unroll
  for (int j2 = 0; j2 < 2; ++j2) {
    const int j = wave + 4 * j2;
    const float M_ = wave_max(smx[j2]);
    const float S = wave_sum(smx[j2] > -INFINITY ? ssm[j2] * __expf(smx[j2] - M_) : 0.f);
    if (lane == 0 && j < k) lse[j] = M_ + logf(S);
  }
  __syncthreads();
  stamp(pf, 2);
  // score = logit - lse + cum ; id = (beam << 20) | token (orders like the flat id beam * V + token, V < 2^20); every pool entry
  // becomes one selection key (common.hpp sel_key: score descending, then id ascending)
  u64 pk[PSL];
  u64 lbest = 0ull;
  const float inv_per_row = 1.0f / (float)per_row;
#pragma unroll
  for (int u = 0; u < PSL; ++u) {
    const int e = tid + 256 * u;
    int j = (int)(((float)e + 0.5f) * inv_per_row);          // e / per_row (e < 8192: exact in fp32), no integer division
    j = j < k ? j : k - 1;                                   // entries beyond the pool: clamped row, masked below
    const float v = pv_[u];
    const float sc_ = (v - lse[j]) + s_cum[j];               // (-inf stays -inf: lse and cum of a live row are finite or -inf, never +inf)
    int tk = pi_[u]; tk = tk > V - 1 ? V - 1 : tk;           // exhausted sub-chunks report INT_MAX with -inf
    pk[u] = e < total ? sel_key(v > -INFINITY ? sc_ : -INFINITY, (j << 20) | tk) : 0ull;
    lbest = key_max(lbest, pk[u]);
  }
  stamp(pf, 3);
  // top-NC of the pool, hierarchically (a block-wide selection round costs a wave reduction, an LDS exchange and a barrier - 2400
  // cycles measured; ten of them were half the kernel): every wave selects the top NC of ITS quarter of the pool on its own (wave
  // maximum of the lane-local bests, retire, 4-level max tree; no barrier), then the 4 NC wave candidates are ranked in one pass -
  // a candidate's rank is the number of candidates with a larger key (keys are distinct: the id is part of the key).
  __shared__ u64 wcand[4 * MAX_CAND];
  for (int rnd = 0; rnd < NC; ++rnd) {
    const u64 wb = wave_max_key(lbest);
    if (lane == 0) wcand[wave * NC + rnd] = wb;
#pragma unroll
    for (int u = 0; u < PSL; ++u) pk[u] = pk[u] == wb ? 0ull : pk[u];
    u64 t[PSL];
#pragma unroll
    for (int u = 0; u < PSL; ++u) t[u] = pk[u];
#pragma unroll
    for (int w = PSL / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int u = 0; u < w; ++u) t[u] = key_max(t[u], t[u + w]);
    lbest = t[0];
  }
  if (tid < MAX_CAND) { cand_v[tid] = -INFINITY; cand_word[tid] = V - 1; cand_org[tid] = 0; }     // fewer real entries than candidates: -inf on a valid row / token
  __syncthreads();
  if (tid < 4 * NC) {
    const u64 mine = wcand[tid];
    int rank = 0;
#pragma unroll
    for (int m2 = 0; m2 < 4 * MAX_CAND; ++m2) rank += (m2 < 4 * NC && wcand[m2] > mine) ? 1 : 0;      // broadcast LDS reads, independent compares
    if (mine != 0ull && rank < NC) {
      const int bi = key_index(mine);
      cand_v[rank] = key_value(mine); cand_word[rank] = bi & 0xFFFFF; cand_org[rank] = (bi >> 20) & 0x7FF;
    }
  }
  __syncthreads();

  stamp(pf, 4);
  // serial bookkeeping
  if (tid == 0) {
    const bool is_last = (step + 1 >= cfg.max_new);
    int nh = nh0, second = k, newh = 0; bool top_finished = false;
    for (int kk = 0; kk < k; ++kk) {
      int next = kk;
      const bool eos = cand_word[kk] == cfg.eot;
      if (eos || is_last) {
        if (kk == 0) top_finished = true;
        if (nh < cfg.max_hyp) {
          hyp_src[newh] = kk; hyp_slot[newh] = nh; hyp_n[newh] = eos ? step : step + 1;   // EOS is not part of the hypothesis
          bs.hyp_score[b * cfg.max_hyp + nh] = cand_v[kk];
          bs.hyp_len[b * cfg.max_hyp + nh] = hyp_n[newh];
          ++newh; ++nh;
        }
        for (int j = second; j < NC; ++j)
          if (cand_word[j] != cfg.eot) { next = j; second = j + 1; break; }
      }
      nb_src[kk] = next; nb_tok[kk] = cand_word[next]; nb_cum[kk] = cand_v[next];
    }
    n_newhyp = newh;
    bs.n_hyp[b] = nh;
    bool fin = is_last;
    if (!fin) fin = cfg.allow_early_exit ? (top_finished && nh >= cfg.early_exit_hyps) : (nh >= cfg.max_candidates);
    s_finished = fin ? 1 : 0;
  }
  __syncthreads();

  stamp(pf, 5);
  const int hist = step;                 // tokens already in alive[] (sh_alive holds the whole rows)
  const int npos = P - 1 + step + 1;     // cache positions valid after this step
  // finished hypotheses of this step
  for (int hh = 0; hh < n_newhyp; ++hh) {
    const int kk = hyp_src[hh], org = cand_org[kk], n = hyp_n[hh];
    int* dst = bs.hyp_tok + ((size_t)b * cfg.max_hyp + hyp_slot[hh]) * cfg.max_new;
    for (int t = tid; t < n; t += 256) dst[t] = (t < hist) ? sh_alive[org * 256 + t] : cand_word[kk];
  }
  if (s_finished) {
    __syncthreads();
    __threadfence_block();
    if (tid == 0) {
      // finalize_result: score / len^length_penalty, best first
      const int nh = bs.n_hyp[b];
      int best = 0; float bsc = -INFINITY;
      for (int i = 0; i < nh; ++i) {
        float s = bs.hyp_score[b * cfg.max_hyp + i];
        const int n = bs.hyp_len[b * cfg.max_hyp + i];
        if (cfg.length_penalty != 0.f) s /= powf((float)n, cfg.length_penalty);
        if (s > bsc) { bsc = s; best = i; }
      }
      cand_org[0] = best;
      bs.out_len[b] = bs.hyp_len[b * cfg.max_hyp + best];
      bs.out_score[b] = bsc;
      bs.done[b] = 1;
      hp_out_len(bs.host)[b] = bs.hyp_len[b * cfg.max_hyp + best];      // the host's copy of the result (fenced below, before the step is published)
      hp_out_score(bs.host)[b] = bsc;
    }
    __syncthreads();
    const int best = cand_org[0], n = bs.hyp_len[b * cfg.max_hyp + best];
    const int* src = bs.hyp_tok + ((size_t)b * cfg.max_hyp + best) * cfg.max_new;
    int* hids = hp_out_ids(bs.host) + (size_t)b * cfg.max_new;
    for (int t = tid; t < n; t += 256) { const int v = src[t]; bs.out_ids[(size_t)b * cfg.max_new + t] = v; hids[t] = v; }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) { atomicAdd(bs.all_done, 1); publish_step(bs); }
    return;
  }
  stamp(pf, 7);
  // next live beams
  for (int j = 0; j < k; ++j) {
    const int org = cand_org[nb_src[j]];
    int* al = bs.alive + (size_t)(r0 + j) * cfg.max_new;
    for (int t = tid; t < hist; t += 256) al[t] = sh_alive[org * 256 + t];
    if (tid == 0) {
      al[hist] = nb_tok[j];
      // KV slot this beam continues from: the merged prefill + first step left the prompt's K/V in the utterance's first slot
      bs.parent[r0 + j] = (step == 0) ? r0 : r0 + org;
      { int* tr = bs.traj + ((size_t)(b * 256 + step) * MAX_R + j) * 2; tr[0] = nb_tok[j]; tr[1] = org; }      // the search's trajectory (a later call's draft)
      bs.cum[r0 + j] = nb_cum[j];
      rm.tok[r0 + j] = nb_tok[j];
      rm.pos[r0 + j] = npos;
      rm.slot[r0 + j] = r0 + j;                    // decode rows own their KV slot (the merged first step ran on prompt rows)
    }
  }
  if (tid == 0) bs.step_u[b] = step + 1;
  __syncthreads();
  if (tid == 0) publish_step(bs);
  stamp(pf, 8);
}
int launch_beam_step(hipStream_t st, const float* st_max, const float* st_sum, const float* st_val, const int* st_idx,
                     const BeamState& bs, const RowMeta& rm, int B, int P, int ctx, const SampleCfg& cfg, unsigned long long* prof) {
  if (cfg.beam > MAX_R || cfg.n_cand > MAX_CAND || cfg.max_new > 256 || ctx > 512 || cfg.n_vocab > (1 << 20)) { set_error("beam_step: config out of range"); return WIS_E_UNSUPPORTED; }
  if (cfg.beam * STAT_SUB * cfg.n_cand <= 16 * 256) hipLaunchKernelGGL(beam_step_kernel<16>, dim3(B), dim3(256), 0, st, st_max, st_sum, st_val, st_idx, bs, rm, P, ctx, cfg, prof);
  else hipLaunchKernelGGL(beam_step_kernel<32>, dim3(B), dim3(256), 0, st, st_max, st_sum, st_val, st_idx, bs, rm, P, ctx, cfg, prof);
  return WIS_OK;
}

// =======================================================================================
// Draft verification at beam > 1 (model.hip generate_impl; one utterance): a window of replayed beam steps is queued WITHOUT host round trips; this
// one-workgroup kernel behind every replayed step matches the live set the step produced (tokens + the beams they continued from, as beam_step_kernel
// recorded them in bs.traj) with the draft's entry for that step - as SETS: two candidates whose scores tie to within the summation order of a pass may
// swap beam slots between the draft's search and this one, which changes nothing about the search itself, so live beam j may sit in ANY slot of the draft
// as long as every live beam is found exactly once: (token, origin) of real beam j == (token, origin) of draft node i with the origin translated
// through the previous step's matching.  vs[8 + j] = the draft node live beam j is matched to: the replay reads beam j's logits from that node's row
// (logit_stats_kernel rowmap).  All matched: vs[0] = steps verified so far.  A live beam the draft does not have - or no draft entry left - : the step
// itself stands (its inputs were verified), but nothing behind it does - done = 2 parks the search (beam_step_kernel and the kernels below return at
// their `done` test) until the host resumes it.
// Cache bookkeeping (book = 1: the steps of a window): the window's pass left node (s, i)'s K / V in slot i (draft index), so a live beam's history is a
// path through slots; per live beam the kernel maintains that path - vs[32 + 32 j + t] = the slot that holds beam j's row of window step t, vs[16 + j] =
// the slot that holds everything before the window - by pulling the parent's path and appending the parent's node; kv_gather_kernel below turns the
// paths into the layout ordinary steps expect (slot j = beam j's whole history) once per window.
// vs layout (ints): 0 steps verified | 2 window steps applied | 8.. matching | 16.. pre-window slot | 32.. paths [8][32]
constexpr int VS_PERM = 8, VS_BASE = 16, VS_PATH = 32, VS_PATH_W = 32, VS_INTS = VS_PATH + MAX_R * VS_PATH_W;
__global__ __launch_bounds__(64) void draft_match_kernel(BeamState bs, const int* __restrict__ draft, int n_draft, int k, int* __restrict__ vs, int book) {
  __shared__ int org_r[MAX_R], tok_r[MAX_R], perm_prev[MAX_R], old_base[MAX_R], old_path[MAX_R][VS_PATH_W];
  const int tid = threadIdx.x;
  if (bs.done[0]) return;                    // (uniform: finished, or parked by an earlier step)
  const int s = bs.step_u[0] - 1;            // the step that just completed
  const int t = vs[2];
  if (tid < k) {
    const int* tr = bs.traj + ((size_t)s * MAX_R + tid) * 2;
    tok_r[tid] = tr[0]; org_r[tid] = tr[1]; perm_prev[tid] = vs[VS_PERM + tid]; old_base[tid] = vs[VS_BASE + tid];
  }
  for (int e = tid; e < k * VS_PATH_W; e += 64) old_path[e / VS_PATH_W][e % VS_PATH_W] = vs[VS_PATH + e];
  __syncthreads();
  if (book) {
    for (int e = tid; e < k * VS_PATH_W; e += 64) {
      const int j = e / VS_PATH_W, u = e % VS_PATH_W, pj = org_r[j];
      if (u < t) vs[VS_PATH + e] = old_path[pj][u];
      else if (u == t) vs[VS_PATH + e] = perm_prev[pj];      // the parent's own row of this step sits in the slot of the draft node it was matched to
    }
    if (tid < k) vs[VS_BASE + tid] = old_base[org_r[tid]];
  }
  if (tid == 0) {
    bool ok = s >= 0 && s < n_draft;
    int pn[MAX_R]; unsigned used = 0;
    if (ok) {
      const int* dr = draft + (size_t)s * MAX_R * 2;
      for (int j = 0; j < k && ok; ++j) {
        const int want_org = perm_prev[org_r[j]];
        int hit = -1;
        for (int i = 0; i < k; ++i) if (!((used >> i) & 1u) && dr[2 * i] == tok_r[j] && dr[2 * i + 1] == want_org) { hit = i; break; }
        if (hit < 0) ok = false; else { pn[j] = hit; used |= 1u << hit; }
      }
    }
    if (ok) { for (int j = 0; j < k; ++j) vs[VS_PERM + j] = pn[j]; vs[0] = s + 1; }
    else bs.done[0] = 2;
    if (book) vs[2] = t + 1;
  }
}
int launch_draft_match(hipStream_t st, const BeamState& bs, const int* draft, int n_draft, int beam, int* vstate, int book) {
  hipLaunchKernelGGL(draft_match_kernel, dim3(1), dim3(64), 0, st, bs, draft, n_draft, beam, vstate, book);
  return WIS_OK;
}
// The paths of draft_match_kernel applied to the cache, once per window: slot j becomes beam j's history - positions before the window from slot
// vs[16 + j], window step u (position w0 + u) from slot vs[32 + 32 j + u], for the vs[2] steps the window's replay completed (a parked search included:
// the step that parked it stands).  grid (8 position slices, 2 L), one thread = one 16-byte column chunk of ALL k slots at a position: loads, waits,
// stores - the in-place gather of kv_reorder_kernel with a parent table per position.
__global__ __launch_bounds__(256) void kv_gather_kernel(f16* __restrict__ kc, f16* __restrict__ vc, size_t lstride, const int* __restrict__ vs, const int* __restrict__ done,
                                                        int k, int w0, int ctx, int d) {
  const int lk = blockIdx.y, tid = threadIdx.x;
  const int nwin = vs[2];
  if (done[0] == 1 || nwin <= 0) return;         // finished inside the window (the cache is not read again) / nothing replayed
  f16* cache = ((lk & 1) ? vc : kc) + (size_t)(lk >> 1) * lstride;
  const int c8 = d >> 3, npos = w0 + nwin;
  for (int p = blockIdx.x; p < npos; p += gridDim.x) {
    int par[MAX_R];
#pragma unroll
    for (int j = 0; j < MAX_R; ++j) par[j] = j < k ? (p < w0 ? vs[VS_BASE + j] : vs[VS_PATH + j * VS_PATH_W + (p - w0)]) : j;
    for (int ch = tid; ch < c8; ch += 256) {
      u32x4 v[MAX_R];
#pragma unroll
      for (int j = 0; j < MAX_R; ++j) if (j < k && par[j] != j) v[j] = *reinterpret_cast<const u32x4*>(cache + ((size_t)par[j] * ctx + p) * d + ch * 8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < MAX_R; ++j) if (j < k && par[j] != j) *reinterpret_cast<u32x4*>(cache + ((size_t)j * ctx + p) * d + ch * 8) = v[j];
    }
  }
}
int launch_kv_gather(hipStream_t st, f16* kc, f16* vc, size_t layer_stride, int L, const int* vstate, const int* done, int beam, int w0, int ctx, int d) {
  hipLaunchKernelGGL(kv_gather_kernel, dim3(8, 2 * L), dim3(256), 0, st, kc, vc, layer_stride, vstate, done, beam, w0, ctx, d);
  return WIS_OK;
}

// =======================================================================================
// Beam reorder of the self-attention KV cache (CTranslate2 gathers its cache tensors by beam_origin after every step): slot j of
// an utterance becomes a copy of slot parent[j] for every layer and every valid position.  grid (8 position slices, 2 L, B);
// thread = one 16-byte column chunk of ALL the utterance's rows at a position: it loads the k parents' chunks, waits, then
// stores - reads and writes of an address meet only inside one thread, so the permutation is safe in place.  Moving
// beam x positions x d bytes per step (16 MB at position 20, large-v2) buys the self-attention kernel a K/V address that
// depends on nothing it has to load: one fabric round trip per layer instead of two.
__global__ __launch_bounds__(256) void kv_reorder_kernel(f16* __restrict__ kc, f16* __restrict__ vc, size_t lstride, const int* __restrict__ parent,
                                                         const int* __restrict__ step_u, const int* __restrict__ done, int k, int P, int ctx, int d) {
  const int b = blockIdx.z, lk = blockIdx.y, tid = threadIdx.x;
  if (k < 2) return;
  const int r0 = b * k;
  // the three control words are requested together (a `done` test in front of the `parent` loads in front of the `step` load is
  // three dependent round trips before the first cache row moves)
  const int dn = done[b], st = step_u[b];
  int par[MAX_R];
#pragma unroll
  for (int j = 0; j < MAX_R; ++j) par[j] = parent[r0 + (j < k ? j : 0)];
  bool ident = true;
#pragma unroll
  for (int j = 0; j < MAX_R; ++j) { if (j < k) ident = ident & (par[j] == r0 + j); else par[j] = r0 + j; }
  if (dn | (int)ident) return;
  const int npos = P - 1 + st;                  // beam_step already counted this step: positions 0 .. npos-1 hold history
  f16* cache = ((lk & 1) ? vc : kc) + (size_t)(lk >> 1) * lstride;
  const int c8 = d >> 3;
  for (int p = blockIdx.x; p < npos; p += gridDim.x) {
    for (int ch = tid; ch < c8; ch += 256) {
      u32x4 v[MAX_R];
#pragma unroll
      for (int j = 0; j < MAX_R; ++j) if (j < k && par[j] != r0 + j) v[j] = *reinterpret_cast<const u32x4*>(cache + ((size_t)par[j] * ctx + p) * d + ch * 8);      // (a slot that continues itself moves nothing: neither read nor written)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every parent chunk is in registers before any row is overwritten
#pragma unroll
      for (int j = 0; j < MAX_R; ++j) if (j < k && par[j] != r0 + j) *reinterpret_cast<u32x4*>(cache + ((size_t)(r0 + j) * ctx + p) * d + ch * 8) = v[j];
    }
  }
}
int launch_kv_reorder(hipStream_t st, f16* kc, f16* vc, size_t layer_stride, int L, const BeamState& bs, int B, int beam, int P, int ctx, int d) {
  if (beam < 2) return WIS_OK;        // greedy: a row always continues itself
  if (beam > MAX_R) { set_error("kv_reorder: beam %d > %d", beam, MAX_R); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(kv_reorder_kernel, dim3(8, 2 * L, B), dim3(256), 0, st, kc, vc, layer_stride, bs.parent, bs.step_u, bs.done, beam, P, ctx, d);
  return WIS_OK;
}

// =======================================================================================
__global__ void lang_probs_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ lang_ids, int n_lang, float* __restrict__ probs) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (size_t)b * ld;
  float mx = -INFINITY;
  for (int i = lane; i < n_lang; i += 64) mx = fmaxf(mx, row[lang_ids[i]]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < n_lang; i += 64) s += __expf(row[lang_ids[i]] - mx);
  s = wave_sum(s);
  for (int i = lane; i < n_lang; i += 64) probs[(size_t)b * n_lang + i] = __expf(row[lang_ids[i]] - mx) / s;
}
int launch_lang_probs(hipStream_t st, const float* logits, int ld, const int* lang_ids, int n_lang, float* probs, int B) {
  hipLaunchKernelGGL(lang_probs_kernel, dim3(B), dim3(64), 0, st, logits, ld, lang_ids, n_lang, probs);
  return WIS_OK;
}

}  // namespace wis
