// dec_kernels.hip — Whisper decoder step, logits processors and beam search on gfx950
// (the reference runs these inside ctranslate2 Whisper.generate,
// call site reference main.py:687-693, semantics SURVEY §3.4 + Appendix C).
//
// A decode step has M = B*beam <= 48 rows: every projection is a weight-streaming, HBM-bound
// skinny GEMM.  MI355X mapping:
//  * weights are re-packed once at load into MFMA A-fragment order ([N/16][K/32][64 lanes][8 f16])
//    so every wave-level load is ONE fully coalesced 1 KiB global_load_dwordx4 that feeds
//    v_mfma_f32_16x16x32_f16 directly (rows of the M<=16 activation block are the B operand, read
//    from LDS); 4 waves split K, partial accumulators meet in LDS, the epilogue (bias, GELU,
//    fp32 residual, KV-cache scatter) is fused.  The pre-LN LayerNorm is fused into the
//    prologue: every workgroup re-normalises the (tiny, L2-resident) M x d activation itself.
//  * self-attention KV cache is never reordered: an ancestry table (anc[slot][pos] -> physical
//    slot) is gathered per step instead of the K/V tensors.
//  * cross-attention K is stored [H][dh/8][T][8] so lane-per-key reads are 16-byte coalesced,
//    V as [H][T][64]; the `beam` query rows of an utterance are folded into one pass over the
//    utterance's K/V (shared per utterance, never tiled per beam); the T axis is split across
//    workgroups with an in-launch, placement-independent last-arriver combine.
//  * suppress masks, log-softmax statistics and the top-2*beam candidates are computed in one
//    pass over the logits (16 chunks per row), the beam bookkeeping runs on device (one wave per
//    utterance) so the host never sees logits and a whole step is CUDA-graph-free replayable.
#include "common.hpp"
#include "kernels.hpp"

namespace wis {

// =======================================================================================
__global__ void pack_gemv_kernel(const f16* __restrict__ W, f16* __restrict__ Wp, int N, int Npad, int K, int n_scale, float scale) {
  const int ksteps = K / 32;
  const size_t total = (size_t)(Npad / 16) * ksteps * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const size_t t = idx >> 6;
    const int ks = (int)(t % ksteps), nt = (int)(t / ksteps);
    const int n = 16 * nt + (lane & 15), k = 32 * ks + 8 * (lane >> 4);
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
int launch_pack_gemv(hipStream_t st, const f16* W, f16* Wp, int N, int Npad, int K, int n_scale, float scale) {
  if (K % 32 || Npad % 16 || Npad < N) { set_error("pack_gemv: bad shape N=%d Npad=%d K=%d", N, Npad, K); return WIS_E_ARG; }
  const size_t total = (size_t)(Npad / 16) * (K / 32) * 64;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_gemv_kernel, dim3(blocks), dim3(256), 0, st, W, Wp, N, Npad, K, n_scale, scale);
  return WIS_OK;
}

// =======================================================================================
// Skinny GEMM.  grid = Npad/16 workgroups of 4 waves; wave w streams the k-steps of its quarter of
// every staged K-chunk.  Dynamic LDS: xs f16 [M][KC+8] | red f32 [4][MB][64][4] | stats f32 [48][2] | sred f32 [4][8]
//
// Memory-latency structure (one dependent round trip): the activation loads (L2-hot, written by the
// previous kernel) are issued FIRST, then a 16-deep prefetch of this wave's weight fragments (HBM);
// vmcnt retires in order, so LayerNorm statistics and the f16 staging of x run from registers while
// the weight stream is in flight, and the MFMAs consume the fragments as they land, refilling the ring.
constexpr int GV_PF = 16;   // weight fragments in flight per wave (16 KiB)

// MODE 0: generic staging from global; 1: fast LayerNorm prologue from registers; 2: fast f16 activations from registers
template <int MB, int MODE>
__global__ __launch_bounds__(256) void gemv_kernel(GemvP p, int KC) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = p.M, K = p.K;
  const int xstr = KC + 8;
  f16* xs = reinterpret_cast<f16*>(smem);
  float* red = reinterpret_cast<float*>(smem + (((size_t)M * xstr * 2 + 15) & ~(size_t)15));
  float* stats = red + 4 * MB * 64 * 4;
  float* sred = stats + 2 * MAX_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockIdx.x;
  const int ksteps = K / 32;
  const int S = KC / 128;                        // k-steps per wave per chunk
  const int ksl0 = wave * S;                     // first chunk-local k-step of this wave
  const u32x4* wp4 = reinterpret_cast<const u32x4*>(p.Wp) + (size_t)nt * ksteps * 64 + lane;
  const int k4n = K >> 2;                        // float4 per row
  const bool ln = p.flags & GV_LN;
  // fast LayerNorm path (host-selected, M <= 8, K <= 2048): the whole M x K activation lives in registers; thread
  // owns float4 columns k4 = tid, tid + 256 of every row (no index arithmetic), statistics are a single shifted pass
  // (c = x[r][0]: var = E[(x-c)^2] - E[x-c]^2) so the block needs ONE reduction + barrier before staging.
  constexpr bool fast = MODE == 1;
  constexpr int RMAX = fast ? 8 : 1;
  float4 xv[RMAX][2], gv[2], bv[2];
  float cshift[RMAX];
  unsigned long long* pf = (blockIdx.x == 0 && tid == 0) ? p.prof : nullptr;
  stamp(pf, 0);
  if (fast) {
    const float4* x4 = reinterpret_cast<const float4*>(p.x);
    const float4* g4 = reinterpret_cast<const float4*>(p.gamma);
    const float4* b4 = reinterpret_cast<const float4*>(p.beta);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k4 = tid + 256 * j;
      if (k4 < k4n) {
        gv[j] = g4[k4]; bv[j] = b4[k4];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) if (r < M) xv[r][j] = x4[(size_t)r * k4n + k4];
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) cshift[r] = (r < M) ? reinterpret_cast<const float*>(p.x)[(size_t)r * K] : 0.f;
  }
  // f16 activations (attention / FFN hidden output of the previous kernel): same idea, up to 13 x 16 B per thread
  constexpr bool fastx = MODE == 2;
  constexpr int NXH = fastx ? 13 : 1;
  u32x4 xh[NXH];
  if (fastx) {
    const u32x4* x8 = reinterpret_cast<const u32x4*>(p.x);
#pragma unroll
    for (int i = 0; i < NXH; ++i) { const int idx = tid + 256 * i; if (idx < M * (K >> 3)) xh[i] = x8[idx]; }
  }
  // weight prefetch for chunk 0 (independent of x)
  u32x4 wf[GV_PF];
  {
    const u32x4* wq = wp4 + (size_t)ksl0 * 64;
#pragma unroll
    for (int u = 0; u < GV_PF; ++u) if (u < S) wf[u] = __builtin_nontemporal_load(wq + (size_t)u * 64);
  }

  stamp(pf, 1);
  if (fast) {
    float sa[RMAX], sb[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      sa[r] = 0.f; sb[r] = 0.f;
      if (r < M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (tid + 256 * j < k4n) {
            const float a = xv[r][j].x - cshift[r], b = xv[r][j].y - cshift[r], c = xv[r][j].z - cshift[r], e = xv[r][j].w - cshift[r];
            sa[r] += (a + b) + (c + e); sb[r] += (a * a + b * b) + (c * c + e * e);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      if (r < M) {
        const float ta = wave_sum(sa[r]), tb = wave_sum(sb[r]);
        if (lane == 0) { sred[wave * 16 + 2 * r] = ta; sred[wave * 16 + 2 * r + 1] = tb; }
      }
    }
    __syncthreads();
    stamp(pf, 2);
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      if (r < M) {
        const float A = ((sred[2 * r] + sred[16 + 2 * r]) + (sred[32 + 2 * r] + sred[48 + 2 * r])) / (float)K;
        const float Bq = ((sred[2 * r + 1] + sred[17 + 2 * r]) + (sred[33 + 2 * r] + sred[49 + 2 * r])) / (float)K;
        const float mu = cshift[r] + A;
        const float rs = 1.0f / sqrtf(fmaxf(Bq - A * A, 0.f) + 1e-5f);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k4 = tid + 256 * j;
          if (k4 < k4n) {
            f16x4 o;
            o[0] = (f16)((xv[r][j].x - mu) * rs * gv[j].x + bv[j].x); o[1] = (f16)((xv[r][j].y - mu) * rs * gv[j].y + bv[j].y);
            o[2] = (f16)((xv[r][j].z - mu) * rs * gv[j].z + bv[j].z); o[3] = (f16)((xv[r][j].w - mu) * rs * gv[j].w + bv[j].w);
            *reinterpret_cast<f16x4*>(xs + (size_t)r * xstr + k4 * 4) = o;
          }
        }
      }
    }
  } else if (fastx) {
    const int c8 = K >> 3;
#pragma unroll
    for (int i = 0; i < NXH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < M * c8) { const int row = idx / c8, k8 = idx - row * c8; *reinterpret_cast<u32x4*>(xs + (size_t)row * xstr + k8 * 8) = xh[i]; }
    }
  } else if (ln) {
    const float* xf = reinterpret_cast<const float*>(p.x);
    for (int r = wave; r < M; r += 4) {
      const float4* x4 = reinterpret_cast<const float4*>(xf + (size_t)r * K);
      float sm = 0.f;
      for (int i = lane; i < k4n; i += 64) { const float4 v = x4[i]; sm += (v.x + v.y) + (v.z + v.w); }
      const float mean = wave_sum(sm) / (float)K;
      float q = 0.f;
      for (int i = lane; i < k4n; i += 64) {
        const float4 v = x4[i];
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
        q += (a * a + b * b) + (c * c + e * e);
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
      if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
    __syncthreads();
  }

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { int r = mb * 16 + (lane & 15); xrow[mb] = (r < M ? r : M - 1) * xstr + 8 * (lane >> 4); }

  const int c8n = KC / 8;
  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    const u32x4* wq = wp4 + (size_t)(kc0 / 32 + ksl0) * 64;
    if (kc0 > 0) {
#pragma unroll
      for (int u = 0; u < GV_PF; ++u) if (u < S) wf[u] = __builtin_nontemporal_load(wq + (size_t)u * 64);
    }
    if (!fast && !fastx) {
      // stage x[:, kc0:kc0+KC] (normalised) as f16
      for (int idx = tid; idx < M * c8n; idx += 256) {
        const int r = idx / c8n, c8 = idx - r * c8n, k = kc0 + c8 * 8;
        f16x8 o;
        if (ln) {
          const float* xr = reinterpret_cast<const float*>(p.x) + (size_t)r * K + k;
          const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
          const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + k), g1 = *reinterpret_cast<const float4*>(p.gamma + k + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(p.beta + k), b1 = *reinterpret_cast<const float4*>(p.beta + k + 4);
          const float mean = stats[2 * r], rstd = stats[2 * r + 1];
          o[0] = (f16)((a.x - mean) * rstd * g0.x + b0.x); o[1] = (f16)((a.y - mean) * rstd * g0.y + b0.y);
          o[2] = (f16)((a.z - mean) * rstd * g0.z + b0.z); o[3] = (f16)((a.w - mean) * rstd * g0.w + b0.w);
          o[4] = (f16)((b.x - mean) * rstd * g1.x + b1.x); o[5] = (f16)((b.y - mean) * rstd * g1.y + b1.y);
          o[6] = (f16)((b.z - mean) * rstd * g1.z + b1.z); o[7] = (f16)((b.w - mean) * rstd * g1.w + b1.w);
        } else {
          o = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p.x) + (size_t)r * K + k);
        }
        *reinterpret_cast<f16x8*>(xs + (size_t)r * xstr + c8 * 8) = o;
      }
    }
    __syncthreads();
    stamp(pf, 3);
    for (int base = 0; base < S; base += GV_PF) {
#pragma unroll
      for (int u = 0; u < GV_PF; ++u) {
        if (base + u < S) {
          const f16x8 a = *reinterpret_cast<const f16x8*>(&wf[u]);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const f16x8 xb = *reinterpret_cast<const f16x8*>(xs + xrow[mb] + (ksl0 + base + u) * 32);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb, acc[mb], 0, 0, 0);
          }
          if (base + u + GV_PF < S) wf[u] = __builtin_nontemporal_load(wq + (size_t)(base + u + GV_PF) * 64);
        }
      }
    }
    __syncthreads();
  }
  stamp(pf, 4);
  // cross-wave reduction; D[i = n][j = m]: lane holds m = lane&15, n = 4*(lane>>4) + r
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    *reinterpret_cast<float4*>(red + ((size_t)(wave * MB + mb) * 64 + lane) * 4) = make_float4(acc[mb][0], acc[mb][1], acc[mb][2], acc[mb][3]);
  __syncthreads();
  stamp(pf, 5);
  if (tid < MB * 64) {
    const int mb = tid >> 6, ln = tid & 63;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + ((size_t)(w * MB + mb) * 64 + ln) * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int m = mb * 16 + (ln & 15), n = 16 * nt + 4 * (ln >> 4);
    if (m < M && n < p.N) {
      if (p.bias) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + n); s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w; }
      if (p.flags & GV_QKV) {
        const int d = p.d;
        if (n < d) {
          *reinterpret_cast<float4*>(p.q + (size_t)m * d + n) = s;
        } else {
          const bool isk = n < 2 * d;
          f16* dst = (isk ? p.kc : p.vc) + ((size_t)p.slot[m] * p.ctx + p.pos[m]) * d + (n - (isk ? d : 2 * d));
          const f16x4 o = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(dst) = o;
        }
      } else {
        if (p.flags & GV_GELU) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
        const size_t o = (size_t)m * p.N + n;
        if (p.flags & GV_RESID) {
          float4* y = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o);
          const float4 t = *y; *y = make_float4(t.x + s.x, t.y + s.y, t.z + s.z, t.w + s.w);
        } else if (p.flags & GV_OUT_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = s;
        } else {
          const f16x4 h = {(f16)s.x, (f16)s.y, (f16)s.z, (f16)s.w};
          *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + o) = h;
        }
      }
    }
  }
  stamp(pf, 6);
}

int launch_gemv(hipStream_t st, const GemvP& p) {
  if (p.M < 1 || p.M > MAX_ROWS || p.K % 128 || p.N % 4) { set_error("gemv: M=%d N=%d K=%d unsupported", p.M, p.N, p.K); return WIS_E_UNSUPPORTED; }
  const int MB = cdiv(p.M, 16);
  // largest K-chunk (multiple of 128 dividing K) whose f16 image of M rows fits 64 KiB
  int KC = p.K;
  const size_t aux = (size_t)4 * MB * 64 * 16 + MAX_ROWS * 8 + 4 * 16 * 4 + 16;   // red + stats + sred (+ alignment)
  while ((size_t)p.M * (KC + 8) * 2 + aux > 65536) {
    int next = 0;
    for (int c = KC - 128; c >= 128; c -= 128) if (p.K % c == 0) { next = c; break; }
    if (!next) { set_error("gemv: cannot chunk K=%d for M=%d", p.K, p.M); return WIS_E_UNSUPPORTED; }
    KC = next;
  }
  const size_t lds = (((size_t)p.M * (KC + 8) * 2 + 15) & ~(size_t)15) + aux;
  const int npad = cdiv(p.N, 16) * 16;
  dim3 grid(npad / 16), block(256);
  int mode = 0;
  if (KC == p.K) {
    if ((p.flags & GV_LN) && p.M <= 8 && p.K <= 2048) mode = 1;
    else if (!(p.flags & GV_LN) && p.M * (p.K / 8) <= 13 * 256) mode = 2;
  }
#define WIS_GV(MBv, MODEv) hipLaunchKernelGGL((gemv_kernel<MBv, MODEv>), grid, block, lds, st, p, KC)
  if (MB == 1) { if (mode == 1) WIS_GV(1, 1); else if (mode == 2) WIS_GV(1, 2); else WIS_GV(1, 0); }
  else if (MB == 2) { if (mode == 2) WIS_GV(2, 2); else WIS_GV(2, 0); }
  else { if (mode == 2) WIS_GV(3, 2); else WIS_GV(3, 0); }
#undef WIS_GV
  return WIS_OK;
}

// =======================================================================================
// x[m] = E[tok[m]] + pos_emb[pos[m]]   (no embedding scale; learned positions) -> fp32
__global__ void dec_embed_kernel(const f16* __restrict__ emb, const f16* __restrict__ pos_emb, const int* __restrict__ tok,
                                 const int* __restrict__ pos, float* __restrict__ x, int d) {
  const int m = blockIdx.x;
  const f16* e = emb + (size_t)tok[m] * d;
  const f16* pe = pos_emb + (size_t)pos[m] * d;
  for (int i = threadIdx.x; i < d; i += blockDim.x) x[(size_t)m * d + i] = (float)e[i] + (float)pe[i];
}
int launch_dec_embed(hipStream_t st, const f16* emb, const f16* pos_emb, const int* tok, const int* pos, float* x, int M, int d) {
  hipLaunchKernelGGL(dec_embed_kernel, dim3(M), dim3(256), 0, st, emb, pos_emb, tok, pos, x, d);
  return WIS_OK;
}

// =======================================================================================
// causal self-attention of one new token per row over its cached history.  grid (M, H), block 64.
// The logical slot of row m is arithmetic ((m / rpu) * sstride + (m % rpu) * rmul: decode rows own their slot,
// prefill rows share the utterance's first slot), so the dependent chain is two round trips:
// {pos, ancestry} -> {K rows (lane = position), V prefetch (lane = dh) for the first 32 positions}.
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const float* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc,
                                                           const int* __restrict__ anc, const int* __restrict__ pos,
                                                           f16* __restrict__ out, int d, int ctx, int rpu, int sstride, int rmul,
                                                           unsigned long long* prof) {
  __shared__ float sq[64];
  __shared__ float sp[512];
  __shared__ int sa[512];
  const int m = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  unsigned long long* pf = (m == 0 && h == 0 && lane == 0) ? prof : nullptr;
  stamp(pf, 0);
  const int ls = (m / rpu) * sstride + (m % rpu) * rmul;
  const int* arow = anc + (size_t)ls * ctx;
  int a0[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int p = lane + 64 * i; a0[i] = (p < ctx) ? arow[p] : 0; }   // issued before len is known
  const int len = pos[m] + 1;
  sq[lane] = q[(size_t)m * d + h * 64 + lane];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int p = lane + 64 * i; if (p < len) sa[p] = a0[i]; }
  __syncthreads();
  stamp(pf, 1);
  // V prefetch: lane = dh, first 32 positions
  const f16* vbase = vc + h * 64 + lane;
  f16 vreg[32];
#pragma unroll
  for (int pp = 0; pp < 32; ++pp) if (pp < len) vreg[pp] = vbase[((size_t)sa[pp] * ctx + pp) * d];
  float mx = -INFINITY;
  for (int p = lane; p < len; p += 64) {
    const f16* kr = kc + ((size_t)sa[p] * ctx + p) * d + h * 64;
    f16x8 kv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kv[c] = *reinterpret_cast<const f16x8*>(kr + 8 * c);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) dot = fmaf((float)kv[c][j], sq[8 * c + j], dot);
    sp[p] = dot; mx = fmaxf(mx, dot);
  }
  stamp(pf, 2);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int p = lane; p < len; p += 64) { const float e = __expf(sp[p] - mx); sp[p] = e; sum += e; }
  sum = wave_sum(sum);
  __syncthreads();
  stamp(pf, 3);
  float acc = 0.f;
#pragma unroll
  for (int pp = 0; pp < 32; ++pp) if (pp < len) acc = fmaf(sp[pp], (float)vreg[pp], acc);
  for (int p = 32; p < len; ++p) acc = fmaf(sp[p], (float)vbase[((size_t)sa[p] * ctx + p) * d], acc);
  out[(size_t)m * d + h * 64 + lane] = (f16)(acc / sum);
  stamp(pf, 4);
}
int launch_dec_self_attn(hipStream_t st, const float* q, const f16* kc, const f16* vc, const int* anc, const int* pos, f16* out,
                         int M, int H, int d, int ctx, int rpu, int sstride, int rmul, unsigned long long* prof) {
  if (ctx > 512 || ctx < 64) { set_error("dec_self_attn: ctx=%d outside [64, 512]", ctx); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(dec_self_attn_kernel, dim3(M, H), dim3(64), 0, st, q, kc, vc, anc, pos, out, d, ctx, rpu, sstride, rmul, prof);
  return WIS_OK;
}

// =======================================================================================
// cross-attention of the R query rows of one utterance over a chunk of <= 256 encoder keys.
// grid (chunks, H, B), block 256.  One memory round trip: every K load (lane = key, 8 coalesced 16-byte
// dh-groups) and every V load (lane = (key%32 slot, dh-group), 1 KiB per wave instruction) is issued up front;
// scores / softmax / P.V then run from registers + LDS.  The chunk partials (o[64], max, sum) are published with
// write-through (sc1) relaxed agent-scope stores and combined by the last-arriving workgroup, which reads them
// back with sc1 loads: placement independent, no fences (guide §6 G16, "8-B/4-B agent atomics both sides").
typedef __attribute__((address_space(1))) unsigned gu32;
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int R>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const float* __restrict__ q, const f16* __restrict__ kx, const f16* __restrict__ vx,
                                                             f16* __restrict__ out, float* part, unsigned* counters,
                                                             int H, int d, int T, int C, int CL, unsigned long long* prof) {
  __shared__ float ssc[R][256];
  __shared__ float sq[R][64];
  __shared__ float sred[4][R];
  __shared__ float so[4][R][64];
  __shared__ float smx[R], ssum[R];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int klo = c * CL, khi = (klo + CL < T) ? klo + CL : T, n = khi - klo;   // n <= 256
  unsigned long long* pf = (c == 0 && h == 0 && b == 0 && tid == 0) ? prof : nullptr;
  stamp(pf, 0);

  // ---- all loads first
  const f16* kb = kx + (size_t)(b * H + h) * 8 * T * 8;
  u32x4 kr[8];
  if (tid < n) {
#pragma unroll
    for (int g = 0; g < 8; ++g) kr[g] = *reinterpret_cast<const u32x4*>(kb + ((size_t)g * T + klo + tid) * 8);
  }
  const int g8 = tid & 7, kq = tid >> 3;
  const f16* vb = vx + (size_t)(b * H + h) * T * 64 + 8 * g8;
  u32x4 vr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int key = klo + kq + 32 * i; if (key < khi) vr[i] = *reinterpret_cast<const u32x4*>(vb + (size_t)key * 64); }
  for (int i = tid; i < R * 64; i += 256) sq[i >> 6][i & 63] = q[(size_t)(b * R + (i >> 6)) * d + h * 64 + (i & 63)];
  __syncthreads();
  stamp(pf, 1);

  // ---- scores: one key per lane
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) s[r] = -INFINITY;
  if (tid < n) {
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f16x8 kv = *reinterpret_cast<const f16x8*>(&kr[g]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float kf = (float)kv[j];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = fmaf(kf, sq[r][8 * g + j], s[r]);
      }
    }
  }
  stamp(pf, 2);
#pragma unroll
  for (int r = 0; r < R; ++r) { const float v = wave_max(s[r]); if (lane == 0) sred[wave][r] = v; }
  __syncthreads();
  if (tid < R) smx[tid] = fmaxf(fmaxf(sred[0][tid], sred[1][tid]), fmaxf(sred[2][tid], sred[3][tid]));
  __syncthreads();
  float e[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { e[r] = (tid < n) ? __expf(s[r] - smx[r]) : 0.f; ssc[r][tid] = e[r]; }
#pragma unroll
  for (int r = 0; r < R; ++r) { const float v = wave_sum(e[r]); if (lane == 0) sred[wave][r] = v; }
  __syncthreads();
  if (tid < R) ssum[tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
  stamp(pf, 3);

  // ---- P.V from the prefetched V registers
  float o[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) o[r][j] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int kk = kq + 32 * i;
    if (kk < n) {
      const f16x8 vv = *reinterpret_cast<const f16x8*>(&vr[i]);
      float vf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vf[j] = (float)vv[j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float pr = ssc[r][kk];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[r][j] = fmaf(pr, vf[j], o[r][j]);
      }
    }
  }
  stamp(pf, 4);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = o[r][j];
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      if (lane < 8) so[wave][r][8 * lane + j] = v;
    }
  __syncthreads();
  stamp(pf, 5);
  if (C == 1) {
    for (int i = tid; i < R * 64; i += 256) {
      const int r = i >> 6, dh = i & 63;
      const float v = (so[0][r][dh] + so[1][r][dh]) + (so[2][r][dh] + so[3][r][dh]);
      out[(size_t)(b * R + r) * d + h * 64 + dh] = (f16)(v / ssum[r]);
    }
    return;
  }
  // ---- split-T: publish the partial with write-through stores; the last-arriving workgroup combines
  float* pbase = part + ((size_t)(b * H + h) * C) * R * 66;
  for (int i = tid; i < R * 64; i += 256) {
    const int r = i >> 6, dh = i & 63;
    const float v = (so[0][r][dh] + so[1][r][dh]) + (so[2][r][dh] + so[3][r][dh]);
    float* pp = pbase + ((size_t)c * R + r) * 66;
    st_sc1(pp + dh, v);
    if (dh == 0) { st_sc1(pp + 64, smx[r]); st_sc1(pp + 65, ssum[r]); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  stamp(pf, 6);
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counters + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (prev == (unsigned)(C - 1));
    if (last) __hip_atomic_store(counters + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    s_last = last;
  }
  __syncthreads();
  stamp(pf, 7);
  if (!s_last) return;
  for (int i = tid; i < R * 64; i += 256) {
    const int r = i >> 6, dh = i & 63;
    float M_ = -INFINITY;
    for (int cc = 0; cc < C; ++cc) M_ = fmaxf(M_, ld_sc1(pbase + ((size_t)cc * R + r) * 66 + 64));
    float L = 0.f, O = 0.f;
    for (int cc = 0; cc < C; ++cc) {
      const float* pp = pbase + ((size_t)cc * R + r) * 66;
      const float w = __expf(ld_sc1(pp + 64) - M_);
      L = fmaf(ld_sc1(pp + 65), w, L); O = fmaf(ld_sc1(pp + dh), w, O);
    }
    out[(size_t)(b * R + r) * d + h * 64 + dh] = (f16)(O / L);
  }
}

int launch_dec_cross_attn(hipStream_t st, const float* q, const f16* kx, const f16* vx, f16* out, float* part, unsigned* counters,
                          int B, int R, int H, int d, int T, int chunks, unsigned long long* prof) {
  if (R < 1 || R > MAX_R || chunks < 1) { set_error("dec_cross_attn: R=%d chunks=%d unsupported", R, chunks); return WIS_E_UNSUPPORTED; }
  const int CL = cdiv(T, chunks);
  if (CL > 256 || chunks > 16) { set_error("dec_cross_attn: %d chunks of %d keys unsupported (<= 16 chunks of <= 256 keys)", chunks, CL); return WIS_E_UNSUPPORTED; }
  dim3 grid(chunks, H, B), block(256);
#define WIS_CA(RR) case RR: hipLaunchKernelGGL(dec_cross_attn_kernel<RR>, grid, block, 0, st, q, kx, vx, out, part, counters, H, d, T, chunks, CL, prof); break;
  switch (R) { WIS_CA(1) WIS_CA(2) WIS_CA(3) WIS_CA(4) WIS_CA(5) WIS_CA(6) WIS_CA(7) WIS_CA(8) }
#undef WIS_CA
  return WIS_OK;
}

// =======================================================================================
// logits processors + log-softmax statistics + per-chunk top-n_cand.  grid (STAT_CHUNKS, M), block 256
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

__global__ __launch_bounds__(256) void logit_stats_kernel(const float* __restrict__ logits, const float* __restrict__ bias_all,
                                                          const float* __restrict__ bias_begin, const int* __restrict__ step_u,
                                                          float* __restrict__ st_max, float* __restrict__ st_sum,
                                                          float* __restrict__ st_val, int* __restrict__ st_idx, SampleCfg cfg) {
  constexpr int PT = 16;   // values per thread: supports n_vocab <= 16*256*16
  __shared__ float sv[4]; __shared__ int si[4];
  __shared__ float s_bv; __shared__ int s_bi;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x, m = blockIdx.y, b = m / cfg.beam;
  const int step = step_u[b];
  const int CHL = cdiv(cfg.n_vocab, STAT_CHUNKS);
  const int lo = c * CHL, hi = (lo + CHL < cfg.n_vocab) ? lo + CHL : cfg.n_vocab;
  const float* row = logits + (size_t)m * cfg.n_vocab_pad;
  const bool first = (step == 0) && cfg.suppress_blank;
  const bool mask_eot = cfg.fixed_new > 0 && step < cfg.fixed_new;
  const bool force_eot = cfg.fixed_new > 0 && step >= cfg.fixed_new;

  float vals[PT];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int nidx = lo + tid + 256 * i;
    float v = -INFINITY;
    if (nidx < hi) {
      v = row[nidx];
      if (bias_all) v += bias_all[nidx];
      if (first) v += bias_begin[nidx];
      if (mask_eot && nidx == cfg.eot) v = -INFINITY;
      if (force_eot && nidx != cfg.eot) v = -INFINITY;
    }
    vals[i] = v; mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if (lane == 0) sv[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
  float sum = 0.f;
  if (mx > -INFINITY) {
#pragma unroll
    for (int i = 0; i < PT; ++i) sum += __expf(vals[i] - mx);   // exp(-inf) = 0 for masked / out of range
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (lane == 0) sv[wave] = sum;
  __syncthreads();
  if (tid == 0) { st_max[m * STAT_CHUNKS + c] = mx; st_sum[m * STAT_CHUNKS + c] = (sv[0] + sv[1]) + (sv[2] + sv[3]); }

  // top-n_cand of the chunk in (value desc, index asc) order: n_cand rounds of block arg-best
  float pv = INFINITY; int pi = -1;   // previous pick
  for (int rnd = 0; rnd < cfg.n_cand; ++rnd) {
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int nidx = lo + tid + 256 * i;
      if (nidx < hi) {
        const float v = vals[i];
        // strictly after the previous pick in the order, and better than the current best
        if (better(pv, pi, v, nidx) && better(v, nidx, bv, bi)) { bv = v; bi = nidx; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float fv = sv[0]; int fi = si[0];
      for (int w = 1; w < 4; ++w) if (better(sv[w], si[w], fv, fi)) { fv = sv[w]; fi = si[w]; }
      s_bv = fv; s_bi = fi;
      st_val[((size_t)m * STAT_CHUNKS + c) * cfg.n_cand + rnd] = fv;
      st_idx[((size_t)m * STAT_CHUNKS + c) * cfg.n_cand + rnd] = fi;
    }
    __syncthreads();
    pv = s_bv; pi = s_bi;
  }
}
int launch_logit_stats(hipStream_t st, const float* logits, const float* bias_all, const float* bias_begin, const int* step_u,
                       float* st_max, float* st_sum, float* st_val, int* st_idx, int B, const SampleCfg& cfg) {
  if (cdiv(cfg.n_vocab, STAT_CHUNKS) > 16 * 256) { set_error("logit_stats: vocab too large"); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(logit_stats_kernel, dim3(STAT_CHUNKS, B * cfg.beam), dim3(256), 0, st, logits, bias_all, bias_begin, step_u,
                     st_max, st_sum, st_val, st_idx, cfg);
  return WIS_OK;
}

// =======================================================================================
// beam search bookkeeping (CTranslate2 4.1.0 BeamSearch::search semantics).
// grid B, block 64 (one wave per utterance).
__global__ __launch_bounds__(64) void beam_step_kernel(const float* __restrict__ st_max, const float* __restrict__ st_sum,
                                                       const float* __restrict__ st_val, const int* __restrict__ st_idx,
                                                       BeamState bs, RowMeta rm, int P, int ctx, SampleCfg cfg) {
  constexpr int POOL_MAX = MAX_R * STAT_CHUNKS * MAX_CAND;   // 2048
  __shared__ float pool_v[POOL_MAX]; __shared__ int pool_id[POOL_MAX];
  __shared__ float lse[MAX_R];
  __shared__ float cand_v[MAX_CAND]; __shared__ int cand_word[MAX_CAND]; __shared__ int cand_org[MAX_CAND];
  __shared__ int nb_src[MAX_R]; __shared__ int nb_tok[MAX_R]; __shared__ float nb_cum[MAX_R];
  __shared__ int hyp_src[MAX_R]; __shared__ int hyp_slot[MAX_R]; __shared__ int hyp_n[MAX_R]; __shared__ int n_newhyp;
  __shared__ int s_finished;
  __shared__ int sh_anc[MAX_R * 512];
  __shared__ int sh_alive[MAX_R * 256];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (bs.done[b]) return;
  const int k = cfg.beam, NC = cfg.n_cand, V = cfg.n_vocab;
  const int step = bs.step_u[b];
  const int r0 = b * k;

  // log-softmax normaliser per live row
  if (lane < k) {
    const int m = r0 + lane;
    float M_ = -INFINITY;
    for (int c = 0; c < STAT_CHUNKS; ++c) M_ = fmaxf(M_, st_max[m * STAT_CHUNKS + c]);
    float S = 0.f;
    for (int c = 0; c < STAT_CHUNKS; ++c) {
      const float mc = st_max[m * STAT_CHUNKS + c];
      if (mc > -INFINITY) S += st_sum[m * STAT_CHUNKS + c] * __expf(mc - M_);
    }
    lse[lane] = M_ + logf(S);
  }
  __syncthreads();
  // candidate pool: score = logit - lse + cum ; flat id = beam * V + token
  const int pool_n = k * STAT_CHUNKS * NC;
  for (int i = lane; i < pool_n; i += 64) {
    const int j = i / (STAT_CHUNKS * NC), m = r0 + j;
    const size_t src = (size_t)m * STAT_CHUNKS * NC + (i - j * STAT_CHUNKS * NC);
    const float v = st_val[src];
    pool_v[i] = (v > -INFINITY) ? (v - lse[j]) + bs.cum[m] : -INFINITY;
    int tk = st_idx[src]; if (tk > V - 1) tk = V - 1;   // exhausted chunks report INT_MAX with -inf
    pool_id[i] = j * V + tk;
  }
  __syncthreads();
  // top-NC of the pool, (score desc, flat id asc)
  float pv = INFINITY; int pi = -1;
  for (int rnd = 0; rnd < NC; ++rnd) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < pool_n; i += 64) {
      const float v = pool_v[i]; const int id = pool_id[i];
      if (better(pv, pi, v, id) && better(v, id, bv, bi)) { bv = v; bi = id; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { cand_v[rnd] = bv; cand_word[rnd] = bi % V; cand_org[rnd] = bi / V; }
    pv = bv; pi = bi;
  }
  __syncthreads();

  // serial bookkeeping
  if (lane == 0) {
    const bool is_last = (step + 1 >= cfg.max_new);
    int nh = bs.n_hyp[b], second = k, newh = 0; bool top_finished = false;
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
    if (!fin) fin = cfg.allow_early_exit ? (top_finished && nh >= cfg.max_candidates) : (nh >= cfg.max_candidates);
    s_finished = fin ? 1 : 0;
  }
  __syncthreads();

  // stage this utterance's token histories and ancestry rows, then write the permuted rows back
  const int hist = step;                 // tokens already in alive[]
  const int npos = P - 1 + step + 1;     // cache positions valid after this step
  for (int i = lane; i < k * hist; i += 64) { const int j = i / hist, t = i - j * hist; sh_alive[j * 256 + t] = bs.alive[(size_t)(r0 + j) * cfg.max_new + t]; }
  for (int i = lane; i < k * npos; i += 64) { const int j = i / npos, p = i - j * npos; sh_anc[j * 512 + p] = bs.anc[(size_t)(r0 + j) * ctx + p]; }
  __syncthreads();
  // finished hypotheses of this step
  for (int hh = 0; hh < n_newhyp; ++hh) {
    const int kk = hyp_src[hh], org = cand_org[kk], n = hyp_n[hh];
    int* dst = bs.hyp_tok + ((size_t)b * cfg.max_hyp + hyp_slot[hh]) * cfg.max_new;
    for (int t = lane; t < n; t += 64) dst[t] = (t < hist) ? sh_alive[org * 256 + t] : cand_word[kk];
  }
  if (s_finished) {
    __syncthreads();
    __threadfence_block();
    if (lane == 0) {
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
      atomicAdd(bs.all_done, 1);
    }
    __syncthreads();
    const int best = cand_org[0], n = bs.hyp_len[b * cfg.max_hyp + best];
    const int* src = bs.hyp_tok + ((size_t)b * cfg.max_hyp + best) * cfg.max_new;
    for (int t = lane; t < n; t += 64) bs.out_ids[(size_t)b * cfg.max_new + t] = src[t];
    return;
  }
  // next live beams
  for (int j = 0; j < k; ++j) {
    const int org = cand_org[nb_src[j]];
    int* al = bs.alive + (size_t)(r0 + j) * cfg.max_new;
    for (int t = lane; t < hist; t += 64) al[t] = sh_alive[org * 256 + t];
    int* an = bs.anc + (size_t)(r0 + j) * ctx;
    for (int p = lane; p < npos; p += 64) an[p] = sh_anc[org * 512 + p];
    if (lane == 0) {
      al[hist] = nb_tok[j];
      if (npos < ctx) an[npos] = r0 + j;           // the next step writes its own K/V at position npos
      bs.cum[r0 + j] = nb_cum[j];
      rm.tok[r0 + j] = nb_tok[j];
      rm.pos[r0 + j] = npos;
    }
  }
  if (lane == 0) bs.step_u[b] = step + 1;
}
int launch_beam_step(hipStream_t st, const float* st_max, const float* st_sum, const float* st_val, const int* st_idx,
                     const BeamState& bs, const RowMeta& rm, int B, int P, int ctx, const SampleCfg& cfg) {
  if (cfg.beam > MAX_R || cfg.n_cand > MAX_CAND || cfg.max_new > 256 || ctx > 512) { set_error("beam_step: config out of range"); return WIS_E_UNSUPPORTED; }
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, st, st_max, st_sum, st_val, st_idx, bs, rm, P, ctx, cfg);
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
