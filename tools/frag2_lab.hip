// Diagnostic bench for the two-n-tile skinny GEMM (csrc/dec_kernels.hip gemv_frag2_kernel: round 3 kept it off because its
// five-row-block instantiation - 168 VGPRs, 40 KiB LDS: three workgroups per CU - returned sporadically wrong tiles at 80 rows x
// 51872 columns), and a first stress of the one-tile kernel.  THIS is how round 4 found the cause: built with the library's round-3
// flags (SLP vectorisation ON: packed-f32 epilogue arithmetic) the `f2` variant below fails on every box it was run on (12-22 of every
// 20-40 launches; features 12 and 14 = the low halves of the v_pk_*_f32 results in lanes 48-63); built with
// -fno-slp-vectorize (same 168 registers, same occupancy) or with AGPR accumulators (two waves per SIMD) it is clean on the same boxes.
//
// What it does (80 rows = 5 row blocks, K = 1280, N = 51872, deterministic pseudo-random operands generated on the device):
//   1. reference = the shipped gemv_frag_kernel<5, 6> alone on the GPU (same MFMA order per wave and same cross-wave sum order as the
//      two-tile kernel, so the two agree BIT FOR BIT when both are right);
//   2. every variant below is launched `iters` times; each result is compared with the reference on the device, word by word;
//      a differing word is recorded with what was read;
//   3. the debug copy of the kernel (frag2_dbg) additionally writes out its LDS exchange buffer after the epilogue and the hardware
//      ids (XCC, CU, SIMD, wave slot) of its waves, so that a wrong tile can be traced to "wrong in LDS" vs "wrong after LDS" and to a
//      place on the chip.
// Variants: f2 (as in the library), f2+24K (24 KiB of unused dynamic LDS: two workgroups per CU), f2dbg, f1 (shipped one-tile kernel
// against its own first launch, concurrently on 4 streams).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include -I willow-inference-server_amd/csrc \
//        -o tools/bin/frag2_lab tools/frag2_lab.hip        (variants of the flags are the experiment: see tools/gpu_session.sh frag2)
#include "../willow-inference-server_amd/csrc/dec_kernels.hip"
#include <cstdarg>
#include <cstring>
#include <vector>

namespace wis {
static thread_local char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
const char* get_error() { return g_err; }
}  // namespace wis
using namespace wis;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash_u(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void fill_f16(f16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (f16)(((float)(hash_u((unsigned)i * 2654435761u + seed) & 0xffff) / 32768.0f - 1.0f) * scale);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = ((float)(hash_u((unsigned)i * 2654435761u + seed) & 0xffff) / 32768.0f - 1.0f) * scale + offset;
}
struct Rec { unsigned idx, got, exp, pad; };
__global__ void compare_words(const unsigned* a, const unsigned* ref, size_t n, unsigned* count, Rec* recs, int max_recs) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned x = a[i], y = ref[i];
    if (x != y) { const unsigned k = atomicAdd(count, 1u); if ((int)k < max_recs) recs[k] = Rec{(unsigned)i, x, y, 0u}; }
  }
}

// ---- debug copy of gemv_frag2_kernel<5, 4, false> (GV_LN | GV_OUT_F32 only): the same statements in the same order, plus at the very
// end (after every output store) the LDS dump and the hardware ids
template <int MB, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void frag2_dbg(GemvP p, float* lds_dump, unsigned* hwid) {      // (three waves per SIMD like the library kernel: 168 VGPRs)
  typedef typename WFrag<false>::T WT;
  constexpr int EPN = (MB + 3) / 4;
  __shared__ __attribute__((aligned(16))) float red[4 * MB * 2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt0 = 2 * blockIdx.x;
  const int M = p.M, K = p.K, ksteps = K >> 5, S = ksteps >> 2;
  const WT* wq0 = reinterpret_cast<const WT*>(p.Wp) + ((size_t)nt0 * ksteps + (size_t)wave * S) * 64 + lane;
  const WT* wq1 = wq0 + (size_t)ksteps * 64;
  const u32x4* xq = reinterpret_cast<const u32x4*>(p.x) + (size_t)wave * S * MB * 64 + lane;
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
  f32x4 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { acc[mb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int base = 0; base < S; base += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (base + u < S) {
        const f16x8 av0 = WFrag<false>::cvt(a0[u]), av1 = WFrag<false>::cvt(a1[u]);
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
  const int l15 = lane & 15, kq = lane >> 4;
  const int nq = K >> 6;
  float4 ep_bias[2], ep_cs[2];
  int ep_n[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    ep_n[nb] = 16 * (nt0 + nb) + 4 * kq;
    ep_bias[nb] = make_float4(0.f, 0.f, 0.f, 0.f); ep_cs[nb] = ep_bias[nb];
    if (ep_n[nb] < p.N) {
      if (p.bias) ep_bias[nb] = *reinterpret_cast<const float4*>(p.bias + ep_n[nb]);
      ep_cs[nb] = *reinterpret_cast<const float4*>(p.csum + ep_n[nb]);
    }
  }
  bool ep_act[EPN]; int ep_m[EPN]; float s1[EPN], s2[EPN], sc0[EPN];
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    const int mb = wave + 4 * e;
    ep_act[e] = mb < MB; ep_m[e] = mb * 16 + l15; s1[e] = 0.f; s2[e] = 0.f; sc0[e] = 0.f;
    if (ep_act[e]) {
      const int mm = ep_m[e] < M ? ep_m[e] : M - 1;
      const float2* row = reinterpret_cast<const float2*>(p.stat_in) + (size_t)mm * (K >> 4);
      const float2* sp = row + (size_t)kq * nq;
      const float c = row[0].x * 0.0625f;
      sc0[e] = c;
#pragma unroll 4
      for (int i = 0; i < nq; ++i) { const float2 v = sp[i]; const float dm = v.x * 0.0625f - c; s1[e] += dm; s2[e] += v.y + 16.0f * dm * dm; }
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
      *reinterpret_cast<float4*>(red + ((size_t)((wave * MB + mb) * 2 + nb) * 64 + lane) * 4) = make_float4(acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EPN; ++e) {
    if (!ep_act[e]) continue;
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
      s.x = rs * (s.x - mu * ep_cs[nb].x) + ep_bias[nb].x; s.y = rs * (s.y - mu * ep_cs[nb].y) + ep_bias[nb].y;
      s.z = rs * (s.z - mu * ep_cs[nb].z) + ep_bias[nb].z; s.w = rs * (s.w - mu * ep_cs[nb].w) + ep_bias[nb].w;
      const int n = ep_n[nb];
      if (m >= M || n >= p.N) continue;
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (size_t)m * p.N + n) = s;
    }
  }
  // ---- diagnostics (behind every product statement)
  __syncthreads();
  if (lds_dump) {
    float* dst = lds_dump + (size_t)blockIdx.x * (4 * MB * 2 * 64 * 4);
    for (int i = tid; i < 4 * MB * 2 * 64; i += 256) *reinterpret_cast<float4*>(dst + (size_t)i * 4) = *reinterpret_cast<const float4*>(red + (size_t)i * 4);
  }
  if (hwid && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    hwid[((size_t)blockIdx.x * 4 + wave) * 2] = hw; hwid[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = xcc;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int M = argc > 2 ? atoi(argv[2]) : 80;
  const int K = 1280, N = 51872, MB = (M + 15) / 16;
  if (MB != 5) { printf("this lab is instantiated for 5 row blocks (65..80 rows)\n"); return 1; }
  hipStream_t st; CK(hipStreamCreate(&st));
  f16 *Wp, *xf; float *stat, *csum, *bias, *yref, *y, *dump, *dump_ref; unsigned *cnt, *hwid; Rec* recs;
  const size_t nW = (size_t)N * K, nx = (size_t)(K / 32) * MB * 64 * 8, nstat = (size_t)M * (K / 16) * 2, ny = (size_t)M * N;
  const size_t ndump = (size_t)(N / 32) * 4 * MB * 2 * 64 * 4;
  CK(hipMalloc(&Wp, nW * 2)); CK(hipMalloc(&xf, nx * 2)); CK(hipMalloc(&stat, nstat * 4)); CK(hipMalloc(&csum, (size_t)N * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
  CK(hipMalloc(&yref, ny * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&dump, ndump * 4)); CK(hipMalloc(&dump_ref, ndump * 4));
  CK(hipMalloc(&cnt, 16)); CK(hipMalloc(&hwid, (size_t)(N / 32) * 4 * 2 * 4)); CK(hipMalloc(&recs, 4096 * sizeof(Rec)));
  fill_f16<<<2048, 256, 0, st>>>(Wp, nW, 11u, 0.05f);
  fill_f16<<<256, 256, 0, st>>>(xf, nx, 23u, 2.0f);
  fill_f32<<<256, 256, 0, st>>>(stat, nstat, 37u, 4.0f, 8.0f);       // (sum, M2) pairs: positive M2 on average; the values only have to be the same for every variant
  fill_f32<<<256, 256, 0, st>>>(csum, (size_t)N, 41u, 0.3f, 0.f);
  fill_f32<<<256, 256, 0, st>>>(bias, (size_t)N, 43u, 1.0f, 0.f);
  CK(hipStreamSynchronize(st));
  GemvP g; memset(&g, 0, sizeof(g));
  g.x = xf; g.Wp = Wp; g.bias = bias; g.csum = csum; g.stat_in = stat; g.M = M; g.N = N; g.K = K; g.flags = GV_LN | GV_OUT_F32; g.xmb = MB; g.rows = 16;
  g.y = yref;
  hipLaunchKernelGGL((gemv_frag_kernel<5, 6, false>), dim3(N / 16), dim3(256), 0, st, g.x, g.Wp, g.M, g.N, g.K, g.wks, g.wk0, 1, g);
  CK(hipStreamSynchronize(st));
  g.y = y;
  std::vector<Rec> hrec(4096);
  auto check = [&](const char* tag, const float* a, const float* ref, size_t nwords, int it, bool verbose) -> unsigned {
    CK(hipMemsetAsync(cnt, 0, 4, st));
    compare_words<<<1024, 256, 0, st>>>(reinterpret_cast<const unsigned*>(a), reinterpret_cast<const unsigned*>(ref), nwords, cnt, recs, 4096);
    unsigned c = 0; CK(hipMemcpyAsync(&c, cnt, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    if (c && verbose) {
      const unsigned nshow = c < 4096 ? c : 4096;
      CK(hipMemcpy(hrec.data(), recs, nshow * sizeof(Rec), hipMemcpyDeviceToHost));
      printf("  [%s] launch %d: %u differing words\n", tag, it, c);
      for (unsigned i = 0; i < nshow && i < 24; ++i) {
        const Rec& r = hrec[i];
        float gv, ev; memcpy(&gv, &r.got, 4); memcpy(&ev, &r.exp, 4);
        printf("    word %u  got %.6f exp %.6f diff %.6f\n", r.idx, gv, ev, gv - ev);
      }
    }
    return c;
  };
  // ---- variant f2: the library kernel, three workgroups per CU
  for (int extra : {0, 24 * 1024}) {
    unsigned bad_launches = 0, bad_words = 0;
    for (int it = 0; it < iters; ++it) {
      CK(hipMemsetAsync(y, 0xff, ny * 4, st));
      hipLaunchKernelGGL((gemv_frag2_kernel<5, 4, false>), dim3(N / 32), dim3(256), extra, st, WIS_GV_LEAD(g), g);
      const unsigned c = check(extra ? "f2+24K" : "f2", y, yref, ny, it, bad_launches < 3);
      if (c) {
        ++bad_launches; bad_words += c;
        if (bad_launches <= 3) {      // which (row, column) -> tile, feature, row block
          const unsigned nshow = c < 4096 ? c : 4096;
          for (unsigned i = 0; i < nshow && i < 24; ++i) { const unsigned row = hrec[i].idx / N, col = hrec[i].idx % N; printf("    row %u (block %u) col %u = tile %u feature %u (workgroup %u, tile %u of it)\n", row, row / 16, col, col / 16, col % 16, col / 32, (col / 16) & 1); }
        }
      }
    }
    printf("variant %s: %u of %d launches differ from the one-tile kernel, %u words\n", extra ? "f2+24K (2 workgroups/CU)" : "f2 (3 workgroups/CU)", bad_launches, iters, bad_words);
  }
  // ---- variant f2dbg: reference LDS image from a two-per-CU launch, then three-per-CU launches
  {
    CK(hipMemsetAsync(dump_ref, 0, ndump * 4, st));
    hipLaunchKernelGGL((frag2_dbg<5, 4>), dim3(N / 32), dim3(256), 24 * 1024, st, g, dump_ref, hwid);
    const unsigned c0 = check("f2dbg+24K", y, yref, ny, -1, true);
    printf("f2dbg reference launch (2 workgroups/CU): %u differing words\n", c0);
    unsigned bad_launches = 0;
    std::vector<unsigned> hhw((size_t)(N / 32) * 8);
    for (int it = 0; it < iters; ++it) {
      CK(hipMemsetAsync(y, 0xff, ny * 4, st));
      hipLaunchKernelGGL((frag2_dbg<5, 4>), dim3(N / 32), dim3(256), 0, st, g, dump, hwid);
      const unsigned c = check("f2dbg", y, yref, ny, it, bad_launches < 4);
      if (c) {
        ++bad_launches;
        if (bad_launches <= 4) {
          std::vector<Rec> out_recs(hrec.begin(), hrec.begin() + (c < 64 ? c : 64));
          CK(hipMemcpy(hhw.data(), hwid, hhw.size() * 4, hipMemcpyDeviceToHost));
          unsigned last_wg = ~0u;
          for (auto& r : out_recs) {
            const unsigned col = r.idx % N, wg = col / 32;
            if (wg == last_wg) continue;
            last_wg = wg;
            printf("    workgroup %u: hw ids (HW_ID, XCC) of its waves:", wg);
            for (int w = 0; w < 4; ++w) { const unsigned hw = hhw[((size_t)wg * 4 + w) * 2]; printf(" [w%d wave_id %u simd %u cu %u sh %u se %u xcc %u]", w, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, hhw[((size_t)wg * 4 + w) * 2 + 1] & 15); }
            printf("\n");
          }
          // the LDS image of this launch against the reference image
          const unsigned cl = check("f2dbg LDS image", dump, dump_ref, ndump, it, false);
          printf("    LDS exchange buffer: %u words differ from the reference image\n", cl);
          if (cl) {
            const unsigned nshow = cl < 4096 ? cl : 4096;
            CK(hipMemcpy(hrec.data(), recs, nshow * sizeof(Rec), hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < nshow && i < 48; ++i) {
              const unsigned per = 4 * MB * 2 * 64 * 4, wg = hrec[i].idx / per, o = hrec[i].idx % per, slot = o / 256, lane = (o % 256) / 4, comp = o % 4;
              float gv, ev; memcpy(&gv, &hrec[i].got, 4); memcpy(&ev, &hrec[i].exp, 4);
              printf("      workgroup %u: writer wave %u row block %u tile %u lane %u comp %u: LDS %.6f reference %.6f\n", wg, slot / (MB * 2), (slot / 2) % MB, slot & 1, lane, comp, gv, ev);
            }
          }
        }
      }
    }
    printf("variant f2dbg (3 workgroups/CU): %u of %d launches differ\n", bad_launches, iters);
  }
  // ---- the two-tile kernel with FOUR launches in flight (tools/frag_stress.hip found: clean alone, wrong tiles as soon as launches overlap):
  // the debug copy on four streams, each with its own output and LDS-dump buffers, against the idle-GPU reference
  {
    hipStream_t ss[4]; float* ys[4]; float* ds[4]; unsigned* hw[4];
    for (int i = 0; i < 4; ++i) { CK(hipStreamCreate(&ss[i])); CK(hipMalloc(&ys[i], ny * 4)); CK(hipMalloc(&ds[i], ndump * 4)); CK(hipMalloc(&hw[i], (size_t)(N / 32) * 4 * 2 * 4)); }
    std::vector<unsigned> hhw((size_t)(N / 32) * 8);
    int shown = 0; unsigned bad = 0;
    for (int it = 0; it < iters && shown < 6; ++it) {
      for (int i = 0; i < 4; ++i) { GemvP gi = g; gi.y = ys[i]; hipLaunchKernelGGL((frag2_dbg<5, 4>), dim3(N / 32), dim3(256), 0, ss[i], gi, ds[i], hw[i]); }
      for (int i = 0; i < 4; ++i) CK(hipStreamSynchronize(ss[i]));
      for (int i = 0; i < 4 && shown < 6; ++i) {
        const unsigned c = check("f2dbg x4 streams", ys[i], yref, ny, it, false);
        if (!c) continue;
        ++bad; ++shown;
        const unsigned nshow = c < 4096 ? c : 4096;
        CK(hipMemcpy(hrec.data(), recs, nshow * sizeof(Rec), hipMemcpyDeviceToHost));
        std::vector<Rec> out_recs(hrec.begin(), hrec.begin() + nshow);
        printf("  [x4] round %d stream %d: %u output words differ\n", it, i, c);
        CK(hipMemcpy(hhw.data(), hw[i], hhw.size() * 4, hipMemcpyDeviceToHost));
        unsigned last_wg = ~0u; int nwg = 0;
        for (auto& r : out_recs) {
          const unsigned row = r.idx / N, col = r.idx % N, wg = col / 32;
          float gv, ev; memcpy(&gv, &r.got, 4); memcpy(&ev, &r.exp, 4);
          if (wg != last_wg) {
            if (++nwg > 6) break;
            last_wg = wg;
            printf("    workgroup %u:", wg);
            for (int w = 0; w < 4; ++w) { const unsigned h = hhw[((size_t)wg * 4 + w) * 2]; printf(" [w%d slot %u simd %u cu %u sh %u se %u xcc %u]", w, h & 15, (h >> 4) & 3, (h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7, hhw[((size_t)wg * 4 + w) * 2 + 1] & 15); }
            printf("\n");
          }
          printf("      row %u (block %u, row-in-block %u) col %u (tile %u of the workgroup, feature %u): got %.5f exp %.5f diff %.5f\n", row, row / 16, row % 16, col, (col / 16) & 1, col % 16, gv, ev, gv - ev);
        }
        const unsigned cl = check("LDS image", ds[i], dump_ref, ndump, it, false);
        printf("    LDS exchange buffer of that launch: %u words differ from the reference image\n", cl);
        if (cl) {
          const unsigned ns2 = cl < 4096 ? cl : 4096;
          CK(hipMemcpy(hrec.data(), recs, ns2 * sizeof(Rec), hipMemcpyDeviceToHost));
          for (unsigned k = 0; k < ns2 && k < 40; ++k) {
            const unsigned per = 4 * MB * 2 * 64 * 4, wg = hrec[k].idx / per, o = hrec[k].idx % per, slot = o / 256, ln = (o % 256) / 4, comp = o % 4;
            float gv, ev; memcpy(&gv, &hrec[k].got, 4); memcpy(&ev, &hrec[k].exp, 4);
            printf("      workgroup %u: writer wave %u row block %u tile %u lane %u comp %u: LDS %.6f reference %.6f\n", wg, slot / (MB * 2), (slot / 2) % MB, slot & 1, ln, comp, gv, ev);
          }
        }
      }
    }
    printf("variant f2dbg, 4 streams in flight: %u differing launches shown\n", bad);
  }
  // ---- the shipped kernel under load: four streams, every launch against the idle-GPU reference
  {
    hipStream_t ss[4]; float* ys[4];
    for (int i = 0; i < 4; ++i) { CK(hipStreamCreate(&ss[i])); CK(hipMalloc(&ys[i], ny * 4)); }
    unsigned bad = 0; const int rounds = iters;
    for (int it = 0; it < rounds; ++it) {
      for (int i = 0; i < 4; ++i) { GemvP gi = g; gi.y = ys[i]; hipLaunchKernelGGL((gemv_frag_kernel<5, 6, false>), dim3(N / 16), dim3(256), 0, ss[i], gi.x, gi.Wp, gi.M, gi.N, gi.K, gi.wks, gi.wk0, 1, gi); }
      for (int i = 0; i < 4; ++i) CK(hipStreamSynchronize(ss[i]));
      for (int i = 0; i < 4; ++i) bad += check("f1 x4 streams", ys[i], yref, ny, it, bad < 3) ? 1 : 0;
    }
    printf("shipped one-tile kernel, 4 streams in flight: %u of %d launches differ from the idle-GPU launch\n", bad, 4 * rounds);
  }
  return 0;
}
