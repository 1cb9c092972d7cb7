// Lab for the encoder attention loop (csrc/enc_kernels.hip enc_attn_lazy_kernel), round 4 -> 5: can the kernel run FOUR waves per SIMD?
//
// The shipped loop holds 158 VGPRs (three waves per SIMD).  Its counters (profiles/r04_pmc_encoder_sq.md) say no unit is busier than
// 0.6 and a wave spends 70 % of its time waiting on its own dependent chain (LDS read -> 4-deep MFMA chain -> 32 serial quarter-rate
// exponentials -> 4-deep MFMA chain -> barrier): one more wave per SIMD is the lever, and it needs <= 128 registers.  Where they go:
// O 32, scores 32, Q fragments 16, reference splat 16, P pairs 16 (overlapping the scores), the K / V tile on its way to LDS 16.
//   attn4<splat, W>  the K / V tiles go to LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into an UNPADDED
//                    image, 16-byte chunk c of row r in slot c ^ ((r >> 1) & 7) (conflict-free for the 32x32x16 operand reads: checked
//                    by enumeration of the ds_read_b128 lane groups) - 32 KiB of LDS per workgroup instead of 36.9; the scores are
//                    taken half tile by half tile (16 score registers at a time), and a reference step simply re-runs the fast path
//                    against the raised reference; the tile that holds keys >= T runs apart with the mask compiled in;
//   attn4<v_sub, W>  the same without the reference splat: the scores leave the MFMA raw and the reference is subtracted on the vector
//                    ALU (32 v_sub per tile back, 16 registers fewer).
// W = waves per SIMD the build is held to (amdgpu_waves_per_eu(W, W)); the run checks each variant against the shipped kernel (one f16
// ulp on a few outputs is the expected difference: the reference step is arranged differently) and times all of them.
//
// MEASURED (MI355X, 20 launches, uniform random operands - more reference steps than the encoder's data; gpurun_out/r4r, r4s):
//   3 utterances (720 workgroups): shipped loop 53.4 us (647 TF; round 2's loop 61.3); attn4<v_sub, 3> (167 VGPRs, no scratch) 58.3 us;
//   8 utterances (1920 workgroups): shipped 142.7 us (646 TF; round 2's 173.3); attn4<v_sub, 3> 139.8 us
// - LDS-DMA and the unpadded image buy nothing to speak of at equal occupancy; attn4<splat, 3> (168 VGPRs + 116 B of scratch) 83 / 212 us;
// held to four waves per SIMD hipcc spills 220 B (v_sub) / 564 B (splat) and the kernels take 97.5 / 218 us and 431 / 1128 us: by count the live set is ~124 registers (O 32 + Q 16 + two half tiles of scores 32 + K fragments 16 + P pairs 8 + ~20 of
// addresses and bookkeeping), the allocator spills the DMA addresses and the Q fragments, and ONE scratch reload behind the DMA issue
// makes its `s_waitcnt vmcnt(0)` wait for the four DMA pieces as well (one in-order counter) - the prefetch overlap is gone.  Four waves
// per SIMD therefore needs either the Q fragments out of registers (they do not fit in LDS beside four workgroups' K / V buffers:
// 4 x (32 + 16) KiB > 160) or a hand-allocated loop; neither is a round-4 change.  (An earlier B = 8 run of this lab faulted: the
// uniform-pointer macro sign-extended the low address half; fixed below.)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include -I willow-inference-server_amd/csrc \
//        -o tools/bin/attn_lab tools/attn_lab.hip ;  run: tools/bin/attn_lab [B=8] [iters=20]
#include "../willow-inference-server_amd/csrc/enc_kernels.hip"
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
__global__ void compare_f16(const f16* a, const f16* ref, size_t n, unsigned* ndiff, float* maxabs) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = (float)a[i], y = (float)ref[i];
    if (!(x == y)) { atomicAdd(ndiff, 1u); const float dlt = fabsf(x - y); atomicMax(reinterpret_cast<unsigned*>(maxabs), __float_as_uint(dlt == dlt ? dlt : 1e30f)); }
  }
}

#define LAB_DMA(src, dst) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src), (lds_ptr_t)(dst), 16, 0, 0)

// grid (ceil(T/128), H, B), block 256: the unsplit form of enc_attn_lazy_kernel (see there for the algorithm)
template <bool NEGM, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void attn4(const f16* __restrict__ qk, const f16* __restrict__ vt, f16* __restrict__ out,
                                                                                        int T, int Tpad, int H, int d) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(1024))) f16 sK0[64 * 64], sK1[64 * 64], sV0[64 * 64], sV1[64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z, qt = blockIdx.x;
  const int q_row = qt * 128 + wave * 32 + l31;
  const int q_c = q_row < T ? q_row : T - 1;
  const int ld = 2 * d;
  f16x8 qf[4];
  {
    const f16* qp = qk + (size_t)(b * T + q_c) * ld + h * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const f16x8*>(qp + kk * 16 + hi * 8);
  }
  // DMA pieces of this thread: rows wave * 16 + 8 j + (lane >> 3) of the K tile (keys) and of the V^T tile (head dims), slot lane & 7.
  // Sources as (wave-uniform base) + (32-bit lane offset): the saddr form of global_load_lds, one VGPR per piece instead of a pointer pair.
  const int drow = wave * 16 + (lane >> 3), dslot = lane & 7;
  const int dc0 = (dslot ^ ((drow >> 1) & 7)) * 8, dc1 = (dslot ^ (((drow + 8) >> 1) & 7)) * 8;      // source chunk (f16 offset) for j = 0, 1
  const char* kbase = reinterpret_cast<const char*>(qk + (size_t)b * T * ld + d + h * 64);
  const char* vbase = reinterpret_cast<const char*>(vt + (size_t)(b * H + h) * 64 * Tpad);
  const unsigned vo0 = (unsigned)(drow * Tpad + dc0) * 2u, vo1 = (unsigned)((drow + 8) * Tpad + dc1) * 2u;
  // (the wave-uniform part goes through readfirstlane: hipcc otherwise re-associates base + lane offset into a loop-invariant 64-bit
  // VGPR pair per piece and the DMA takes the two-register address form)
#define LAB_UNIFORM(ptr) reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)(ptr) >> 32)) << 32) | \
                                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(ptr)))      /* (readfirstlane returns int: without the unsigned cast the low half sign-extends) */
#define LAB_LOAD(kt, SK, SV)                                                               \
  {                                                                                        \
    const int kr = T - 1 - (kt) * AKT;      /* last valid row of this tile (>= 63 except in the last tile) */ \
    const int r0 = drow < kr ? drow : kr, r1 = drow + 8 < kr ? drow + 8 : kr;               \
    const char* kb_ = LAB_UNIFORM(kbase + (size_t)(kt) * AKT * ld * 2);                    \
    LAB_DMA(kb_ + (unsigned)(r0 * ld + dc0) * 2u, SK + wave * 1024);                       \
    LAB_DMA(kb_ + (unsigned)(r1 * ld + dc1) * 2u, SK + wave * 1024 + 512);                 \
    const char* vb_ = LAB_UNIFORM(vbase + (size_t)(kt) * AKT * 2);                         \
    LAB_DMA(vb_ + vo0, SV + wave * 1024);                                                  \
    LAB_DMA(vb_ + vo1, SV + wave * 1024 + 512);                                            \
  }
  // operand reads: this lane's rows are t2 * 32 + l31 (K) / dt * 32 + l31 (V^T): row offset + swizzled chunk; the four chunk offsets of a
  // lane (k-steps 0..3) are the same for K and V^T
  const int rsw = (l31 >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = l31 * 64 + (((kk * 2 + hi) ^ rsw) << 3);
  // scores of one 32-key half tile, relative to the current reference (keys >= T, last tile only: -inf)
#define LAB_SCORES(SK, t2, stt, MASK)                                                            \
  {                                                                                        \
    f16x8 kf[4];                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const f16x8*>(&SK[(t2) * 32 * 64 + fo[kk]]); \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                       \
      stt = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk], qf[kk], kk == 0 ? (NEGM ? negm : zero16) : stt, 0, 0, 0); \
    if (!NEGM) { _Pragma("unroll") for (int r = 0; r < 16; ++r) stt[r] -= m_run; }         \
    if (MASK) {                                                                            \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                       \
        if (kt * AKT + 4 * hi + (t2) * 32 + (r & 3) + 8 * (r >> 2) >= T) stt[r] = -INFINITY; \
    }                                                                                      \
  }
  // the tile's weights, half tile by half tile (16 score registers live at a time; the exponentials of one half run under the MFMAs of
  // the next): f16 pairs in P.V operand order and the row sum of the rounded weights
#define LAB_FAST(SK, MASK)                                                                     \
  rs = 0.f;                                                                                \
  _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                       \
    f32x16 stt;                                                                            \
    LAB_SCORES(SK, t2, stt, MASK)                                                          \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                       \
      _Pragma("unroll") for (int j = 0; j < 8; j += 2) {                                   \
        const f16x2 pr = {(f16)__builtin_amdgcn_exp2f(stt[8 * s2 + j]), (f16)__builtin_amdgcn_exp2f(stt[8 * s2 + j + 1])}; \
        pf[2 * t2 + s2][j] = pr[0]; pf[2 * t2 + s2][j + 1] = pr[1];                        \
        rs = __builtin_amdgcn_fdot2(pr, f16x2{(f16)1.f, (f16)1.f}, rs, false);             \
      }                                                                                    \
    __builtin_amdgcn_sched_barrier(0);      /* the halves do NOT overlap: 16 score registers, not 32 */ \
  }
  // one key tile: the fast path; when its row sums leave the f16 range (or nothing is known yet: first tile) the reference is raised to
  // the tile's maximum and the fast path is simply run again, now relative to the new reference
#define LAB_TILE(KT, SK, SV, FIRST, MASK)                                                        \
  {                                                                                        \
    const int kt = (KT); const bool first = (FIRST);                                       \
    f16x8 pf[4];                                                                           \
    float rs = 0.f;                                                                        \
    bool slow = true;                                                                      \
    if (!first) { LAB_FAST(SK, MASK) slow = __any(!(rs < 32768.f)); }                            \
    if (slow) {                                                                            \
      float mx = -INFINITY;                                                                \
      _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2) {                                   \
        f32x16 stt;                                                                        \
        LAB_SCORES(SK, t2, stt, MASK)                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) mx = fmaxf(mx, stt[r]);             \
      }                                                                                    \
      mx = fmaxf(mx, __shfl_xor(mx, 32));                                                  \
      const float delta = first ? mx : fmaxf(mx, 0.f);                                     \
      const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);                    \
      m_run += delta;                                                                      \
      l_run *= alpha;                                                                      \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o[0][r] = mul_scalar(o[0][r], alpha); o[1][r] = mul_scalar(o[1][r], alpha); if (NEGM) negm[r] = -m_run; } \
      LAB_FAST(SK, MASK)                                                                   \
    }                                                                                      \
    l_run += rs;                                                                           \
    _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                     \
      f16x8 vf[4];                                                                         \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) vf[s] = *reinterpret_cast<const f16x8*>(&SV[dt * 32 * 64 + fo[s]]); \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s], pf[s], o[dt], 0, 0, 0); \
    }                                                                                      \
  }

  f32x16 o[2], negm;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
  float m_run = 0.f, l_run = 0.f;
  const int nt_all = cdiv(T, AKT), nt_full = T / AKT;      // the tile that holds keys >= T (if any) runs apart, with the mask compiled in
  LAB_LOAD(0, sK0, sV0)
  __syncthreads();      // (its fence drains the DMA queue: vmcnt(0) in front of the barrier)
  for (int kb = 0; kb < nt_full; kb += 2) {
    if (kb + 1 < nt_all) LAB_LOAD(kb + 1, sK1, sV1)
    LAB_TILE(kb, sK0, sV0, kb == 0, false)
    __syncthreads();
    if (kb + 1 >= nt_full) break;
    if (kb + 2 < nt_all) LAB_LOAD(kb + 2, sK0, sV0)
    LAB_TILE(kb + 1, sK1, sV1, false, false)
    __syncthreads();
  }
  if (nt_full < nt_all) {
    const f16* sKt = (nt_full & 1) ? sK1 : sK0;
    const f16* sVt = (nt_full & 1) ? sV1 : sV0;
    LAB_TILE(nt_full, sKt, sVt, nt_full == 0, true)
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

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, iters = argc > 2 ? atoi(argv[2]) : 20;
  const int T = 1500, H = 20, d = H * 64, Tpad = 1536;
  f16 *qk, *vt, *o_ref, *o_a, *o_b; unsigned* ndiff; float* maxabs;
  const size_t nqk = (size_t)B * T * 2 * d, nvt = (size_t)B * H * 64 * Tpad, no = (size_t)B * T * d;
  CK(hipMalloc(&qk, nqk * 2)); CK(hipMalloc(&vt, nvt * 2)); CK(hipMalloc(&o_ref, no * 2)); CK(hipMalloc(&o_a, no * 2)); CK(hipMalloc(&o_b, no * 2));
  CK(hipMalloc(&ndiff, 4)); CK(hipMalloc(&maxabs, 4));
  hipLaunchKernelGGL(fill_f16, dim3(2048), dim3(256), 0, 0, qk, nqk, 1u, 1.6f);      // Q (log2 domain, pre-scaled) | K: scores ~ N(0, 4.3^2)
  hipLaunchKernelGGL(fill_f16, dim3(2048), dim3(256), 0, 0, vt, nvt, 2u, 1.0f);
  CK(hipDeviceSynchronize());
  const dim3 grid(cdiv(T, 128), H, B), blk(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, f16* outp, bool exact) {
    CK(hipMemset(outp, 0, no * 2));
    launch(outp); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(outp);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned nd = 0; float ma = 0.f;
    if (outp != o_ref) {
      CK(hipMemset(ndiff, 0, 4)); CK(hipMemset(maxabs, 0, 4));
      hipLaunchKernelGGL(compare_f16, dim3(2048), dim3(256), 0, 0, outp, o_ref, no, ndiff, maxabs);
      CK(hipMemcpy(&nd, ndiff, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ma, maxabs, 4, hipMemcpyDeviceToHost));
    }
    const double flop = 4.0 * T * (double)T * 64 * H * B;
    printf("%-34s %8.2f us per launch  %7.1f TF   vs shipped: %u of %zu outputs differ, max abs %.3e%s\n", name, ms * 1000.f / iters, flop / (ms / iters * 1e-3) / 1e12, nd, no, ma,
           outp == o_ref ? "" : (exact ? (nd ? "   <-- MUST BE IDENTICAL" : "   (identical)") : ""));
  };
  printf("encoder attention lab: B=%d T=%d H=%d, %d launches each\n", B, T, H, iters);
  timeit("shipped enc_attn_lazy_kernel<false>", [&](f16* o) { hipLaunchKernelGGL((enc_attn_lazy_kernel<false>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d, (float*)nullptr, (unsigned*)nullptr); }, o_ref, true);
  timeit("round-2 enc_attn_kernel<false>", [&](f16* o) { hipLaunchKernelGGL((enc_attn_kernel<false>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d, (float*)nullptr, (unsigned*)nullptr); }, o_a, false);
  timeit("attn4<splat, 3 waves/SIMD> (DMA)", [&](f16* o) { hipLaunchKernelGGL((attn4<true, 3>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d); }, o_a, false);
  timeit("attn4<v_sub, 3 waves/SIMD> (DMA)", [&](f16* o) { hipLaunchKernelGGL((attn4<false, 3>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d); }, o_b, false);
  timeit("attn4<splat, 4 waves/SIMD> (DMA)", [&](f16* o) { hipLaunchKernelGGL((attn4<true, 4>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d); }, o_a, false);
  timeit("attn4<v_sub, 4 waves/SIMD> (DMA)", [&](f16* o) { hipLaunchKernelGGL((attn4<false, 4>), grid, blk, 0, 0, qk, vt, o, T, Tpad, H, d); }, o_b, false);
  return 0;
}
