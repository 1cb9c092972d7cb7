// Stress harness of the SHIPPED batched skinny GEMM (csrc/dec_kernels.hip gemv_frag_kernel / gemv_frag3_kernel; round-3 review item 2:
// "prove the shipped kernels clean"): every instantiation the decode step can take - 1..6 row blocks x the ring depths of K = 1280 and
// K = 5120 / 2 x f16 and 8-bit weights - at Whisper large-v2's real shapes and epilogues
//     QKV (3840 x 1280, LayerNorm-folded, q / K-cache / V-cache scatter), out-projection (1280 x 1280, residual rows + fragment image
//     + LayerNorm partials), cross-Q (1280 x 1280, LayerNorm-folded, fp32), FFN1 (5120 x 1280, GELU, f16 fragment image), FFN2
//     (1280 x 5120, residual, K split over two workgroups per n-tile with the in-launch ticket merge), vocabulary (51872 x 1280),
//     and the three-problem launch of the folded cross-Q (out-projection + two halves of q_raw),
// launched back to back on FOUR streams at once (what four replicas of a GPU do: workgroups of different launches share CUs), every
// single launch compared WORD FOR WORD on the device with the result of the same launch done alone on the idle GPU.  FFN1 and the
// vocabulary take the two-n-tile kernel (the product default since round 4; WIS_FRAG_NB=1 stresses the one-tile form on them instead;
// WIS_FRAG_NB=2 set explicitly restricts the run to those two shapes - the round-4 diagnosis runs).
// Prints one line per (row blocks, shape, weight type) and a final "TOTAL ... mismatching launches"; exit code 1 on any mismatch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include -I willow-inference-server_amd/csrc \
//        -o tools/bin/frag_stress tools/frag_stress.hip          run: tools/bin/frag_stress [launches per stream = 2500]
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
#define CKW(x) do { int r_ = (x); if (r_ != WIS_OK) { printf("launch error %d (%s) at line %d\n", r_, get_error(), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash_u(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void fill_f16(f16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (f16)(((float)(hash_u((unsigned)i * 2654435761u + seed) & 0xffff) / 32768.0f - 1.0f) * scale);
}
__global__ void fill_u8(unsigned char* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)(hash_u((unsigned)i * 2654435761u + seed) >> 13);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float offset) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = ((float)(hash_u((unsigned)i * 2654435761u + seed) & 0xffff) / 32768.0f - 1.0f) * scale + offset;
}
// one slot per launch: number of words that differ from the reference arena
__global__ void compare_words(const unsigned* a, const unsigned* ref, size_t n, unsigned* slot) {
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != ref[i];
  if (bad) atomicAdd(slot, bad);
}

// per-stream output arena (everything a launch may write), identical layout for every stream and the reference
struct Arena { char* base; size_t bytes; float* y; f16* yxf; float* stat; float* q; float* q2; f16* kc; f16* vc; float* kpart; unsigned* kcnt; };
static Arena make_arena() {
  const size_t ny = (size_t)MAX_ROWS * 51872 * 4, nxf = (size_t)160 * 6 * 1024, nst = (size_t)MAX_ROWS * 80 * 8, nq = (size_t)MAX_ROWS * 1280 * 4, nkv = (size_t)MAX_ROWS * 4 * 1280 * 2,
               nkp = (size_t)80 * 2 * 6 * 1024, nkc = 80 * 4;
  Arena a; a.bytes = ny + nxf + nst + 2 * nq + 2 * nkv + nkp + nkc;
  CK(hipMalloc(&a.base, a.bytes)); CK(hipMemset(a.base, 0, a.bytes));
  char* p = a.base;
  a.y = (float*)p; p += ny; a.yxf = (f16*)p; p += nxf; a.stat = (float*)p; p += nst; a.q = (float*)p; p += nq; a.q2 = (float*)p; p += nq;
  a.kc = (f16*)p; p += nkv; a.vc = (f16*)p; p += nkv; a.kpart = (float*)p; p += nkp; a.kcnt = (unsigned*)p;
  return a;
}

int main(int argc, char** argv) {
  const int per_stream = argc > 1 ? atoi(argv[1]) : 2500;
  const bool nb2 = getenv("WIS_FRAG_NB") && atoi(getenv("WIS_FRAG_NB")) == 2;
  const int d = 1280, NS = 4;
  hipStream_t ss[NS]; for (int i = 0; i < NS; ++i) CK(hipStreamCreate(&ss[i]));
  // shared, read-only operands: random packed weights (any bytes are a valid fragment image), activation images, partials, vectors
  f16 *Wp, *xf; unsigned char* W8; float *stat_in, *csum, *bias, *wscale, *y0; int *slot, *pos;
  CK(hipMalloc(&Wp, (size_t)51872 * 1280 * 2)); CK(hipMalloc(&W8, (size_t)51872 * 1280)); CK(hipMalloc(&xf, (size_t)160 * 6 * 1024)); CK(hipMalloc(&stat_in, (size_t)MAX_ROWS * 320 * 8));
  CK(hipMalloc(&csum, 51872 * 4)); CK(hipMalloc(&bias, 51872 * 4)); CK(hipMalloc(&wscale, 51872 * 4)); CK(hipMalloc(&y0, (size_t)MAX_ROWS * d * 4));
  CK(hipMalloc(&slot, MAX_ROWS * 4)); CK(hipMalloc(&pos, MAX_ROWS * 4));
  fill_f16<<<2048, 256>>>(Wp, (size_t)51872 * 1280, 11u, 0.05f); fill_u8<<<2048, 256>>>(W8, (size_t)51872 * 1280, 13u);
  fill_f16<<<256, 256>>>(xf, (size_t)160 * 6 * 512, 23u, 2.0f);
  fill_f32<<<256, 256>>>(stat_in, (size_t)MAX_ROWS * 320 * 2, 37u, 4.0f, 8.0f);
  fill_f32<<<256, 256>>>(csum, 51872, 41u, 0.3f, 0.f); fill_f32<<<256, 256>>>(bias, 51872, 43u, 1.0f, 0.f); fill_f32<<<256, 256>>>(wscale, 51872, 47u, 0.0002f, 0.0005f);
  fill_f32<<<256, 256>>>(y0, (size_t)MAX_ROWS * d, 53u, 1.0f, 0.2f);
  { std::vector<int> hs(MAX_ROWS), hp(MAX_ROWS); for (int m = 0; m < MAX_ROWS; ++m) { hs[m] = m; hp[m] = m & 3; } CK(hipMemcpy(slot, hs.data(), MAX_ROWS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pos, hp.data(), MAX_ROWS * 4, hipMemcpyHostToDevice)); }
  CK(hipDeviceSynchronize());
  Arena ref = make_arena(), ar[NS]; for (int i = 0; i < NS; ++i) ar[i] = make_arena();
  unsigned* slots; const size_t nslots = (size_t)NS * per_stream; CK(hipMalloc(&slots, nslots * 4));
  std::vector<unsigned> hslots(nslots);

  struct Shape { const char* name; int N, K, flags, ksplit, kind; };      // kind 0 = one problem, 1 = the three-problem fold launch
  const Shape shapes[] = {{"QKV 3840x1280 LN+scatter", 3 * d, d, GV_LN | GV_QKV, 1, 0}, {"out 1280x1280 resid", d, d, GV_RESID, 1, 0}, {"cross-Q 1280x1280 LN f32", d, d, GV_LN | GV_OUT_F32, 1, 0},
                          {"FFN1 5120x1280 LN+GELU image", 4 * d, d, GV_LN | GV_GELU, 1, 0}, {"FFN2 1280x5120 resid ksplit2|msplit", d, 4 * d, GV_RESID, 2, 0},
                          {"vocab 51872x1280 LN f32", 51872, d, GV_LN | GV_OUT_F32, 1, 0}, {"fold3 out+qA+qB 1280x1280", d, d, 0, 1, 1}};
  auto fill_g = [&](const Shape& s, const Arena& a, int M, bool w8, GemvP* g3) {
    const int MB = (M + 15) / 16;
    GemvP g; memset(&g, 0, sizeof(g));
    g.x = xf; g.Wp = w8 ? reinterpret_cast<const f16*>(W8) : Wp; g.wscale = w8 ? wscale : nullptr; g.bias = bias; g.M = M; g.N = s.N; g.K = s.K; g.flags = s.flags; g.xmb = MB; g.rows = 16;
    if (s.flags & GV_LN) { g.csum = csum; g.stat_in = stat_in; }
    if (s.flags & GV_QKV) { g.q = a.q; g.kc = a.kc; g.vc = a.vc; g.slot = slot; g.pos = pos; g.d = d; g.ctx = 4; }
    if (s.flags & GV_RESID) { g.y = a.y; g.y_xf = a.yxf; g.ymb = MB; g.stat_out = a.stat; }
    else if (s.flags & GV_GELU) { g.y = a.yxf; g.ymb = MB; }
    else g.y = a.y;
    if (s.ksplit > 1) { g.ksplit = s.ksplit; g.kpart = a.kpart; g.kcnt = a.kcnt; }
    if (s.kind == 1) {
      g3[0] = g; g3[0].flags = GV_RESID; g3[0].y = a.y; g3[0].y_xf = nullptr; g3[0].ymb = MB; g3[0].stat_out = a.stat;
      g3[1] = g; g3[1].flags = GV_OUT_F32; g3[1].y = a.q; g3[1].wks = 2 * d / 32; g3[1].wk0 = 0;
      g3[2] = g; g3[2].flags = GV_OUT_F32; g3[2].y = a.q2; g3[2].bias = nullptr; g3[2].wks = 2 * d / 32; g3[2].wk0 = d / 32;
    } else g3[0] = g;
  };
  auto launch = [&](const Shape& s, const Arena& a, int M, bool w8, hipStream_t st) {
    GemvP g3[3]; fill_g(s, a, M, w8, g3);
    if ((s.flags & GV_RESID) || s.kind == 1) CK(hipMemcpyAsync(a.y, y0, (size_t)M * d * 4, hipMemcpyDeviceToDevice, st));      // in-place residual rows start from the same values
    if (s.kind == 1) CKW(launch_gemv_frag3(st, g3, 3)); else CKW(launch_gemv_frag(st, g3[0]));
  };
  unsigned long long total_launches = 0, total_bad = 0;
  for (int w8 = 0; w8 < 2; ++w8) {
    for (int MB = 1; MB <= 6; ++MB) {
      const int M = 16 * MB - 3;
      for (const Shape& s : shapes) {
        if (w8 && s.kind == 1) continue;                           // the fold launch exists for f16 weights only (model.hip cq_fold)
        if (nb2 && !(((s.flags & GV_LN) && !(s.flags & (GV_RESID | GV_QKV)) && s.N / 16 > 256))) continue;      // WIS_FRAG_NB=2: only the shapes the two-tile kernel takes
        // reference: alone on the idle GPU
        CK(hipDeviceSynchronize());
        CK(hipMemsetAsync(ref.base, 0, ref.bytes, ss[0]));
        launch(s, ref, M, w8, ss[0]);
        CK(hipStreamSynchronize(ss[0]));
        for (int i = 0; i < NS; ++i) CK(hipMemsetAsync(ar[i].base, 0, ar[i].bytes, ss[i]));
        CK(hipMemsetAsync(slots, 0, nslots * 4, ss[0]));
        CK(hipDeviceSynchronize());
        // the words a launch of this shape can touch: compare the output regions it uses (y rows up to N columns, images, q, caches, slices)
        const size_t cmp_bytes = (s.N == 51872 ? (size_t)M * 51872 * 4 : ref.bytes - (size_t)MAX_ROWS * 51872 * 4 + (size_t)M * 5120 * 4);
        for (int it = 0; it < per_stream; ++it)
          for (int i = 0; i < NS; ++i) {
            launch(s, ar[i], M, w8, ss[i]);
            if (s.N == 51872) compare_words<<<512, 256, 0, ss[i]>>>((const unsigned*)ar[i].y, (const unsigned*)ref.y, cmp_bytes / 4, slots + (size_t)i * per_stream + it);
            else {
              compare_words<<<64, 256, 0, ss[i]>>>((const unsigned*)ar[i].y, (const unsigned*)ref.y, (size_t)M * 5120, slots + (size_t)i * per_stream + it);
              compare_words<<<64, 256, 0, ss[i]>>>((const unsigned*)ar[i].yxf, (const unsigned*)ref.yxf, (ref.bytes - (size_t)MAX_ROWS * 51872 * 4) / 4, slots + (size_t)i * per_stream + it);
            }
          }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hslots.data(), slots, nslots * 4, hipMemcpyDeviceToHost));
        unsigned bad_launches = 0; unsigned long long bad_words = 0;
        for (unsigned v : hslots) { bad_launches += v != 0; bad_words += v; }
        total_launches += nslots; total_bad += bad_launches;
        printf("%s weights, %d row blocks (%2d rows), %-30s: %u of %zu launches differ from the idle-GPU launch (%llu words)\n", w8 ? "8-bit" : "f16 ", MB, M, s.name, bad_launches, nslots, bad_words);
        fflush(stdout);
      }
    }
  }
  printf("TOTAL: %llu mismatching launches of %llu (4 streams in flight%s)\n", total_bad, total_launches, nb2 ? ", WIS_FRAG_NB=2" : "");
  return total_bad ? 1 : 0;
}
