// model.hip — model lifecycle, weight re-packing, the generate / detect_language drivers and
// the C-ABI entry points of libwis_hip.so (include/wis_hip.h).
//
// Replaces the ctranslate2.models.Whisper object the reference builds at main.py:341-444 and
// drives at main.py:535-537, 638-639, 685-693.  One handle = one replica on one GPU: weights are
// converted ONCE into the layouts the kernels stream (row-major f16 [N][K] for the encoder MFMA
// GEMM, MFMA-fragment-packed for the decoder's skinny GEMMs), all activations / KV caches live in
// HBM for the handle's lifetime, and a decode step (~260 kernels) is captured once into a HIP
// graph and replayed: every step-dependent value (positions, tokens, beam parents, scores)
// lives in device memory, so the host only launches graphs and polls a done counter.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

using namespace wis;

namespace wis {
void build_mel_filters(float* out);

// ---- small utility kernels -------------------------------------------------------------
// generic convert: src (f16|f32) [rows][cols] -> dst (f16|f32) [rows][dst_ld]; rows < n_scale scaled
__global__ void convert_kernel(const void* __restrict__ src, int src_f16, void* __restrict__ dst, int dst_f16,
                               int64_t rows, int64_t cols, int64_t dst_ld, int64_t n_scale, float scale) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    float v = src_f16 ? (float)reinterpret_cast<const f16*>(src)[i] : reinterpret_cast<const float*>(src)[i];
    if (r < n_scale) v *= scale;
    if (dst_f16) reinterpret_cast<f16*>(dst)[r * dst_ld + c] = (f16)v; else reinterpret_cast<float*>(dst)[r * dst_ld + c] = v;
  }
}
// conv weight [out][in][3] -> f16 [out][row_len], element (o, k*cpad + c) = W[o][c][k]; channel padding and the row tail
// [3*cpad, row_len) are zero (row_len = K rounded up to the GEMM's 64-deep k-tile)
__global__ void conv_pack_kernel(const void* __restrict__ src, int src_f16, f16* __restrict__ dst, int out, int in, int cpad, int row_len) {
  const int64_t total = (int64_t)out * row_len;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i / row_len), rem = (int)(i - (int64_t)o * row_len), k = rem / cpad, c = rem - k * cpad;
    float v = 0.f;
    if (k < 3 && c < in) {
      const int64_t s = ((int64_t)o * in + c) * 3 + k;
      v = src_f16 ? (float)reinterpret_cast<const f16*>(src)[s] : reinterpret_cast<const float*>(src)[s];
    }
    dst[i] = (f16)v;
  }
}
// mel f32 [B][80][3000] -> conv1 input image f16 [B][3002][96]
__global__ void mel_to_image_kernel(const float* __restrict__ mel, f16* __restrict__ img) {
  __shared__ float s_t[80][65];
  const int tid = threadIdx.x, w = blockIdx.y, f0 = blockIdx.x * 64;
  for (int o = tid; o < 80 * 64; o += 256) {
    const int m = o >> 6, f = o & 63;
    if (f0 + f < 3000) s_t[m][f] = mel[((size_t)w * 80 + m) * 3000 + f0 + f];
  }
  __syncthreads();
  for (int o = tid; o < 64 * 96; o += 256) {
    const int f = o / 96, c = o - f * 96;
    if (f0 + f < 3000) img[((size_t)w * 3002 + f0 + f + 1) * 96 + c] = (c < 80) ? (f16)s_t[c][f] : (f16)0.f;
  }
}
// dst[c][r] = src[r][c] as f16 (src f16 or f32, [rows][cols])
__global__ void transpose_to_f16_kernel(const void* __restrict__ src, int src_f16, f16* __restrict__ dst, int rows, int cols) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) t[i][tx] = src_f16 ? (float)reinterpret_cast<const f16*>(src)[(size_t)r * cols + c] : reinterpret_cast<const float*>(src)[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = (f16)t[tx][i];
  }
}
// y[n] = sum_k W[n][k] x[k]  (W f16 rows of pitch ld, x fp32); one wave per row
__global__ void matvec_rows_kernel(const f16* __restrict__ W, int ld, const float* __restrict__ x, float* __restrict__ y, int N, int K) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n >= N) return;
  float a = 0.f;
  for (int k = lane; k < K; k += 64) a += (float)W[(size_t)n * ld + k] * x[k];
  a = wave_sum(a);
  if (lane == 0) y[n] = a;
}
__global__ void f16_to_f32_kernel(const f16* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
}  // namespace wis

// ---------------------------------------------------------------------------------------
struct EncLayerW {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  f16 *w_qkv, *w_out, *w_f1, *w_f2;
  float *b_qkv, *b_out, *b_f1, *b_f2;
};
struct DecLayerW {
  float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  f16 *p_qkv, *p_out, *p_cq, *p_cout, *p_f1, *p_f2;   // MFMA-fragment packed
  f16 *w_ckv;                                        // row-major [2d][d] (encoder-side GEMM)
  float *b_qkv, *b_out, *b_cq, *b_ckv, *b_cout, *b_f1, *b_f2;
  float *s_qkv = nullptr, *s_out = nullptr, *s_cq = nullptr, *s_cout = nullptr, *s_f1 = nullptr, *s_f2 = nullptr;   // int8_float16: row scales
  float *c_qkv = nullptr, *c_cq = nullptr, *c_f1 = nullptr;   // column sums of the LayerNorm-folded weights (dec_kernels.hip fold_ln_kernel)
  f16* p_cqo = nullptr; float* b_cqo = nullptr;               // cross-Q folded THROUGH the out-projection: packed [W'q | W'q Wo] ([d][2d]) and W'q bo (fused_out_cq)
};

struct GraphKey {
  int B, beam, P, max_new, fixed_new, suppress_blank, suppress_default, early_exit, spin, sa_nb; float lp, patience;
  bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) < 0; }
};

// the device slabs that hold a model's converted weights: shared by every replica cloned from it on the same GPU (wis_model_clone),
// freed when the last of them is destroyed
struct WeightSlabs {
  int device; std::vector<void*> slabs; size_t bytes = 0;
  ~WeightSlabs() { hipSetDevice(device); for (void* p : slabs) hipFree(p); }
};

struct wis_model {
  wis_config_t cfg;
  int device;
  DeviceCtx* ctx;
  hipStream_t st;
  std::shared_ptr<WeightSlabs> wslabs;      // weights (read-only after load): possibly shared with clones
  std::vector<void*> allocs;    // this replica's own slabs (activations, caches, state), carved by dalloc()
  char* slab_cur = nullptr; size_t slab_left = 0;
  size_t bytes = 0;
  // weights
  f16 *w_conv1, *w_conv2; float *b_conv1, *b_conv2, *enc_pos, *enc_ln_g, *enc_ln_b;
  std::vector<EncLayerW> enc;
  std::vector<DecLayerW> dec;
  f16 *emb, *dec_pos, *p_proj; float *dec_ln_g, *dec_ln_b;
  float *bias_all, *bias_begin; int* d_lang_ids;
  int n_vocab_pad;
  // activations
  int Tpad;
  f16 *img, *c1, *xn, *qk, *vt, *ao, *hbuf, *mem;
  float* x; float* skbuf; float* enc_part = nullptr; unsigned* enc_cnt = nullptr;
  std::vector<f16*> kx, vx;             // per decoder layer cross K / V
  std::vector<f16*> kc, vc;             // per decoder layer self KV cache [slots][ctx][d] (views into kc_all / vc_all)
  f16 *kc_all = nullptr, *vc_all = nullptr; size_t kv_layer_stride = 0;
  f16 *kx_all = nullptr, *vx_all = nullptr, *w_ckv_all = nullptr; float* b_ckv_all = nullptr;   // cross K/V: one block each, all layers
  size_t kx_lstride = 0, vx_lstride = 0;
  // decode state
  float *dx, *dq, *logits, *part; f16 *dao, *dh, *dln; unsigned* counters;
  float* dq2 = nullptr;         // batched fold: second half of the cross-attention q_raw (dec_forward_frag)
  int sa_nb = 8;      // 8-position blocks the step's self-attention asks for per pass (generate_impl: by the step index; 8 outside the decode loop)
  float* gf_part = nullptr; unsigned* gf_cnt = nullptr; int gf_ksplit = 1;      // K split of the batched FFN2 skinny GEMM: slice sums, tickets (GemvP::ksplit)
  unsigned long long* ca_gran = nullptr; unsigned* ca_epoch = nullptr;      // granule hand-off of the decoder cross-attention (small grids): slots, flag + epochs
  unsigned long long* sa_gran = nullptr; unsigned* sa_epoch = nullptr;      // ... of q / k / v from the QKV projection to the self-attention fused into its launch: slots [H][8][3][64], epochs [H]
  bool spin_off = false;        // sticky: a combiner's bounded spin ran out once on this handle - it keeps to the ticket hand-off from then on
  int handoff_retries = 0;      // calls repeated because a combiner's spin ran out (wis_debug_handoff: expected to stay 0)
  bool spin_now = true;         // this call's decision (SpinClaim): dec_forward passes the granule buffers only when set
  f16 *dxf = nullptr, *daoxf = nullptr, *dhxf = nullptr; float* dstat = nullptr;   // batched rows: fragment images of x / attention out / FFN hidden, row partial sums
  RowMeta rm; BeamState bs;
  RowMeta rm_win;               // row metadata of a draft-verification window (wis_generate_draft_beam): the search's own rows (rm, written by beam_step_kernel) stay untouched
  float *st_max, *st_sum, *st_val; int* st_idx;
  float* d_in; int64_t* d_nsamp; float* d_probs;
  int* vstep = nullptr; int* pick_tok = nullptr; float* pick_lp = nullptr;      // wis_generate_draft: per-row step index, picked token / log-probability of the teacher-forced rows
  int* d_draft = nullptr; int* d_anc = nullptr; int* d_vstate = nullptr; int* d_base = nullptr;        // wis_generate_draft_beam: the draft trajectory [256][MAX_R][2], the window rows' ancestor slots [MAX_ROWS][32], {steps verified}
  float* lm_logspec = nullptr; unsigned* lm_gmax = nullptr;   // log-mel scratch of THIS replica (never shared with other callers)
  int* h_pin;      // pinned host scratch
  unsigned long long* h_prog = nullptr;      // host-mapped progress block of the beam search (kernels.hpp HP_*): the decode loop polls it
  unsigned gen = 0;                          // generation of the current search (records of an earlier call's over-run step are ignored)
  int last_B = 0, last_beam = 0;             // shape of the last generate call (wis_last_trajectory)
  bool overrun_left = false;                 // the last call returned with an over-run decode step still queued (its give-up flag, if any, is not the next call's)
  hipEvent_t ev[8];
  hipStream_t st_enc = nullptr;             // wis_generate's front half (log-mel, encoder) runs here: it overlaps the previous call's over-run decode step on `st`
  hipEvent_t ev_enc = nullptr, ev_ckv = nullptr;      // encoder output ready (st_enc -> st); cross-K/V projection done reading it (st -> the next call's st_enc)
  wis_timing_t timing;
  std::map<GraphKey, hipGraphExec_t> graphs;
  bool use_graph;
  float* s_proj = nullptr;      // int8_float16: row scales of the vocabulary projection
  float* c_proj = nullptr; float* b_proj = nullptr;   // LayerNorm-folded vocabulary projection: column sums, W . beta
  bool w8 = false;              // decoder weights stored as 8-bit packed fragments
  bool cq_fold = false;         // f16 decoder weights: the fused out-projection + cross-Q stage is available (p_cqo)
  f16* dxh = nullptr;           // f16 row-major copy of the layer input rows (written by the embedding / FFN2 epilogues)
  unsigned long long* d_prof;   // [L*8][16] stamp rows, one per layer kernel (wis_debug_phase_cycles / wis_debug_timeline)
  bool prof_on; bool prof_all;
  size_t enc_part_cap = 0;      // (utterance, head, query tile) triples the split-key encoder attention buffers were sized for
  std::atomic_flag busy = ATOMIC_FLAG_INIT;   // one compute call at a time per handle (BusyGuard)
};

namespace {

// Bump allocator over large slabs: the ~700 tensors and buffers of a replica come from a handful of hipMalloc calls, so the
// small per-layer parameters (biases, LayerNorm vectors: 2.5-10 KB) sit next to their matrices inside the same large-page
// fragments instead of each owning a 4 KB page - every decode kernel touches 3-5 of them on its critical path, and a
// translation miss there costs more than the kernel's whole weight stream.
constexpr size_t SLAB_BYTES = (size_t)1 << 30;
// K-splits of the encoder's FFN2 at M rows (0 = one workgroup per output tile over the whole K): split while the 128x128 tiles
// are too few for 256 CUs; four ways while that still leaves <= 512 workgroups, else two (the split length stays a multiple of
// the 64-deep k-tile for every Whisper width)
static int enc_splitk(int d, int M) {
  const int tiles = (d / 128) * cdiv(M, 128);
  if (tiles >= 200) return 0;
  return (tiles <= 128 && (4 * d) % (4 * 64) == 0) ? 4 : 2;
}
template <class T>
int dalloc(wis_model* m, T** p, size_t n_elems) {
  size_t b = n_elems * sizeof(T); if (b == 0) b = 16;
  const size_t align = b >= ((size_t)1 << 20) ? ((size_t)1 << 16) : 256;
  size_t pad = (align - (reinterpret_cast<uintptr_t>(m->slab_cur) & (align - 1))) & (align - 1);
  if (!m->slab_cur || pad + b > m->slab_left) {
    const size_t sb = b > SLAB_BYTES ? ((b + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) : SLAB_BYTES;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, sb);
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", sb, hipGetErrorString(e)); return WIS_E_NOMEM; }
    m->allocs.push_back(q); m->slab_cur = static_cast<char*>(q); m->slab_left = sb; pad = 0;
  }
  m->slab_cur += pad; m->slab_left -= pad;
  *p = reinterpret_cast<T*>(m->slab_cur);
  m->slab_cur += b; m->slab_left -= b; m->bytes += b + pad;
  return WIS_OK;
}
inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

struct TensorSrc { const void* p; int f16; int64_t rows, cols; };

struct Loader {
  const wis_tensor_t* t; int n; const char* arena;   // arena: DEVICE base pointer
  size_t arena_bytes;
  const wis_tensor_t* find(const std::string& name) const {
    for (int i = 0; i < n; ++i) if (name == t[i].name) return &t[i];
    return nullptr;
  }
  int get(const std::string& name, int64_t rows, int64_t cols, TensorSrc* out) const {
    const wis_tensor_t* x = find(name);
    if (!x) { set_error("weight '%s' missing from the index", name.c_str()); return WIS_E_FORMAT; }
    int64_t ne = 1; for (int i = 0; i < x->rank; ++i) ne *= x->shape[i];
    if (ne != rows * cols) { set_error("weight '%s': %lld elements, expected %lld", name.c_str(), (long long)ne, (long long)(rows * cols)); return WIS_E_FORMAT; }
    const size_t es = x->dtype == WIS_DT_F16 ? 2 : 4;
    if (x->dtype != WIS_DT_F16 && x->dtype != WIS_DT_F32) { set_error("weight '%s': dtype %d unsupported", name.c_str(), x->dtype); return WIS_E_FORMAT; }
    if (x->offset + (size_t)ne * es > arena_bytes) { set_error("weight '%s' exceeds the arena", name.c_str()); return WIS_E_FORMAT; }
    out->p = arena + x->offset; out->f16 = x->dtype == WIS_DT_F16; out->rows = rows; out->cols = cols;
    return WIS_OK;
  }
};

int to_f32(wis_model* m, const Loader& L, const std::string& name, int64_t n, float** out, int64_t n_scale = 0, float scale = 1.f) {
  TensorSrc s; WIS_RET(L.get(name, n, 1, &s));
  if (!*out) WIS_RET(dalloc(m, out, (size_t)n));          // *out preset: a view into a block the caller allocated
  hipLaunchKernelGGL(convert_kernel, dim3(blocks_for(n)), dim3(256), 0, m->st, s.p, s.f16, *out, 0, n, (int64_t)1, (int64_t)1, n_scale, scale);
  return WIS_OK;
}
int to_f16_mat(wis_model* m, const Loader& L, const std::string& name, int64_t rows, int64_t cols, f16** out, int64_t n_scale = 0, float scale = 1.f) {
  TensorSrc s; WIS_RET(L.get(name, rows, cols, &s));
  if (!*out) WIS_RET(dalloc(m, out, (size_t)rows * cols));
  hipLaunchKernelGGL(convert_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, m->st, s.p, s.f16, *out, 1, rows, cols, cols, n_scale, scale);
  return WIS_OK;
}
// row-major source -> MFMA-fragment packed (through a temporary f16 image)
// ln_gamma / ln_beta (device, fp32 [K]) != nullptr: the projection follows a LayerNorm - fold it (fold_ln_kernel): `bias`
// (device fp32, >= N entries, already loaded) receives W . beta, *csum_out the column sums of the folded weights.
int to_packed(wis_model* m, const Loader& L, const std::string& name, int N, int K, f16** out, f16* tmp, int n_scale = 0, float scale = 1.f, int* npad_out = nullptr,
              float** scale_out = nullptr, const float* ln_gamma = nullptr, const float* ln_beta = nullptr, float* bias = nullptr, float** csum_out = nullptr) {
  TensorSrc s; WIS_RET(L.get(name, N, K, &s));
  const int rows = gemv_rows_for(N, K);
  const int Npad = cdiv(N, rows) * rows;
  hipLaunchKernelGGL(convert_kernel, dim3(blocks_for((int64_t)N * K)), dim3(256), 0, m->st, s.p, s.f16, tmp, 1, (int64_t)N, (int64_t)K, (int64_t)K, (int64_t)0, 1.f);
  if (ln_gamma) {
    if (!bias || !csum_out) { set_error("to_packed: LayerNorm folding needs a bias vector and a column-sum output"); return WIS_E_ARG; }
    WIS_RET(dalloc(m, csum_out, (size_t)Npad));
    WIS_HIP_CHECK(hipMemsetAsync(*csum_out, 0, (size_t)Npad * 4, m->st));
    WIS_RET(launch_fold_ln(m->st, tmp, ln_gamma, ln_beta, bias, *csum_out, N, K, n_scale, scale));
  }
  if (m->w8 && scale_out) {       // 8-bit fragments + per-row dequantisation scales (the query scaling is folded into the scales)
    unsigned char* q8 = nullptr;
    WIS_RET(dalloc(m, &q8, (size_t)Npad * K));
    WIS_RET(dalloc(m, scale_out, (size_t)Npad));
    WIS_RET(launch_pack_gemv8(m->st, tmp, q8, *scale_out, N, Npad, K, n_scale, scale));
    if (ln_gamma) WIS_RET(launch_csum8(m->st, tmp, *scale_out, *csum_out, N, K, n_scale, scale));   // sums of what the MFMA will really see
    *out = reinterpret_cast<f16*>(q8);
  } else {
    WIS_RET(dalloc(m, out, (size_t)Npad * K));
    WIS_RET(launch_pack_gemv(m->st, tmp, *out, N, Npad, K, n_scale, scale, rows));
  }
  if (npad_out) *npad_out = Npad;
  return WIS_OK;
}

int load_weights(wis_model* m, const Loader& L) {
  const wis_config_t& c = m->cfg;
  const int d = c.d_model, V = c.n_vocab;
  const float qs = 0.125f;   // 1/sqrt(64), folded into the query projections (exact: power of two)
  f16* tmp = nullptr;        // staging for packed conversions: largest matrix = embeddings [V][d]
  f16 *fcat = nullptr, *fwot = nullptr, *fwqo = nullptr;      // cross-Q fold staging: [W'q | W'q Wo] [d][2d], Wo^T, W'q Wo
  {
    size_t big = (size_t)V * d; if ((size_t)4 * d * d > big) big = (size_t)4 * d * d;
    WIS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tmp), big * 2));
    if (m->cq_fold) {
      WIS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&fcat), (size_t)2 * d * d * 2));
      WIS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&fwot), (size_t)d * d * 2));
      WIS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&fwqo), (size_t)d * d * 2));
    }
  }
  int rc = WIS_OK;
  do {
    // ---- encoder
    {
      TensorSrc s;
      if ((rc = L.get("encoder/conv1/weight", d, (int64_t)c.n_mels * 3, &s))) break;
      if ((rc = dalloc(m, &m->w_conv1, (size_t)d * 320))) break;     // K = 3*96 = 288 padded to 320 = 5 x 64
      hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks_for((int64_t)d * 320)), dim3(256), 0, m->st, s.p, s.f16, m->w_conv1, d, c.n_mels, 96, 320);
      if ((rc = L.get("encoder/conv2/weight", d, (int64_t)d * 3, &s))) break;
      if ((rc = dalloc(m, &m->w_conv2, (size_t)d * 3 * d))) break;
      hipLaunchKernelGGL(conv_pack_kernel, dim3(blocks_for((int64_t)d * 3 * d)), dim3(256), 0, m->st, s.p, s.f16, m->w_conv2, d, d, d, 3 * d);
    }
    if ((rc = to_f32(m, L, "encoder/conv1/bias", d, &m->b_conv1))) break;
    if ((rc = to_f32(m, L, "encoder/conv2/bias", d, &m->b_conv2))) break;
    if (L.find("encoder/position_encodings/encodings")) {
      if ((rc = to_f32(m, L, "encoder/position_encodings/encodings", (int64_t)c.n_audio_ctx * d, &m->enc_pos))) break;
    } else {
      // Whisper sinusoids: inc = ln(10000)/(d/2-1); pos[t] = [sin(t inv) | cos(t inv)]
      std::vector<float> pe((size_t)c.n_audio_ctx * d);
      const int half = d / 2; const double inc = log(10000.0) / (half - 1);
      for (int t = 0; t < c.n_audio_ctx; ++t)
        for (int i = 0; i < half; ++i) {
          const float inv = (float)exp(-inc * i);           // fp32 table like torch.exp(float32)
          const float a = (float)t * inv;
          pe[(size_t)t * d + i] = sinf(a); pe[(size_t)t * d + half + i] = cosf(a);
        }
      if ((rc = dalloc(m, &m->enc_pos, pe.size()))) break;
      WIS_HIP_CHECK(hipMemcpyAsync(m->enc_pos, pe.data(), pe.size() * 4, hipMemcpyHostToDevice, m->st));
      WIS_HIP_CHECK(hipStreamSynchronize(m->st));
    }
    if ((rc = to_f32(m, L, "encoder/layer_norm/gamma", d, &m->enc_ln_g))) break;
    if ((rc = to_f32(m, L, "encoder/layer_norm/beta", d, &m->enc_ln_b))) break;
    m->enc.resize(c.n_enc_layers);
    for (int l = 0; l < c.n_enc_layers && !rc; ++l) {
      const std::string p = "encoder/layer_" + std::to_string(l) + "/";
      EncLayerW& w = m->enc[l];
      if ((rc = to_f32(m, L, p + "self_attention/layer_norm/gamma", d, &w.ln1_g))) break;
      if ((rc = to_f32(m, L, p + "self_attention/layer_norm/beta", d, &w.ln1_b))) break;
      // (the lazy-reference attention loop takes exp2 of the MFMA result directly: log2(e) rides on the query projection too)
      const float qs_enc = enc_attn_lazy() ? qs * 1.4426950408889634f : qs;
      if ((rc = to_f16_mat(m, L, p + "self_attention/linear_0/weight", 3 * d, d, &w.w_qkv, d, qs_enc))) break;
      if ((rc = to_f32(m, L, p + "self_attention/linear_0/bias", 3 * d, &w.b_qkv, d, qs_enc))) break;
      if ((rc = to_f16_mat(m, L, p + "self_attention/linear_1/weight", d, d, &w.w_out))) break;
      if ((rc = to_f32(m, L, p + "self_attention/linear_1/bias", d, &w.b_out))) break;
      if ((rc = to_f32(m, L, p + "ffn/layer_norm/gamma", d, &w.ln2_g))) break;
      if ((rc = to_f32(m, L, p + "ffn/layer_norm/beta", d, &w.ln2_b))) break;
      if ((rc = to_f16_mat(m, L, p + "ffn/linear_0/weight", 4 * d, d, &w.w_f1))) break;
      if ((rc = to_f32(m, L, p + "ffn/linear_0/bias", 4 * d, &w.b_f1))) break;
      if ((rc = to_f16_mat(m, L, p + "ffn/linear_1/weight", d, 4 * d, &w.w_f2))) break;
      if ((rc = to_f32(m, L, p + "ffn/linear_1/bias", d, &w.b_f2))) break;
    }
    if (rc) break;
    // ---- decoder
    if ((rc = to_f16_mat(m, L, "decoder/embeddings/weight", V, d, &m->emb))) break;
    if ((rc = to_f16_mat(m, L, "decoder/position_encodings/encodings", c.n_text_ctx, d, &m->dec_pos))) break;
    if ((rc = to_f32(m, L, "decoder/layer_norm/gamma", d, &m->dec_ln_g))) break;
    if ((rc = to_f32(m, L, "decoder/layer_norm/beta", d, &m->dec_ln_b))) break;
    // every projection that follows a LayerNorm is stored LayerNorm-folded (W o gamma, bias + W . beta, column sums): the final
    // LayerNorm into the tied vocabulary projection (which has no bias of its own) ...
    if ((rc = dalloc(m, &m->b_proj, (size_t)V + 64))) break;
    if (hipMemsetAsync(m->b_proj, 0, ((size_t)V + 64) * 4, m->st) != hipSuccess) { set_error("memset failed"); rc = WIS_E_HIP; break; }
    if ((rc = to_packed(m, L, "decoder/embeddings/weight", V, d, &m->p_proj, tmp, 0, 1.f, &m->n_vocab_pad, &m->s_proj, m->dec_ln_g, m->dec_ln_b, m->b_proj, &m->c_proj))) break;
    m->dec.resize(c.n_dec_layers);
    if ((rc = dalloc(m, &m->w_ckv_all, (size_t)c.n_dec_layers * 2 * d * d))) break;      // one [L * 2d][d] matrix: a single GEMM projects every layer
    if ((rc = dalloc(m, &m->b_ckv_all, (size_t)c.n_dec_layers * 2 * d))) break;
    for (int l = 0; l < c.n_dec_layers && !rc; ++l) {
      const std::string p = "decoder/layer_" + std::to_string(l) + "/";
      DecLayerW& w = m->dec[l];
      if ((rc = to_f32(m, L, p + "self_attention/layer_norm/gamma", d, &w.ln1_g))) break;
      if ((rc = to_f32(m, L, p + "self_attention/layer_norm/beta", d, &w.ln1_b))) break;
      // ... and ln1 -> QKV, ln2 -> cross-Q, ln3 -> FFN1 of every layer (bias first: the fold adds W . beta to it)
      if ((rc = to_f32(m, L, p + "self_attention/linear_0/bias", 3 * d, &w.b_qkv, d, qs))) break;
      if ((rc = to_packed(m, L, p + "self_attention/linear_0/weight", 3 * d, d, &w.p_qkv, tmp, d, qs, nullptr, &w.s_qkv, w.ln1_g, w.ln1_b, w.b_qkv, &w.c_qkv))) break;
      if ((rc = to_packed(m, L, p + "self_attention/linear_1/weight", d, d, &w.p_out, tmp, 0, 1.f, nullptr, &w.s_out))) break;
      if ((rc = to_f32(m, L, p + "self_attention/linear_1/bias", d, &w.b_out))) break;
      if ((rc = to_f32(m, L, p + "attention/layer_norm/gamma", d, &w.ln2_g))) break;
      if ((rc = to_f32(m, L, p + "attention/layer_norm/beta", d, &w.ln2_b))) break;
      if ((rc = to_f32(m, L, p + "attention/linear_0/bias", d, &w.b_cq, d, qs))) break;
      if ((rc = to_packed(m, L, p + "attention/linear_0/weight", d, d, &w.p_cq, tmp, d, qs, nullptr, &w.s_cq, w.ln2_g, w.ln2_b, w.b_cq, &w.c_cq))) break;
      if (m->cq_fold) {
        // Cross-attention query folded through the self-attention output projection (one dependent stage less per layer):
        //   x1 = x0 + Wo a + bo,   q = rs(x1) (W'q x1 - mu(x1) c) + b'      (LayerNorm-folded form, W'q = (Wq o gamma) / 8)
        //   W'q x1 = W'q x0 + (W'q Wo) a + W'q bo  =: q_raw  - computable from the LAYER INPUT x0 and the attention output a, i.e.
        // in the same launch as the out-projection; mu / rs of x1 are applied by the cross-attention kernel (dec_kernels.hip).
        // tmp holds f16(Wq o gamma) (to_packed above); W'q Wo is rounded to f16 like every other stored weight.
        TensorSrc so; if ((rc = L.get(p + "self_attention/linear_1/weight", d, d, &so))) break;
        hipLaunchKernelGGL(convert_kernel, dim3(blocks_for((int64_t)d * d)), dim3(256), 0, m->st, tmp, 1, fcat, 1, (int64_t)d, (int64_t)d, (int64_t)2 * d, (int64_t)d, qs);   // left half: W'q
        hipLaunchKernelGGL(transpose_to_f16_kernel, dim3(cdiv(d, 32), cdiv(d, 32)), dim3(256), 0, m->st, so.p, so.f16, fwot, d, d);
        GemmP gp = gemm_plain(fcat, 2 * d, fwot, d, d, d);
        if ((rc = launch_gemm_generic(m->st, gp, nullptr, nullptr, fwqo, 0))) break;                                     // W'q . Wo  (f16 inputs, fp32 accumulate)
        hipLaunchKernelGGL(convert_kernel, dim3(blocks_for((int64_t)d * d)), dim3(256), 0, m->st, fwqo, 1, fcat + d, 1, (int64_t)d, (int64_t)d, (int64_t)2 * d, (int64_t)0, 1.f);   // right half
        if ((rc = dalloc(m, &w.p_cqo, (size_t)d * 2 * d))) break;
        if ((rc = launch_pack_gemv(m->st, fcat, w.p_cqo, d, d, 2 * d, 0, 1.f, 16))) break;
        if ((rc = dalloc(m, &w.b_cqo, (size_t)d))) break;
        hipLaunchKernelGGL(matvec_rows_kernel, dim3(d), dim3(64), 0, m->st, fcat, 2 * d, w.b_out, w.b_cqo, d, d);
      }
      w.w_ckv = m->w_ckv_all + (size_t)l * 2 * d * d; w.b_ckv = m->b_ckv_all + (size_t)l * 2 * d;
      if ((rc = to_f16_mat(m, L, p + "attention/linear_1/weight", 2 * d, d, &w.w_ckv))) break;
      if ((rc = to_f32(m, L, p + "attention/linear_1/bias", 2 * d, &w.b_ckv))) break;
      if ((rc = to_packed(m, L, p + "attention/linear_2/weight", d, d, &w.p_cout, tmp, 0, 1.f, nullptr, &w.s_cout))) break;
      if ((rc = to_f32(m, L, p + "attention/linear_2/bias", d, &w.b_cout))) break;
      if ((rc = to_f32(m, L, p + "ffn/layer_norm/gamma", d, &w.ln3_g))) break;
      if ((rc = to_f32(m, L, p + "ffn/layer_norm/beta", d, &w.ln3_b))) break;
      if ((rc = to_f32(m, L, p + "ffn/linear_0/bias", 4 * d, &w.b_f1))) break;
      if ((rc = to_packed(m, L, p + "ffn/linear_0/weight", 4 * d, d, &w.p_f1, tmp, 0, 1.f, nullptr, &w.s_f1, w.ln3_g, w.ln3_b, w.b_f1, &w.c_f1))) break;
      if ((rc = to_packed(m, L, p + "ffn/linear_1/weight", d, 4 * d, &w.p_f2, tmp, 0, 1.f, nullptr, &w.s_f2))) break;
      if ((rc = to_f32(m, L, p + "ffn/linear_1/bias", d, &w.b_f2))) break;
    }
  } while (0);
  hipError_t e = hipStreamSynchronize(m->st);
  hipFree(tmp); hipFree(fcat); hipFree(fwot); hipFree(fwqo);
  if (rc) return rc;
  if (e != hipSuccess) { set_error("weight conversion failed: %s", hipGetErrorString(e)); return WIS_E_HIP; }
  return WIS_OK;
}

int alloc_buffers(wis_model* m) {
  const wis_config_t& c = m->cfg;
  const int d = c.d_model, H = c.n_heads, Bm = c.max_batch, T = c.n_audio_ctx, L = c.n_dec_layers;
  const int slots = Bm * c.max_beam, ctx = c.n_text_ctx;
  m->Tpad = cdiv(T, 64) * 64;
  WIS_RET(dalloc(m, &m->img, (size_t)Bm * 3002 * 96 + 64));     // +64: the last window's zero-weighted over-read
  WIS_RET(dalloc(m, &m->c1, (size_t)Bm * 3002 * d));
  WIS_RET(dalloc(m, &m->x, (size_t)Bm * T * d));
  WIS_RET(dalloc(m, &m->xn, (size_t)Bm * T * d));
  WIS_RET(dalloc(m, &m->qk, (size_t)Bm * T * 2 * d));
  WIS_RET(dalloc(m, &m->vt, (size_t)Bm * H * 64 * m->Tpad));
  WIS_RET(dalloc(m, &m->ao, (size_t)Bm * T * d));
  WIS_RET(dalloc(m, &m->hbuf, (size_t)Bm * T * 4 * d));
  WIS_RET(dalloc(m, &m->mem, (size_t)Bm * T * d));
  {  // split-key encoder attention (small batches): per-workgroup softmax states and the pair tickets
    int bs = 600 / (H * cdiv(T, 128)); if (bs > Bm) bs = Bm;
    if (bs >= 1) {
      WIS_RET(dalloc(m, &m->enc_part, enc_attention_part_floats(bs, T, H)));
      WIS_RET(dalloc(m, &m->enc_cnt, (size_t)bs * H * cdiv(T, 128)));
      WIS_HIP_CHECK(hipMemsetAsync(m->enc_cnt, 0, (size_t)bs * H * cdiv(T, 128) * 4, m->st));
      m->enc_part_cap = (size_t)bs * H * cdiv(T, 128);
    }
  }
  {  // fp32 partial tiles of the K-split FFN2 (small row counts only)
    size_t sk = 0;
    for (int b = 1; b <= Bm; ++b) { const size_t need = (size_t)enc_splitk(d, b * T) * b * T * d; if (need > sk) sk = need; }      // not monotonic in b: 4 splits, then 2, then none
    m->skbuf = nullptr;
    if (sk) WIS_RET(dalloc(m, &m->skbuf, sk));
  }
  WIS_HIP_CHECK(hipMemsetAsync(m->img, 0, ((size_t)Bm * 3002 * 96 + 64) * 2, m->st));
  WIS_HIP_CHECK(hipMemsetAsync(m->c1, 0, (size_t)Bm * 3002 * d * 2, m->st));
  WIS_HIP_CHECK(hipMemsetAsync(m->vt, 0, (size_t)Bm * H * 64 * m->Tpad * 2, m->st));
  m->kx.resize(L); m->vx.resize(L); m->kc.resize(L); m->vc.resize(L);
  m->kv_layer_stride = (size_t)slots * ctx * d;
  WIS_RET(dalloc(m, &m->kc_all, m->kv_layer_stride * L));
  WIS_RET(dalloc(m, &m->vc_all, m->kv_layer_stride * L));
  m->kx_lstride = (size_t)Bm * H * 8 * T * 8; m->vx_lstride = (size_t)Bm * H * 64 * m->Tpad;
  WIS_RET(dalloc(m, &m->kx_all, m->kx_lstride * L));
  WIS_RET(dalloc(m, &m->vx_all, m->vx_lstride * L));
  WIS_HIP_CHECK(hipMemsetAsync(m->vx_all, 0, m->vx_lstride * L * 2, m->st));
  for (int l = 0; l < L; ++l) {
    m->kx[l] = m->kx_all + (size_t)l * m->kx_lstride;
    m->vx[l] = m->vx_all + (size_t)l * m->vx_lstride;
    m->kc[l] = m->kc_all + (size_t)l * slots * ctx * d;     // one block per cache: kv_reorder_kernel walks the layers by stride
    m->vc[l] = m->vc_all + (size_t)l * slots * ctx * d;
  }
  WIS_RET(dalloc(m, &m->dx, (size_t)MAX_ROWS * d));
  WIS_RET(dalloc(m, &m->dq, (size_t)MAX_ROWS * d));
  WIS_RET(dalloc(m, &m->dq2, (size_t)MAX_ROWS * d));
  WIS_RET(dalloc(m, &m->dao, (size_t)MAX_ROWS * d));
  WIS_RET(dalloc(m, &m->dh, (size_t)MAX_ROWS * 4 * d));
  WIS_RET(dalloc(m, &m->dln, (size_t)MAX_ROWS * d));
  WIS_RET(dalloc(m, &m->dxh, (size_t)MAX_ROWS * d));
  {  // fragment images (kernels.hpp xf_index): [K/32][3 row blocks][64][8] f16; zeroed once (rows >= M are never written)
    const size_t blk = (size_t)(MAX_ROWS / 16) * 64 * 8;
    WIS_RET(dalloc(m, &m->dxf, (size_t)(d / 32) * blk)); WIS_RET(dalloc(m, &m->daoxf, (size_t)(d / 32) * blk)); WIS_RET(dalloc(m, &m->dhxf, (size_t)(4 * d / 32) * blk));
    WIS_RET(dalloc(m, &m->dstat, (size_t)MAX_ROWS * (d / 16) * 2));
    WIS_HIP_CHECK(hipMemsetAsync(m->dxf, 0, (size_t)(d / 32) * blk * 2, m->st)); WIS_HIP_CHECK(hipMemsetAsync(m->daoxf, 0, (size_t)(d / 32) * blk * 2, m->st));
    WIS_HIP_CHECK(hipMemsetAsync(m->dhxf, 0, (size_t)(4 * d / 32) * blk * 2, m->st));
  }
  WIS_RET(dalloc(m, &m->logits, (size_t)MAX_ROWS * m->n_vocab_pad));
  // (row groups of the cross-attention: utterances - or, verifying a beam-search draft, up to MAX_ROWS / 16 groups of 16 tree rows of ONE utterance)
  const int Bg = Bm > MAX_ROWS / 16 ? Bm : MAX_ROWS / 16;
  WIS_RET(dalloc(m, &m->part, (size_t)Bg * H * 16 * 16 * 66));
  WIS_RET(dalloc(m, &m->counters, (size_t)Bg * H));
  WIS_HIP_CHECK(hipMemsetAsync(m->counters, 0, (size_t)Bg * H * 4, m->st));
  {
    // measured (decode ms per utterance batch, 17 steps): 8 utterances 40.7 unsplit / 38.6 two slices / 38.9 four; 12: 46.7 two / 46.9 four;
    // 16: 60.9 unsplit / 58.7 two / 59.5 four
    static const int env_ks = getenv("WIS_FRAG_KSPLIT") ? atoi(getenv("WIS_FRAG_KSPLIT")) : 2;      // 1: no split (A/B switch)
    m->gf_ksplit = env_ks >= 1 && env_ks <= 8 ? env_ks : 2;
    const size_t nt = (size_t)cdiv(d, 16);
    WIS_RET(dalloc(m, &m->gf_part, nt * m->gf_ksplit * (MAX_ROWS / 16) * 64 * 4));
    WIS_RET(dalloc(m, &m->gf_cnt, nt));
    WIS_HIP_CHECK(hipMemsetAsync(m->gf_cnt, 0, nt * 4, m->st));
  }
  {
    const int bh = Bm * H < CA_SPIN_MAX_BH ? Bm * H : CA_SPIN_MAX_BH;
    WIS_RET(dalloc(m, &m->ca_gran, (size_t)bh * 6 * 8 * 66));
    WIS_RET(dalloc(m, &m->ca_epoch, (size_t)bh + 1));
    WIS_HIP_CHECK(hipMemsetAsync(m->ca_gran, 0, (size_t)bh * 6 * 8 * 66 * 8, m->st));
    WIS_HIP_CHECK(hipMemsetAsync(m->ca_epoch, 0, ((size_t)bh + 1) * 4, m->st));
    WIS_RET(dalloc(m, &m->sa_gran, (size_t)H * 8 * 3 * 64));
    WIS_RET(dalloc(m, &m->sa_epoch, (size_t)H));
    WIS_HIP_CHECK(hipMemsetAsync(m->sa_gran, 0, (size_t)H * 8 * 3 * 64 * 8, m->st));
    WIS_HIP_CHECK(hipMemsetAsync(m->sa_epoch, 0, (size_t)H * 4, m->st));
  }
  WIS_RET(dalloc(m, &m->rm.tok, MAX_ROWS)); WIS_RET(dalloc(m, &m->rm.pos, MAX_ROWS));
  WIS_RET(dalloc(m, &m->rm.slot, MAX_ROWS)); WIS_RET(dalloc(m, &m->rm.lslot, MAX_ROWS));
  WIS_RET(dalloc(m, &m->rm_win.tok, MAX_ROWS)); WIS_RET(dalloc(m, &m->rm_win.pos, MAX_ROWS)); WIS_RET(dalloc(m, &m->rm_win.slot, MAX_ROWS)); WIS_RET(dalloc(m, &m->rm_win.lslot, MAX_ROWS));
  const int max_new = 256, max_hyp = MAX_HYP;
  WIS_RET(dalloc(m, &m->bs.step_u, Bm)); WIS_RET(dalloc(m, &m->bs.done, Bm)); WIS_RET(dalloc(m, &m->bs.n_hyp, Bm));
  WIS_RET(dalloc(m, &m->bs.cum, slots));
  WIS_RET(dalloc(m, &m->bs.alive, (size_t)slots * max_new));
  WIS_RET(dalloc(m, &m->bs.parent, (size_t)slots));
  WIS_RET(dalloc(m, &m->bs.hyp_score, (size_t)Bm * max_hyp)); WIS_RET(dalloc(m, &m->bs.hyp_len, (size_t)Bm * max_hyp));
  WIS_RET(dalloc(m, &m->bs.hyp_tok, (size_t)Bm * max_hyp * max_new));
  WIS_RET(dalloc(m, &m->bs.all_done, 4));
  WIS_RET(dalloc(m, &m->bs.tick, 4));
  WIS_RET(dalloc(m, &m->vstep, MAX_ROWS)); WIS_RET(dalloc(m, &m->pick_tok, MAX_ROWS)); WIS_RET(dalloc(m, &m->pick_lp, MAX_ROWS));
  WIS_RET(dalloc(m, &m->bs.traj, (size_t)Bm * 256 * MAX_R * 2));
  WIS_RET(dalloc(m, &m->d_draft, (size_t)256 * MAX_R * 2)); WIS_RET(dalloc(m, &m->d_anc, (size_t)MAX_ROWS * 32)); WIS_RET(dalloc(m, &m->d_vstate, DRAFT_VS_INTS)); WIS_RET(dalloc(m, &m->d_base, MAX_ROWS));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.tick, 0, 16, m->st));
  WIS_RET(dalloc(m, &m->bs.out_ids, (size_t)Bm * max_new)); WIS_RET(dalloc(m, &m->bs.out_len, Bm)); WIS_RET(dalloc(m, &m->bs.out_score, Bm));
  WIS_RET(dalloc(m, &m->st_max, (size_t)MAX_ROWS * STAT_SUB)); WIS_RET(dalloc(m, &m->st_sum, (size_t)MAX_ROWS * STAT_SUB));
  WIS_RET(dalloc(m, &m->st_val, (size_t)MAX_ROWS * STAT_SUB * MAX_CAND)); WIS_RET(dalloc(m, &m->st_idx, (size_t)MAX_ROWS * STAT_SUB * MAX_CAND));
  WIS_RET(dalloc(m, &m->d_in, (size_t)Bm * WIS_N_SAMPLES));
  WIS_RET(dalloc(m, &m->d_nsamp, Bm));
  WIS_RET(dalloc(m, &m->lm_logspec, (size_t)Bm * WIS_N_MELS * WIS_N_FRAMES));
  WIS_RET(dalloc(m, &m->lm_gmax, Bm));
  WIS_RET(dalloc(m, &m->d_probs, (size_t)Bm * (c.n_lang > 0 ? c.n_lang : 1)));
  WIS_RET(dalloc(m, &m->d_prof, ((size_t)c.n_dec_layers * 8 + 2) * 16));   // + sampling kernels (tap builds)
  WIS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m->h_pin), 65536, hipHostMallocDefault));
  // fine-grained (coherent) host memory mapped into the device: beam_step_kernel's system-scope stores land here while the
  // stream is still running
  WIS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m->h_prog), HP_BYTES, hipHostMallocMapped | hipHostMallocCoherent));
  memset(m->h_prog, 0, HP_BYTES);
  { void* dp = nullptr; WIS_HIP_CHECK(hipHostGetDevicePointer(&dp, m->h_prog, 0)); m->bs.host = static_cast<unsigned long long*>(dp); }
  m->bs.giveup = m->ca_epoch;
  for (int i = 0; i < 8; ++i) WIS_HIP_CHECK(hipEventCreate(&m->ev[i]));
  WIS_HIP_CHECK(hipStreamCreateWithFlags(&m->st_enc, hipStreamNonBlocking));
  WIS_HIP_CHECK(hipEventCreateWithFlags(&m->ev_enc, hipEventDisableTiming));
  WIS_HIP_CHECK(hipEventCreateWithFlags(&m->ev_ckv, hipEventDisableTiming));
  WIS_HIP_CHECK(hipEventRecord(m->ev_ckv, m->st));
  return WIS_OK;
}

// ---- input -> conv1 image --------------------------------------------------------------
int stage_input(wis_model* m, const float* input, int kind, int B, hipStream_t on = nullptr) {
  hipStream_t st = on ? on : m->st;
  if (kind == WIS_IN_PCM_HOST || kind == WIS_IN_PCM_DEV) {
    const float* dp = input;
    if (kind == WIS_IN_PCM_HOST) {
      WIS_HIP_CHECK(hipMemcpyAsync(m->d_in, input, (size_t)B * WIS_N_SAMPLES * 4, hipMemcpyHostToDevice, st));
      dp = m->d_in;
    }
    int64_t* hn = reinterpret_cast<int64_t*>(m->h_pin);
    for (int b = 0; b < B; ++b) hn[b] = WIS_N_SAMPLES;
    WIS_HIP_CHECK(hipMemcpyAsync(m->d_nsamp, hn, (size_t)B * 8, hipMemcpyHostToDevice, st));
    WIS_RET(logmel_device(m->ctx, st, m->lm_logspec, m->lm_gmax, dp, WIS_N_SAMPLES, m->d_nsamp, B, nullptr, m->img));
  } else if (kind == WIS_IN_MEL_HOST || kind == WIS_IN_MEL_DEV) {
    const float* dm = input;
    if (kind == WIS_IN_MEL_HOST) {
      WIS_HIP_CHECK(hipMemcpyAsync(m->d_in, input, (size_t)B * 80 * 3000 * 4, hipMemcpyHostToDevice, st));
      dm = m->d_in;
    }
    hipLaunchKernelGGL(mel_to_image_kernel, dim3(cdiv(3000, 64), B), dim3(256), 0, st, dm, m->img);
  } else { set_error("bad input_kind %d", kind); return WIS_E_ARG; }
  return WIS_OK;
}

// ---- encoder + cross K/V ---------------------------------------------------------------
int run_encoder(wis_model* m, int B, hipStream_t on = nullptr) {
  const wis_config_t& c = m->cfg; hipStream_t st = on ? on : m->st;
  const int d = c.d_model, H = c.n_heads, T = c.n_audio_ctx, M = B * T;
  {  // conv1: implicit im2col over the [3002][96] image, K = 288 (+32 zero-weighted columns that read into the next row)
    GemmP p; p.klen = 0; p.A = m->img; p.a_bs = (int64_t)3002 * 96; p.a_rs = 96; p.a_rpb = 3000; p.W = m->w_conv1; p.M = B * 3000; p.N = d; p.K = 320;
    WIS_RET(launch_gemm_conv1(st, p, m->b_conv1, m->c1, 3000));
  }
  {  // conv2 (stride 2) + GELU + positions -> fp32 residual stream
    GemmP p; p.klen = 0; p.A = m->c1; p.a_bs = (int64_t)3002 * d; p.a_rs = 2 * d; p.a_rpb = T; p.W = m->w_conv2; p.M = M; p.N = d; p.K = 3 * d;
    WIS_RET(launch_gemm_conv2(st, p, m->b_conv2, m->enc_pos, m->x, T));
  }
  // Few row tiles (one utterance of the larger models): FFN2 (N = d, K = 4d) has too few 128x128 tiles for 256 CUs, so K is split
  // over workgroups (240-480 of them) and the reduction launch carries the LayerNorm of whatever consumes the rows next - the next
  // layer's ln1, or ln_post after the last layer.  Measured for large-v2 at M = 1500 with the weights streamed from HBM
  // (tools/gemm_lab.hip): 64x128 tiles over the whole K 52 us; 2 splits 42.6 + 6.0 (reduce) + 6.2 (LayerNorm) us; 4 splits + fused
  // reduce-LayerNorm: see profiles/.
  const int splits = enc_splitk(d, M);
  const bool split = splits > 0;
  bool xn_ready = false;
  for (int l = 0; l < c.n_enc_layers; ++l) {
    const EncLayerW& w = m->enc[l];
    if (!xn_ready) WIS_RET(launch_layernorm(st, m->x, w.ln1_g, w.ln1_b, m->xn, M, d));
    WIS_RET(launch_gemm_qkv(st, gemm_plain(m->xn, d, w.w_qkv, M, 3 * d, d), w.b_qkv, m->qk, m->vt, d, T, m->Tpad, H));
    WIS_RET(launch_enc_attention(st, m->qk, m->vt, m->ao, B, T, m->Tpad, H, m->enc_part, m->enc_cnt, m->enc_part_cap));
    WIS_RET(launch_gemm_generic(st, gemm_plain(m->ao, d, w.w_out, M, d, d), w.b_out, m->x, m->x, 2 | 4));
    WIS_RET(launch_layernorm(st, m->x, w.ln2_g, w.ln2_b, m->xn, M, d));
    WIS_RET(launch_gemm_generic(st, gemm_plain(m->xn, d, w.w_f1, M, 4 * d, d), w.b_f1, nullptr, m->hbuf, 1));
    if (split) {
      const bool last = l + 1 == c.n_enc_layers;
      const float* g = last ? m->enc_ln_g : m->enc[l + 1].ln1_g;
      const float* b = last ? m->enc_ln_b : m->enc[l + 1].ln1_b;
      WIS_RET(launch_gemm_splitk_resid(st, gemm_plain(m->hbuf, 4 * d, w.w_f2, M, d, 4 * d), splits, m->skbuf, w.b_f2, m->x, m->x, g, b, last ? m->mem : m->xn));
      xn_ready = true;
    } else {
      WIS_RET(launch_gemm_generic(st, gemm_plain(m->hbuf, 4 * d, w.w_f2, M, d, 4 * d), w.b_f2, m->x, m->x, 2 | 4));
    }
  }
  if (!split) WIS_RET(launch_layernorm(st, m->x, m->enc_ln_g, m->enc_ln_b, m->mem, M, d));
  return WIS_OK;
}
int run_cross_kv(wis_model* m, int B) {
  const wis_config_t& c = m->cfg;
  const int d = c.d_model, T = c.n_audio_ctx;
  // every decoder layer's K/V projection of the encoder memory in one GEMM (weights, biases and outputs are single blocks)
  return launch_gemm_crosskv(m->st, gemm_plain(m->mem, d, m->w_ckv_all, B * T, c.n_dec_layers * 2 * d, d), m->b_ckv_all, m->kx_all, m->vx_all, d, T, m->Tpad,
                             c.n_heads, (int64_t)m->kx_lstride, (int64_t)m->vx_lstride);
}

// LayerNorm + skinny GEMM.  Up to 8 rows the LayerNorm is fused into the GEMM prologue (register resident); with more rows
// (batched decode: B*beam up to 48) every workgroup re-normalising all rows costs more than one extra launch, so the rows are
// normalised once by layernorm_kernel into an f16 buffer and the GEMM takes its f16-activation path.
static int launch_ln_gemv(wis_model* m, hipStream_t st, GemvP g) {
  if (g.M > 8 && (g.flags & GV_LN)) {   // plain normalisation (the affine part lives in the folded weights / bias), then f16 activations
    WIS_RET(launch_layernorm(st, reinterpret_cast<const float*>(g.x), nullptr, nullptr, m->dln, g.M, g.K));
    g.x = m->dln; g.csum = nullptr; g.flags &= ~GV_LN;
  }
  return launch_gemv(st, g);
}

// ---- granule hand-off of the cross-attention: when it may be used -------------------------------
// Its progress argument ("at most CA_SPIN_MAX_BH combiners spin, fewer than the chip's CUs, so a producer always finds a slot") is about
// everything that runs on the GPU at the same time, so the budget is kept per DEVICE and claimed per CALL: a call that wants the
// granule form adds its B x heads combiners to the device's count for its duration and takes the ticket form when that would
// exceed the budget.  (Until round 5 the budget was divided by the handles ALIVE on the device - a server holding several model
// sizes at four replicas each never got the granule form although at most a few of its handles decode at any time, while bench.py,
// with one handle, always did: the published decode numbers were for a path the server did not run.)  Whatever happens, a spin
// that runs out never fails a request: the flag travels with every step's progress record / is read after every tap pass, the handle
// switches to the ticket form for good and the call is run again (wis_generate / wis_detect_language / the logits taps).
static inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}
static std::atomic<int> g_spin_bh[64];
// claims whose call has returned while ONE over-run decode step of it may still be running on the handle's stream (a search that ended on
// EOT does not wait for the step queued behind the one that finished it): the combiners of that step still spin, so its share of the
// budget is released only when the event recorded behind it has completed - checked by whoever claims next on the device (advisor, round 5:
// released at return, the budget could be over-subscribed for the length of one step)
struct DeferredClaim { hipEvent_t ev; int n; };
static std::mutex g_spin_mu;
static std::vector<DeferredClaim> g_spin_deferred[64];
static void spin_collect(int device) {
  std::lock_guard<std::mutex> lk(g_spin_mu);
  auto& v = g_spin_deferred[device & 63];
  for (size_t i = 0; i < v.size();) {
    if (hipEventQuery(v[i].ev) != hipErrorNotReady) { g_spin_bh[device & 63].fetch_sub(v[i].n, std::memory_order_relaxed); v[i] = v.back(); v.pop_back(); }
    else ++i;
  }
}
struct SpinClaim {
  wis_model* m; int n = 0;
  SpinClaim(wis_model* mm, int B) : m(mm) {
    static const bool env_share = getenv("WIS_CA_SPIN_SHARED") != nullptr;      // test switch: ignore the budget (exercises the shared-GPU hazard on purpose)
    m->spin_now = false;
    spin_collect(m->device);
    if (m->spin_off) return;
    if (env_share) { m->spin_now = true; return; }
    const int need = B * m->cfg.n_heads;
    std::atomic<int>& a = g_spin_bh[m->device & 63];
    int cur = a.load(std::memory_order_relaxed);
    while (cur + need <= CA_SPIN_MAX_BH)
      if (a.compare_exchange_weak(cur, cur + need, std::memory_order_relaxed)) { n = need; m->spin_now = true; return; }
  }
  ~SpinClaim() { if (n) g_spin_bh[m->device & 63].fetch_sub(n, std::memory_order_relaxed); }
  // the call returns with work of its own still queued on `st`: hand the claim to the deferred list behind an event on that stream
  void defer(hipStream_t st) {
    if (!n) return;
    if (hipEventRecord(m->ev[6], st) != hipSuccess) return;      // (then the destructor releases as before)
    std::lock_guard<std::mutex> lk(g_spin_mu);
    auto& v = g_spin_deferred[m->device & 63];
    for (auto& d : v) if (d.ev == m->ev[6]) { d.n += n; n = 0; return; }      // (re-recorded: the earlier share now waits for the later record too)
    v.push_back({m->ev[6], n}); n = 0;
  }
  SpinClaim(const SpinClaim&) = delete;
  SpinClaim& operator=(const SpinClaim&) = delete;
};
// reads and clears the give-up flag (word 0 of the epoch block); true = a combiner gave up: results of the pass are garbage
static int spin_gave_up(wis_model* m, bool* gave_up) {
  int* h = m->h_pin + 2;
  WIS_HIP_CHECK(hipMemcpyAsync(h, m->ca_epoch, 4, hipMemcpyDeviceToHost, m->st));
  WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  *gave_up = *h != 0;
  if (*gave_up) {
    WIS_HIP_CHECK(hipMemsetAsync(m->ca_epoch, 0, 4, m->st));
    m->spin_off = true; ++m->handoff_retries;
    fprintf(stderr, "[wis_hip] device %d: decoder cross-attention granule hand-off timed out; this handle uses the ticket hand-off from now on, the call is repeated\n", m->device);
  }
  return WIS_OK;
}

// ---- one decoder forward over the current row metadata ---------------------------------
// sstride / rmul: logical-slot mapping of the rows (decode rows: beam, 1; prefill rows: beam, 0; single rows: 1, 0)
// Batched rows (8 < M <= 48): every activation a projection reads lives in HBM as an MFMA fragment image, LayerNorm statistics
// travel as per-16-column partial sums from the residual epilogues (kernels.hpp launch_gemv_frag): 8 launches per layer, no
// LayerNorm launch, no per-workgroup LDS staging of the activations.
// tw (draft verification at beam > 1): the M rows are nodes of ONE utterance's beam tree - self-attention by ancestor table (anc [M][aw], first
// window position w0), cross-attention as B = M / 16 groups of R = 16 rows that all read utterance 0's K / V
struct TreeWin { const int* anc; int w0, aw; const int* base = nullptr; };
constexpr int ANC_W = 32;      // ancestor-table entries per row = the most steps a window holds (beam 2 / 3: 32 steps = 64 / 96 rows)
static int dec_forward_frag(wis_model* m, int M, int R, int B, bool want_logits, int sstride, int rmul, int chunks, const TreeWin* tw = nullptr) {
  const wis_config_t& c = m->cfg; hipStream_t st = m->st;
  const int d = c.d_model, H = c.n_heads, T = c.n_audio_ctx, ctx = c.n_text_ctx, MB = cdiv(M, 16);
  if (tw && (M % 16 != 0 || R != 16 || B != M / 16)) { set_error("dec_forward_frag: a tree window takes whole groups of 16 rows"); return WIS_E_ARG; }
  WIS_RET(launch_dec_embed_xf(st, m->emb, m->dec_pos, m->rm.tok, m->rm.pos, m->dx, m->dxf, m->dstat, M, d, MB));
  // cross-Q folded through the self-attention out-projection (f16 weights; <= 8 rows per utterance: the cross-attention kernel's
  // statistics prologue); WIS_NO_FRAG_FOLD=1 keeps the two-launch form (A/B switch)
  static const bool no_frag_fold = getenv("WIS_NO_FRAG_FOLD") != nullptr;
  const bool fold = m->cq_fold && !no_frag_fold && R <= 8;
  auto base = [&](const void* x, const f16* Wp, const float* wscale, const float* bias, int N, int K, int flags) {
    GemvP g; memset(&g, 0, sizeof(g));
    g.x = x; g.Wp = Wp; g.wscale = wscale; g.bias = bias; g.M = M; g.N = N; g.K = K; g.flags = flags; g.xmb = MB; g.rows = 16;
    return g;
  };
  for (int l = 0; l < c.n_dec_layers; ++l) {
    const DecLayerW& w = m->dec[l];
    // (tap builds: stamp rows of layer 0's kernels, same row numbering as dec_forward: 0 QKV, 1 self-attn, 2 out-proj (+ q halves), 4 cross-attn,
    // 5 cross-out, 6 FFN1, 7 FFN2)
    unsigned long long* pr = (m->prof_on && l == 0) ? m->d_prof : nullptr;
    GemvP g = base(m->dxf, w.p_qkv, w.s_qkv, w.b_qkv, 3 * d, d, GV_LN | GV_QKV);
    g.csum = w.c_qkv; g.stat_in = m->dstat; g.q = m->dq; g.kc = m->kc[l]; g.vc = m->vc[l]; g.slot = m->rm.slot; g.pos = m->rm.pos; g.d = d; g.ctx = ctx;
    g.prof = pr;
    WIS_RET(launch_gemv_frag(st, g));
    WIS_RET(launch_dec_self_attn(st, m->dq, m->kc[l], m->vc[l], m->rm.pos, m->daoxf, M, H, d, ctx, R, sstride, rmul, nullptr, MB, tw ? tw->anc : nullptr, tw ? tw->w0 : 0, tw ? tw->aw : 0, tw ? tw->base : nullptr, m->sa_nb));
    if (fold) {
      // ONE launch, three d x d problems on 3 d / 16 workgroups: x1 = x0 + Wo a + bo (residual rows + their LayerNorm partials; nobody
      // reads x1's fragment image any more, so none is written and x0's image stays valid for the other two), q_A = W'q x0 + W'q bo
      // from the layer input's image, q_B = (W'q Wo) a from the attention output's - the two k-step halves of the packed [W'q | W'q Wo]
      // matrix of the one-utterance step (load_weights: p_cqo).  The cross-attention kernel adds the halves and finishes
      // q = rs (q_A + q_B - mu c) + b' with mu / rs merged from x1's partials: the LayerNorm-folded cross-Q projection as a launch of
      // its own (6.4 us per layer at 8 utterances) is gone.
      GemvP g3[3];
      g3[0] = base(m->daoxf, w.p_out, nullptr, w.b_out, d, d, GV_RESID);
      g3[0].y = m->dx; g3[0].ymb = MB; g3[0].stat_out = m->dstat; g3[0].prof = pr ? pr + 32 : nullptr;
      g3[1] = base(m->dxf, w.p_cqo, nullptr, w.b_cqo, d, d, GV_OUT_F32);
      g3[1].y = m->dq; g3[1].wks = 2 * d / 32; g3[1].wk0 = 0;
      g3[2] = base(m->daoxf, w.p_cqo, nullptr, nullptr, d, d, GV_OUT_F32);
      g3[2].y = m->dq2; g3[2].wks = 2 * d / 32; g3[2].wk0 = d / 32;
      WIS_RET(launch_gemv_frag3(st, g3, 3));
      WIS_RET(launch_dec_cross_attn(st, m->dq, m->kx[l], m->vx[l], m->daoxf, m->part, m->counters, B, R, H, d, T, m->Tpad, chunks, pr ? pr + 64 : nullptr, MB, m->dstat, w.c_cq, w.b_cq,
                                    m->spin_now ? m->ca_gran : nullptr, m->ca_epoch, m->dq2, 1));
    } else {
    g = base(m->daoxf, w.p_out, w.s_out, w.b_out, d, d, GV_RESID);
    g.y = m->dx; g.y_xf = m->dxf; g.ymb = MB; g.stat_out = m->dstat;
    WIS_RET(launch_gemv_frag(st, g));
    g = base(m->dxf, w.p_cq, w.s_cq, w.b_cq, d, d, GV_LN | GV_OUT_F32);
    g.csum = w.c_cq; g.stat_in = m->dstat; g.y = m->dq;
    WIS_RET(launch_gemv_frag(st, g));
    WIS_RET(launch_dec_cross_attn(st, m->dq, m->kx[l], m->vx[l], m->daoxf, m->part, m->counters, B, R, H, d, T, m->Tpad, chunks, nullptr, MB, nullptr, nullptr, nullptr,
                                  (m->spin_now && !tw) ? m->ca_gran : nullptr, m->ca_epoch, nullptr, 0, tw ? 1 : 0));
    }
    g = base(m->daoxf, w.p_cout, w.s_cout, w.b_cout, d, d, GV_RESID);
    g.y = m->dx; g.y_xf = m->dxf; g.ymb = MB; g.stat_out = m->dstat; g.prof = pr ? pr + 80 : nullptr;
    WIS_RET(launch_gemv_frag(st, g));
    g = base(m->dxf, w.p_f1, w.s_f1, w.b_f1, 4 * d, d, GV_LN | GV_GELU);
    g.csum = w.c_f1; g.stat_in = m->dstat; g.y = m->dhxf; g.ymb = MB; g.prof = pr ? pr + 96 : nullptr;
    WIS_RET(launch_gemv_frag(st, g));
    g = base(m->dhxf, w.p_f2, w.s_f2, w.b_f2, d, 4 * d, GV_RESID);
    g.y = m->dx; g.y_xf = m->dxf; g.ymb = MB; g.stat_out = m->dstat; g.prof = pr ? pr + 112 : nullptr;
    if (m->gf_ksplit > 1 && (4 * d / 32) % (4 * m->gf_ksplit) == 0) { g.ksplit = m->gf_ksplit; g.kpart = m->gf_part; g.kcnt = m->gf_cnt; }      // K = 4d over `ksplit` workgroups per n-tile
    WIS_RET(launch_gemv_frag(st, g));
  }
  if (want_logits) {
    GemvP g = base(m->dxf, m->p_proj, m->s_proj, m->b_proj, m->n_vocab_pad, d, GV_LN | GV_OUT_F32);
    g.csum = m->c_proj; g.stat_in = m->dstat; g.y = m->logits;
    WIS_RET(launch_gemv_frag(st, g));
  }
  return WIS_OK;
}

// WIS_SA_FUSE=1: the step's self-attention rides in the QKV projection's launch (dec_kernels.hip sa_consume).  OFF by default - built, correct (the GPU
// suite is green with it) and measured SLOWER, same call on MI355X: decode step 1.280 against 1.250 ms, utterance 27.38 against 26.97 ms
// (profiles/r06_sa_fuse_ab.txt).  The eager trace shows why: the fused launch averages 9.0 us where the two launches take 6.2 + 5.0, but its FLOOR
// is 7.9 us against 3.3 + 2.3: behind the last projection tile the consumer still needs granule store -> L2 -> poll (~2 us) and the whole softmax,
// which costs what the 1.6 us boundary + the stand-alone kernel's start cost once the graph replays them back to back - the same lesson as the
// persistent skeleton and the flag-gated early start: an in-launch hand-off is two fabric round trips, like a kernel boundary.
// Read per call so that a test can compare both forms in one process.
static bool sa_fuse_enabled() { const char* e = getenv("WIS_SA_FUSE"); return e && atoi(e) != 0; }
int dec_forward(wis_model* m, int M, int R, int B, bool want_logits, int sstride, int rmul, const TreeWin* tw = nullptr) {
  const wis_config_t& c = m->cfg; hipStream_t st = m->st;
  const int d = c.d_model, H = c.n_heads, T = c.n_audio_ctx, ctx = c.n_text_ctx;
  static const int env_chunks = getenv("WIS_CROSS_CHUNKS") ? atoi(getenv("WIS_CROSS_CHUNKS")) : 0;
  // 256-key chunks (6 per utterance-head): measured faster than 128-key chunks at every batch size (fewer partials to publish and combine)
  const int chunks = env_chunks ? env_chunks : 6;
  static const bool no_frag = getenv("WIS_NO_FRAG") != nullptr;      // A/B switch: the round-1 batched path (LayerNorm launches + LDS-staged rows)
  if (tw) return dec_forward_frag(m, M, R, B, want_logits, sstride, rmul, chunks, tw);
  if (M > 8 && !no_frag) return dec_forward_frag(m, M, R, B, want_logits, sstride, rmul, chunks);
  // fused out-proj + cross-Q stage (load_weights: cq_fold): f16 decoder weights, <= 8 rows (the LayerNorm-fused row counts)
  const bool fold = m->cq_fold && M <= 8;
  // (r5) what the LayerNorm-folded projections (QKV, FFN1, the vocabulary) read - WIS_B1_LN = rows | f16 | partials:
  //   rows      (default) the fp32 rows (25.6 KB per workgroup at five rows), statistics in all four waves behind the weight stream: rounds 2-5
  //   f16       (off: the statistics of rounded rows cost parity at large-v2) the f16 copy of the rows (GV_LN16: 12.8 KB) that whoever produces residual rows leaves next to them (embedding, cross-
  //             attention output projection, FFN2), statistics from those same values, still in all four waves - a third of a launch's
  //             requests through the CU's address path gone, nothing added to its tail
  //   partials  f16 rows + per-16-column (sum, M2) pairs from the producers' epilogues (GV_LNP): fewer requests still, but the merge sits in
  //             the one epilogue wave BEHIND the reduction barrier - measured slower than `rows` (1.357 against 1.345 ms per step)
  static const int ln_form = [] { const char* e = getenv("WIS_B1_LN"); return !e ? 0 : (!strcmp(e, "f16") ? 2 : (!strcmp(e, "partials") ? 1 : 0)); }();
  const bool ln_ok = fold && d % 64 == 0 && M * (d / 8) <= 13 * 256;
  // (r6) the self-attention inside the QKV projection's launch (dec_kernels.hip sa_consume): decode rows that own their KV slot, f16 weights, the
  // granule hand-off allowed for this call (it shares the cross-attention's budget and give-up protocol); WIS_SA_FUSE=0: two launches (A/B)
  const bool sa_fuse = m->spin_now && sa_fuse_enabled() && rmul == 1 && M <= 8 && !m->w8 && d % 64 == 0 && ctx >= 64 && !m->prof_on;
  const bool lnp = ln_ok && ln_form == 1 && d <= 1280, ln16 = ln_ok && ln_form == 2 && d <= 2048;
  WIS_RET(launch_dec_embed(st, m->emb, m->dec_pos, m->rm.tok, m->rm.pos, m->dx, M, d, fold ? m->dxh : nullptr, lnp ? m->dstat : nullptr));
  for (int l = 0; l < c.n_dec_layers; ++l) {
    const DecLayerW& w = m->dec[l];
    // stamp rows of this layer's 8 kernels: QKV, self-attn, out, cross-Q, cross-attn, cross-out, FFN1, FFN2
    unsigned long long* pr = (m->prof_on && (l == 0 || m->prof_all)) ? m->d_prof + (size_t)l * 8 * 16 : nullptr;
    GemvP g; memset(&g, 0, sizeof(g));
    // self-attention block
    g.x = m->dx; g.csum = w.c_qkv; g.Wp = w.p_qkv; g.wscale = w.s_qkv; g.bias = w.b_qkv; g.M = M; g.N = 3 * d; g.K = d;
    g.flags = GV_LN | GV_QKV; g.q = m->dq; g.kc = m->kc[l]; g.vc = m->vc[l]; g.slot = m->rm.slot; g.pos = m->rm.pos; g.d = d; g.ctx = ctx;
    g.prof = pr;
    if (lnp) { g.x = m->dxh; g.stat_in = m->dstat; g.flags = GV_LNP | GV_QKV; }
    if (ln16) { g.x = m->dxh; g.flags = GV_LN16 | GV_QKV; }
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    if (sa_fuse && !lnp && !ln16) { g.sa_gran = m->sa_gran; g.sa_epoch = m->sa_epoch; g.sa_flag = m->ca_epoch; g.sa_out = m->dao; g.sa_rpu = R; g.sa_sstride = sstride; }
    WIS_RET(launch_ln_gemv(m, st, g));
    if (!g.sa_gran) WIS_RET(launch_dec_self_attn(st, m->dq, m->kc[l], m->vc[l], m->rm.pos, m->dao, M, H, d, ctx, R, sstride, rmul, pr ? pr + 16 : nullptr, 0, nullptr, 0, 0, nullptr, m->sa_nb));
    if (fold) {
      // ONE launch: x1 = x0 + Wo a + bo (tiles [0, d/16)) and q_raw = W'q x0 + (W'q Wo) a + W'q bo (the other d/16 tiles); the
      // cross-attention kernel applies the LayerNorm statistics of x1 (rs, mu) and b' to q_raw
      GemvP ga; memset(&ga, 0, sizeof(ga));
      ga.x = m->dao; ga.Wp = w.p_out; ga.bias = w.b_out; ga.y = m->dx; ga.M = M; ga.N = d; ga.K = d; ga.flags = GV_RESID; ga.prof = pr ? pr + 32 : nullptr;
      // (r5) the out-projection leaves LayerNorm partials of x1 (80 pairs per row); the cross-attention's prologue merges them instead of
      // every one of its 120 workgroups re-reading and re-summing the five 1280-float rows (WIS_B1_STAT_ROWS=1: the round-2..4 form, A/B)
      static const bool stat_rows = getenv("WIS_B1_STAT_ROWS") != nullptr;
      if (!stat_rows) ga.stat_out = m->dstat;
      GemvP gb; memset(&gb, 0, sizeof(gb));
      gb.x = m->dxh; gb.x2 = m->dao; gb.xsplit = d; gb.Wp = w.p_cqo; gb.bias = w.b_cqo; gb.y = m->dq; gb.M = M; gb.N = d; gb.K = 2 * d; gb.flags = GV_OUT_F32;
      static const bool ca_pf = getenv("WIS_CA_PREFETCH") && atoi(getenv("WIS_CA_PREFETCH")) != 0;      // (r6 A/B: L2 prefetch riders for the cross-attention's K / V)
      if (ca_pf && B == 1) WIS_RET(launch_gemv_dual(st, ga, gb, m->kx[l], m->vx[l], T, m->Tpad, chunks, H));
      else
      WIS_RET(launch_gemv_dual(st, ga, gb));
      if (!stat_rows) WIS_RET(launch_dec_cross_attn(st, m->dq, m->kx[l], m->vx[l], m->dao, m->part, m->counters, B, R, H, d, T, m->Tpad, chunks, pr ? pr + 64 : nullptr, 0,
                                                    m->dstat, w.c_cq, w.b_cq, m->spin_now ? m->ca_gran : nullptr, m->ca_epoch, nullptr, 1));
      else
      WIS_RET(launch_dec_cross_attn(st, m->dq, m->kx[l], m->vx[l], m->dao, m->part, m->counters, B, R, H, d, T, m->Tpad, chunks, pr ? pr + 64 : nullptr, 0,
                                    m->dx, w.c_cq, w.b_cq, m->spin_now ? m->ca_gran : nullptr, m->ca_epoch));
    } else {
    memset(&g, 0, sizeof(g));
    g.x = m->dao; g.Wp = w.p_out; g.wscale = w.s_out; g.bias = w.b_out; g.y = m->dx; g.M = M; g.N = d; g.K = d; g.flags = GV_RESID; g.prof = pr ? pr + 32 : nullptr;
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
    // cross-attention block
    memset(&g, 0, sizeof(g));
    g.x = m->dx; g.csum = w.c_cq; g.Wp = w.p_cq; g.wscale = w.s_cq; g.bias = w.b_cq; g.y = m->dq; g.M = M; g.N = d; g.K = d; g.flags = GV_LN | GV_OUT_F32; g.prof = pr ? pr + 48 : nullptr;
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
    WIS_RET(launch_dec_cross_attn(st, m->dq, m->kx[l], m->vx[l], m->dao, m->part, m->counters, B, R, H, d, T, m->Tpad, chunks, pr ? pr + 64 : nullptr, 0, nullptr, nullptr, nullptr,
                                  m->spin_now ? m->ca_gran : nullptr, m->ca_epoch));
    }
    memset(&g, 0, sizeof(g));
    g.x = m->dao; g.Wp = w.p_cout; g.wscale = w.s_cout; g.bias = w.b_cout; g.y = m->dx; g.M = M; g.N = d; g.K = d; g.flags = GV_RESID; g.prof = pr ? pr + 80 : nullptr;
    if (lnp) { g.y16 = m->dln; g.stat_out = m->dstat; }      // FFN1's input: f16 rows (dln is free at <= 8 rows) + partials
    if (ln16) g.y16 = m->dln;
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
    // FFN
    memset(&g, 0, sizeof(g));
    g.x = m->dx; g.csum = w.c_f1; g.Wp = w.p_f1; g.wscale = w.s_f1; g.bias = w.b_f1; g.y = m->dh; g.M = M; g.N = 4 * d; g.K = d; g.flags = GV_LN | GV_GELU; g.prof = pr ? pr + 96 : nullptr;
    if (lnp) { g.x = m->dln; g.stat_in = m->dstat; g.flags = GV_LNP | GV_GELU; }
    if (ln16) { g.x = m->dln; g.flags = GV_LN16 | GV_GELU; }
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
    memset(&g, 0, sizeof(g));
    g.x = m->dh; g.Wp = w.p_f2; g.wscale = w.s_f2; g.bias = w.b_f2; g.y = m->dx; g.M = M; g.N = d; g.K = 4 * d; g.flags = GV_RESID; g.prof = pr ? pr + 112 : nullptr;
    g.y16 = fold ? m->dxh : nullptr;             // the next layer's x0 in f16
    if (lnp) g.stat_out = m->dstat;              // ... and its LayerNorm partials (next layer's QKV / the vocabulary projection)
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
  }
  if (want_logits) {
    GemvP g; memset(&g, 0, sizeof(g));
    g.x = m->dx; g.csum = m->c_proj; g.bias = m->b_proj; g.Wp = m->p_proj; g.wscale = m->s_proj; g.y = m->logits; g.M = M; g.N = m->n_vocab_pad; g.K = d; g.flags = GV_LN | GV_OUT_F32;
    if (lnp) { g.x = m->dxh; g.stat_in = m->dstat; g.flags = GV_LNP | GV_OUT_F32; }
    if (ln16) { g.x = m->dxh; g.flags = GV_LN16 | GV_OUT_F32; }
    g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
    WIS_RET(launch_ln_gemv(m, st, g));
  }
  return WIS_OK;
}

// search state of a device batch: beams tiled up front with scores [0, -inf, ...] (CT2 GPU path), counters cleared
static int init_beam_state(wis_model* m, int B, int beam) {
  hipStream_t st = m->st;
  const int Mrows = B * beam;
  // staged in pinned memory of its own (h_pin + 2048 ..): the copies run when the stream gets there, nothing waits for them here
  float* cum = reinterpret_cast<float*>(m->h_pin + 2048);
  for (int r = 0; r < Mrows; ++r) cum[r] = (r % beam == 0) ? 0.f : -INFINITY;
  unsigned* tk = reinterpret_cast<unsigned*>(m->h_pin + 2048 + MAX_ROWS);
  m->gen = (m->gen % 0xFFFFu) + 1u;      // 1 .. 65535: never the 0 of a cleared record
  tk[0] = 0; tk[1] = m->gen; tk[2] = 0; tk[3] = 0;
  WIS_HIP_CHECK(hipMemcpyAsync(m->bs.cum, cum, (size_t)Mrows * 4, hipMemcpyHostToDevice, st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->bs.tick, tk, 16, hipMemcpyHostToDevice, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.step_u, 0, (size_t)B * 4, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.done, 0, (size_t)B * 4, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.n_hyp, 0, (size_t)B * 4, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.all_done, 0, 16, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->bs.out_len, 0, (size_t)B * 4, st));
  return WIS_OK;
}
// decoding options -> what the sampling kernels take (CTranslate2 4.1.0 BeamSearch defaults where WIS passes none, main.py:687-693)
static SampleCfg make_sample_cfg(const wis_model* m, const wis_gen_opts_t* o, int beam, int max_new, float* patience_out) {
  const wis_config_t& c = m->cfg;
  SampleCfg sc; memset(&sc, 0, sizeof(sc));
  sc.n_vocab = c.n_vocab; sc.n_vocab_pad = m->n_vocab_pad; sc.eot = c.eot; sc.beam = beam; sc.n_cand = 2 * beam; sc.max_new = max_new;
  sc.fixed_new = o->fixed_new_tokens; sc.suppress_blank = o->suppress_blank; sc.greedy = beam == 1;
  sc.length_penalty = o->length_penalty; sc.max_hyp = MAX_HYP;
  const float patience = o->patience > 0.f ? o->patience : 1.f;
  // hypotheses an utterance can hold: the search ends once max_candidates exist and one step adds at most `beam`, so
  // max_candidates + beam - 1 slots never overflow (MAX_HYP = 3 MAX_R: patience <= 2 at any beam; check_patience refuses more)
  sc.max_candidates = (int)lroundf((float)beam * patience); if (sc.max_candidates < 1) sc.max_candidates = 1;
  // CT2: allow_early_exit = patience == 1 && length_penalty == 0 && coverage_penalty == 0
  sc.allow_early_exit = (patience == 1.f && o->length_penalty == 0.f) ? 1 : 0;
  sc.early_exit_hyps = WIS_EARLY_EXIT_NUM_HYPOTHESES ? 1 : sc.max_candidates;      // num_hypotheses is 1 at this boundary (kernels.hpp)
  *patience_out = patience;
  return sc;
}

// A patience the hypothesis storage cannot honour is an argument error, not a silently shorter search (CTranslate2 would keep
// searching until round(beam x patience) hypotheses exist and return different ids)
static int check_patience(int beam, float patience) {
  const float p = patience > 0.f ? patience : 1.f;
  const long want = lroundf((float)beam * p);
  if (want > MAX_HYP - beam + 1) {
    set_error("patience %.3g at beam_size %d needs %ld finished hypotheses; the engine holds %d (patience <= %.3g at this beam)", (double)p, beam, want, MAX_HYP - beam + 1,
              (double)(MAX_HYP - beam + 1) / (double)beam);
    return WIS_E_ARG;
  }
  return WIS_OK;
}

int upload_rows(wis_model* m, const std::vector<int>& tok, const std::vector<int>& pos, const std::vector<int>& slot, const std::vector<int>& lslot, bool wait = true, int stage = 1024) {
  const size_t n = tok.size();
  int* h = m->h_pin + stage;      // (a caller that uploads twice without a wait in between passes a second staging area)
  memcpy(h, tok.data(), n * 4); memcpy(h + MAX_ROWS, pos.data(), n * 4); memcpy(h + 2 * MAX_ROWS, slot.data(), n * 4); memcpy(h + 3 * MAX_ROWS, lslot.data(), n * 4);
  WIS_HIP_CHECK(hipMemcpyAsync(m->rm.tok, h, n * 4, hipMemcpyHostToDevice, m->st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->rm.pos, h + MAX_ROWS, n * 4, hipMemcpyHostToDevice, m->st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->rm.slot, h + 2 * MAX_ROWS, n * 4, hipMemcpyHostToDevice, m->st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->rm.lslot, h + 3 * MAX_ROWS, n * 4, hipMemcpyHostToDevice, m->st));
  // the pinned staging area is reused by the next upload: make sure the copies are done (wis_generate uploads once per call and
  // drains its stream before it returns or re-enters, so it does not wait here)
  if (wait) WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  return WIS_OK;
}

int check_batch(wis_model* m, int B, int beam) {
  if (B < 1 || B > m->cfg.max_batch) { set_error("batch %d outside [1, max_batch=%d]", B, m->cfg.max_batch); return WIS_E_STATE; }
  if (beam < 1 || beam > m->cfg.max_beam || beam > MAX_R) { set_error("beam_size %d outside [1, %d]", beam, m->cfg.max_beam < MAX_R ? m->cfg.max_beam : MAX_R); return WIS_E_STATE; }
  if (B * beam > MAX_ROWS) { set_error("B*beam = %d exceeds %d decoder rows per device batch", B * beam, MAX_ROWS); return WIS_E_STATE; }
  return WIS_OK;
}

// A handle runs ONE compute call at a time (its activations, KV caches and stream are single-instance; the Python shim feeds every
// replica from one worker thread).  A second thread entering the same handle is refused with WIS_E_STATE instead of silently
// corrupting the first call's state - SURVEY 8(b) asks for thread safety at the boundary: concurrency comes from replicas and the
// micro-batcher, never from two calls inside one replica.
static std::atomic<int> g_active_calls[64];      // compute calls running per device (all handles): generate_impl's stream choice
struct BusyGuard {
  wis_model* m; bool ok;
  explicit BusyGuard(wis_model* mm) : m(mm), ok(!mm->busy.test_and_set(std::memory_order_acquire)) { if (ok) g_active_calls[m->device & 63].fetch_add(1, std::memory_order_relaxed); }
  ~BusyGuard() { if (ok) { g_active_calls[m->device & 63].fetch_sub(1, std::memory_order_relaxed); m->busy.clear(std::memory_order_release); } }
};
#define WIS_ENTER(m, what)                                                                                      \
  BusyGuard _busy(m);                                                                                           \
  if (!_busy.ok) { set_error(what ": another call is running on this handle (one call at a time per replica)"); return WIS_E_STATE; }

}  // namespace

// =======================================================================================
extern "C" {

int wis_version(void) { return WIS_ABI_VERSION; }
const char* wis_last_error(void) { return get_error(); }
int wis_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int wis_supported_compute_types(int device, char* out, size_t cap) {
  (void)device;
  const char* s = "float16,float32,int8_float16";
  if (!out || cap < strlen(s) + 1) { set_error("buffer too small"); return WIS_E_ARG; }
  strcpy(out, s); return WIS_OK;
}

int wis_model_create(const wis_config_t* cfg, const void* arena, size_t arena_bytes, int arena_on_device,
                     const wis_tensor_t* tensors, int n_tensors, int device, wis_model_t** out) {
  if (!cfg || !arena || !tensors || !out || n_tensors <= 0) { set_error("wis_model_create: bad argument"); return WIS_E_ARG; }
  if (cfg->d_model % 128 || cfg->d_model != cfg->n_heads * 64 || cfg->d_model > 2048 || cfg->n_mels != 80 || cfg->n_audio_ctx != 1500 ||
      cfg->n_text_ctx > 512 || cfg->max_batch < 1 || cfg->max_beam < 1 || cfg->max_beam > MAX_R || cfg->n_vocab < 1024) {
    set_error("wis_model_create: unsupported config (d_model %% 128, head_dim 64, n_mels 80, n_audio_ctx 1500, max_beam <= %d)", MAX_R);
    return WIS_E_UNSUPPORTED;
  }
  DeviceCtx* ctx; WIS_RET(get_ctx(device, &ctx));
  wis_model* m = new wis_model();
  m->cfg = *cfg; m->device = device; m->ctx = ctx;
  m->w8 = cfg->decoder_weight_bits == 8;
  m->cq_fold = !m->w8 && getenv("WIS_NO_CQFOLD") == nullptr;     // (8-bit weights keep the two-stage form: the fold would change what is quantised)
  m->use_graph = getenv("WIS_NO_GRAPH") == nullptr;
  memset(&m->timing, 0, sizeof(m->timing));
  int rc = WIS_OK;
  void* d_arena = nullptr;
  do {
    if (hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) != hipSuccess) { set_error("stream create failed"); rc = WIS_E_HIP; break; }
    const char* base = reinterpret_cast<const char*>(arena);
    if (!arena_on_device) {
      if (hipMalloc(&d_arena, arena_bytes) != hipSuccess) { set_error("hipMalloc(arena %zu) failed", arena_bytes); rc = WIS_E_NOMEM; break; }
      if (hipMemcpy(d_arena, arena, arena_bytes, hipMemcpyHostToDevice) != hipSuccess) { set_error("arena upload failed"); rc = WIS_E_HIP; break; }
      base = reinterpret_cast<const char*>(d_arena);
    }
    Loader L{tensors, n_tensors, base, arena_bytes};
    if ((rc = load_weights(m, L))) break;
    // logits processors as additive masks (0 / -inf) over the padded vocabulary
    {
      std::vector<float> ba(m->n_vocab_pad, 0.f), bb(m->n_vocab_pad, 0.f);
      for (int i = 0; i < cfg->n_suppress; ++i) if (cfg->suppress_ids[i] >= 0 && cfg->suppress_ids[i] < cfg->n_vocab) ba[cfg->suppress_ids[i]] = -INFINITY;
      for (int i = 0; i < cfg->n_suppress_begin; ++i) if (cfg->suppress_ids_begin[i] >= 0 && cfg->suppress_ids_begin[i] < cfg->n_vocab) bb[cfg->suppress_ids_begin[i]] = -INFINITY;
      if ((rc = dalloc(m, &m->bias_all, ba.size())) || (rc = dalloc(m, &m->bias_begin, bb.size()))) break;
      hipMemcpy(m->bias_all, ba.data(), ba.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(m->bias_begin, bb.data(), bb.size() * 4, hipMemcpyHostToDevice);
      if ((rc = dalloc(m, &m->d_lang_ids, (size_t)(cfg->n_lang > 0 ? cfg->n_lang : 1)))) break;
      if (cfg->n_lang > 0) hipMemcpy(m->d_lang_ids, cfg->lang_ids, (size_t)cfg->n_lang * 4, hipMemcpyHostToDevice);
    }
    m->cfg.suppress_ids = nullptr; m->cfg.suppress_ids_begin = nullptr; m->cfg.lang_ids = nullptr;   // caller-owned memory is not retained
    // everything carved so far is the (read-only) weight set: it moves into a shareable block; the buffers start on fresh slabs
    m->wslabs = std::make_shared<WeightSlabs>();
    m->wslabs->device = device; m->wslabs->slabs.swap(m->allocs); m->wslabs->bytes = m->bytes;
    m->slab_cur = nullptr; m->slab_left = 0;
    if ((rc = alloc_buffers(m))) break;
    if (hipStreamSynchronize(m->st) != hipSuccess) { set_error("model init failed"); rc = WIS_E_HIP; break; }
  } while (0);
  if (d_arena) hipFree(d_arena);
  if (rc) { wis_model_destroy(m); return rc; }
  *out = m;
  return WIS_OK;
}

void wis_model_destroy(wis_model_t* m) {
  if (!m) return;
  hipSetDevice(m->device);
  if (m->st) hipStreamSynchronize(m->st);
  spin_collect(m->device);      // (a deferred spin claim of this handle waits on one of its events: completed by the synchronise above)
  for (auto& kv : m->graphs) hipGraphExecDestroy(kv.second);
  for (void* p : m->allocs) hipFree(p);
  if (m->h_pin) hipHostFree(m->h_pin);
  if (m->h_prog) hipHostFree(m->h_prog);
  for (int i = 0; i < 8; ++i) if (m->ev[i]) hipEventDestroy(m->ev[i]);
  if (m->st_enc) { hipStreamSynchronize(m->st_enc); hipStreamDestroy(m->st_enc); }
  if (m->ev_enc) hipEventDestroy(m->ev_enc);
  if (m->ev_ckv) hipEventDestroy(m->ev_ckv);
  if (m->st) hipStreamDestroy(m->st);
  delete m;
}
size_t wis_model_device_bytes(const wis_model_t* m) { return m ? m->bytes : 0; }

int wis_model_clone(wis_model_t* parent, wis_model_t** out) {
  if (!parent || !out || !parent->wslabs) { set_error("wis_model_clone: bad argument"); return WIS_E_ARG; }
  WIS_HIP_CHECK(hipSetDevice(parent->device));
  wis_model* m = new wis_model();
  m->cfg = parent->cfg; m->device = parent->device; m->ctx = parent->ctx;
  m->wslabs = parent->wslabs;
  // the weight set (device pointers into the shared slabs)
  m->w_conv1 = parent->w_conv1; m->w_conv2 = parent->w_conv2; m->b_conv1 = parent->b_conv1; m->b_conv2 = parent->b_conv2;
  m->enc_pos = parent->enc_pos; m->enc_ln_g = parent->enc_ln_g; m->enc_ln_b = parent->enc_ln_b;
  m->enc = parent->enc; m->dec = parent->dec;
  m->emb = parent->emb; m->dec_pos = parent->dec_pos; m->p_proj = parent->p_proj; m->dec_ln_g = parent->dec_ln_g; m->dec_ln_b = parent->dec_ln_b;
  m->bias_all = parent->bias_all; m->bias_begin = parent->bias_begin; m->d_lang_ids = parent->d_lang_ids; m->n_vocab_pad = parent->n_vocab_pad;
  m->w_ckv_all = parent->w_ckv_all; m->b_ckv_all = parent->b_ckv_all;
  m->s_proj = parent->s_proj; m->c_proj = parent->c_proj; m->b_proj = parent->b_proj; m->w8 = parent->w8; m->cq_fold = parent->cq_fold;
  m->use_graph = parent->use_graph;
  memset(&m->timing, 0, sizeof(m->timing));
  int rc = WIS_OK;
  do {
    if (hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) != hipSuccess) { set_error("stream create failed"); rc = WIS_E_HIP; break; }
    if ((rc = alloc_buffers(m))) break;
    if (hipStreamSynchronize(m->st) != hipSuccess) { set_error("clone init failed"); rc = WIS_E_HIP; break; }
  } while (0);
  if (rc) { wis_model_destroy(m); return rc; }
  *out = m;
  return WIS_OK;
}

}  // extern "C" (reopened below: the generate driver is a static helper)

// The decoder rows of window steps s0 .. s0 + Rw - 1 of a beam trajectory (hd: [step][MAX_R][2] = token, origin; one utterance, k beams), step-major:
// row (s, j) feeds the token live beam j got at step s - 1 at position P - 1 + s and keeps its K / V in slot j; ha[row][ANC_W] = the slot of the
// row's ancestor at every window step (entry 0 doubles as the slot that holds everything before the window).  Padded to whole groups of 16 rows
// with copies of the last row (they write the same K / V to the same place).  Returns the padded row count.
static int fill_tree_window(const int* hd, int s0, int Rw, int k, int P, std::vector<int>& tok, std::vector<int>& pos, std::vector<int>& slot, std::vector<int>& ls, int* ha) {
  const int Mreal = k * Rw, Mpad = cdiv(Mreal, 16) * 16;
  tok.assign(Mpad, 0); pos.assign(Mpad, 0); slot.assign(Mpad, 0); ls.assign(Mpad, 0);
  for (int t = 0; t < Rw; ++t) for (int j = 0; j < k; ++j) {
    const int s_ = s0 + t, r = t * k + j;
    tok[r] = hd[((s_ - 1) * MAX_R + j) * 2]; pos[r] = P - 1 + s_; slot[r] = j; ls[r] = j;
    int a = j;                                    // ancestor of (s_, j) at window step sp, walking the origins back to s0
    for (int sp = s_; sp >= s0; --sp) { ha[r * ANC_W + (sp - s0)] = a; a = hd[((sp - 1) * MAX_R + a) * 2 + 1]; }
    for (int u = t + 1; u < ANC_W; ++u) ha[r * ANC_W + u] = ha[r * ANC_W + t];
  }
  for (int r = Mreal; r < Mpad; ++r) {
    tok[r] = tok[Mreal - 1]; pos[r] = pos[Mreal - 1]; slot[r] = slot[Mreal - 1]; ls[r] = ls[Mreal - 1];
    for (int u = 0; u < ANC_W; ++u) ha[r * ANC_W + u] = ha[(Mreal - 1) * ANC_W + u];
  }
  return Mpad;
}

// draft / n_draft / accepted: wis_generate_draft (one utterance, beam 1): the tokens of an earlier hypothesis to verify first
// draft_org (beam > 1, wis_generate_draft_beam): draft = [n_draft][beam] tokens, draft_org = [n_draft][beam] the beam slot each continued from -
// the trajectory of an earlier SEARCH (wis_last_trajectory); accepted = steps verified
static int generate_impl(wis_model_t* m, const float* input, int B, const int32_t* prompt, int P,
                 const wis_gen_opts_t* o, int32_t* out_ids, int32_t* out_len, float* out_score, bool* retry,
                 const int32_t* draft = nullptr, int n_draft = 0, int* accepted = nullptr, const int32_t* draft_org = nullptr) {
  *retry = false;
  if (accepted) *accepted = 0;
  WIS_HIP_CHECK(hipSetDevice(m->device));
  SpinClaim claim(m, B);
  const wis_config_t& c = m->cfg;
  const int beam = o->beam_size < 1 ? 1 : o->beam_size;
  WIS_RET(check_batch(m, B, beam));
  WIS_RET(check_patience(beam, o->patience));
  if (P < 1 || P > 16 || B * P > MAX_ROWS) { set_error("prompt length %d unsupported (1..16, B*P <= %d)", P, MAX_ROWS); return WIS_E_UNSUPPORTED; }
  int max_new = o->max_new_tokens > 0 ? o->max_new_tokens : std::min(c.n_text_ctx / 2, c.n_text_ctx - P);
  if (max_new > 256) max_new = 256;
  if (P - 1 + max_new > c.n_text_ctx) max_new = c.n_text_ctx - (P - 1);
  for (int i = 0; i < B * P; ++i) if (prompt[i] < 0 || prompt[i] >= c.n_vocab) { set_error("prompt token %d out of range", prompt[i]); return WIS_E_ARG; }
  hipStream_t st = m->st;
  const int ctx = c.n_text_ctx;
  auto t0 = std::chrono::steady_clock::now();

  // ---- decode state first: search counters and the prompt rows depend on nothing the encoder produces, and staged from pinned
  // memory they cost the host no wait - the whole chain log-mel -> encoder -> cross-K/V -> prefill is enqueued behind them in one go
  // (the prefill's ~230 launches are issued while the encoder runs instead of after two stream drains)
  const int Mrows = B * beam;
  WIS_RET(init_beam_state(m, B, beam));
  float patience;
  const SampleCfg sc = make_sample_cfg(m, o, beam, max_new, &patience);
  const float* bias_all = o->suppress_default ? m->bias_all : nullptr;
  const bool drafting = draft != nullptr && n_draft > 0;
  const bool beam_draft = drafting && draft_org != nullptr;
  if (drafting && B != 1) { set_error("wis_generate_draft: one utterance per call (got B = %d)", B); return WIS_E_UNSUPPORTED; }
  if (drafting && !beam_draft && beam != 1) { set_error("wis_generate_draft: beam_size 1 (a beam search is drafted by its trajectory: wis_generate_draft_beam)"); return WIS_E_UNSUPPORTED; }
  if (beam_draft && (beam < 2 || n_draft > 256)) { set_error("wis_generate_draft_beam: beam_size >= 2 and at most 256 draft steps (got %d, %d)", beam, n_draft); return WIS_E_ARG; }
  if (drafting) for (int i = 0; i < n_draft * (beam_draft ? beam : 1); ++i) if (draft[i] < 0 || draft[i] >= c.n_vocab) { set_error("draft token %d out of range", draft[i]); return WIS_E_ARG; }
  if (beam_draft) for (int i = 0; i < n_draft * beam; ++i) if (draft_org[i] < 0 || draft_org[i] >= beam) { set_error("draft origin %d outside [0, beam_size)", draft_org[i]); return WIS_E_ARG; }
  m->last_B = B; m->last_beam = beam;
  // Verification passes keep to the ticket hand-off of the cross-attention: their picks are accepted on the host pass by pass, outside the
  // progress record that carries the granule form's give-up flag (advisor, round 5) - the ordinary steps behind them take the call's form again
  const bool spin_call = m->spin_now;
  if (drafting) m->spin_now = false;
  if (!drafting || beam_draft) {
    std::vector<int> tok(B * P), pos(B * P), slot(B * P), ls(B * P);
    for (int b = 0; b < B; ++b) for (int i = 0; i < P; ++i) { tok[b * P + i] = prompt[b * P + i]; pos[b * P + i] = i; slot[b * P + i] = b * beam; ls[b * P + i] = b * beam; }
    WIS_RET(upload_rows(m, tok, pos, slot, ls, false));
  }
  const unsigned long long gen = m->gen;
  // a give-up flag raised by the PREVIOUS call's over-run step (it ran after that call had returned) says nothing about this call: cleared
  // behind that step, in stream order, before this call's first decoder pass can raise it again
  if (m->overrun_left) { WIS_HIP_CHECK(hipMemsetAsync(m->ca_epoch, 0, 4, st)); m->overrun_left = false; }
  __atomic_store_n(&m->h_prog[HP_REC], 0ull, __ATOMIC_RELAXED);      // (a record of an earlier call's over-run step may still land here: it carries that call's generation)
  m->h_prog[HP_DONE_STEP] = 0; m->h_prog[HP_DONE_STAMP] = 0; m->h_prog[HP_STAMP0] = 0;

  // Front half on a stream of its own.  A search that ended on EOT may have left ONE over-run decode step running on `st` (below): the
  // log-mel and the encoder of this call touch none of the decoder's buffers, so they start at once beside it instead of behind it;
  // the cross-K/V projection (which overwrites what that step still reads) and everything after it stay on `st`.  st_enc waits for the
  // previous call's cross-K/V projection - the last reader of the encoder's output buffer.
  // ... but only while this is the ONLY call running on the GPU.  HIP streams share a handful of hardware queues (four by default): with
  // several replicas decoding at once a second stream per handle puts one replica's encoder and another's decode chain into the same
  // queue, and the chain waits behind 100 us GEMMs - measured, 8 utterances per batch, 2 / 3 / 4 batches in flight: 130 / 151 / 145
  // utterances/s with the second stream against 165 / 178 / 165 without.  Under that load the over-run step costs next to nothing anyway
  // (the GPU is shared; the step is a thin chain).
  static const bool one_stream = getenv("WIS_ONE_STREAM") != nullptr;      // A/B switch
  static const bool two_streams = getenv("WIS_TWO_STREAMS") != nullptr;    // A/B switch: the second stream whatever else runs
  const bool alone = g_active_calls[m->device & 63].load(std::memory_order_relaxed) <= 1;
  hipStream_t se = (one_stream || !(alone || two_streams)) ? st : m->st_enc;
  if (se != st) WIS_HIP_CHECK(hipStreamWaitEvent(se, m->ev_ckv, 0));
  WIS_HIP_CHECK(hipEventRecord(m->ev[0], se));
  WIS_RET(stage_input(m, input, o->input_kind, B, se));
  WIS_HIP_CHECK(hipEventRecord(m->ev[1], se));
  WIS_RET(run_encoder(m, B, se));
  WIS_HIP_CHECK(hipEventRecord(m->ev[2], se));
  if (se != st) { WIS_HIP_CHECK(hipEventRecord(m->ev_enc, se)); WIS_HIP_CHECK(hipStreamWaitEvent(st, m->ev_enc, 0)); }
  WIS_RET(run_cross_kv(m, B));
  WIS_HIP_CHECK(hipEventRecord(m->ev[3], st));
  WIS_HIP_CHECK(hipEventRecord(m->ev_ckv, st));

  // ---- prefill + FIRST decode step in one pass: all P prompt tokens of an utterance are rows (b, i) at positions i in the
  // utterance's first KV slot (causal by position); the logits of the last prompt row seed the beams (CT2 forwards
  // prompt[:-1] and then feeds prompt[-1] as the first decoder input — the same arithmetic, one weight pass instead of two)
  int steps = 1;            // decoder passes done: the first step runs with the prefill pass
  bool spec_done = false;   // the draft verification already met the end of the utterance
  bool beam_fin = false;    // ... of a beam search: the replayed beam steps finished it, results are where beam_step_kernel puts them
  std::vector<int> spec_gen; float spec_cum = 0.f; int spec_len = 0;
  if (!drafting) {
    WIS_RET(dec_forward(m, B * P, P, B, true, beam, 0));
    WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->bs.step_u, m->st_max, m->st_sum, m->st_val, m->st_idx, B, sc, P, 0, P - 1, WIS_TAPS ? m->d_prof + (size_t)m->cfg.n_dec_layers * 8 * 16 + 16 : nullptr));
    WIS_RET(launch_beam_step(st, m->st_max, m->st_sum, m->st_val, m->st_idx, m->bs, m->rm, B, P, ctx, sc, WIS_TAPS ? m->d_prof + (size_t)m->cfg.n_dec_layers * 8 * 16 : nullptr));
    WIS_RET(launch_kv_reorder(st, m->kc_all, m->vc_all, m->kv_layer_stride, c.n_dec_layers, m->bs, B, beam, P, ctx, c.d_model));
  } else if (beam_draft) {
    // ---- verify the draft of a BEAM SEARCH (round 6; BASELINE configs[4] at the reference's long-audio beam, main.py:582-586).  The draft is
    // the trajectory of an earlier search over (most of) the same audio: per step s the live set it left - k tokens and the beam slot each
    // continued from.  If the search over THIS window has followed it up to step s0 - 1, the decoder rows of steps s0 .. s0 + Rw - 1 are
    // known without running those steps: row (s, j) feeds the draft's token of live beam j after step s - 1 at position P - 1 + s.  They form a
    // TREE (a beam's history is a path through earlier live sets), so the pass runs the self-attention by ancestor table (dec_self_attn_kernel
    // TREE: node (s, j) keeps its K / V in slot j, row (s, j) reads position P - 1 + s' from the slot of its ancestor at step s') and the
    // cross-attention as groups of 16 rows over the utterance's one K / V.  One weight stream then yields the logits of Rw steps x k beams;
    // the steps are REPLAYED on them by the ordinary sampling kernels (logit_stats, beam_step: search state and hypothesis list end up exactly
    // where Rw ordinary steps would leave them), each followed by draft_match_kernel: if the live set a replayed step produced is the draft's -
    // as a SET: near-tied candidates swap slots between two searches all the time, so live beam j may be any draft node as long as every beam is
    // found once; the next step reads beam j's logits from the row of its node - the next step's rows were the right ones; the first step with a
    // beam the draft does not have still stands (its own inputs were verified), parks the search (done = 2), and ordinary steps resume behind
    // it.  The cache: the pass left node (s, i)'s K / V in slot i of the draft's numbering; the matching kernel keeps every live beam's path
    // through those slots and kv_gather_kernel turns the paths into "slot j = beam j's history" once per window.  A whole window is queued
    // without a host round trip; the host looks once per window.  Exact by construction: every accepted step ran beam_step_kernel on the logits
    // of its true inputs (summed in the multi-row order, as any other batch shape of the engine).
    const int k = beam;
    const int nd = std::min(n_draft, max_new - 1);
    int* hd = m->h_pin + 8192;                         // the draft in BeamState::traj's layout ([step][MAX_R][2])
    for (int s_ = 0; s_ < nd; ++s_) for (int j = 0; j < k; ++j) { hd[(s_ * MAX_R + j) * 2] = draft[s_ * k + j]; hd[(s_ * MAX_R + j) * 2 + 1] = draft_org[s_ * k + j]; }
    int* hv = m->h_pin + 12288;                        // staging of the verification state (dec_kernels.hip draft_match_kernel: vs), read back per window
    for (int i = 0; i < 32; ++i) hv[i] = 0;
    for (int j = 0; j < MAX_R; ++j) { hv[DRAFT_VS_PERM + j] = j; hv[DRAFT_VS_BASE + j] = j; }
    if (nd > 0) WIS_HIP_CHECK(hipMemcpyAsync(m->d_draft, hd, (size_t)nd * MAX_R * 2 * 4, hipMemcpyHostToDevice, st));
    WIS_HIP_CHECK(hipMemcpyAsync(m->d_vstate, hv, 32 * 4, hipMemcpyHostToDevice, st));
    // merged prefill + first step as in the ordinary call, then: is the search where the draft's step 0 says?
    WIS_RET(dec_forward(m, P, P, 1, true, beam, 0));
    WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->bs.step_u, m->st_max, m->st_sum, m->st_val, m->st_idx, 1, sc, P, 0, P - 1));
    WIS_RET(launch_beam_step(st, m->st_max, m->st_sum, m->st_val, m->st_idx, m->bs, m->rm, 1, P, ctx, sc));
    WIS_RET(launch_kv_reorder(st, m->kc_all, m->vc_all, m->kv_layer_stride, c.n_dec_layers, m->bs, 1, beam, P, ctx, c.d_model));
    WIS_RET(launch_draft_match(st, m->bs, m->d_draft, nd, k, m->d_vstate, 0));
    const int RW = std::min(ANC_W, MAX_ROWS / k);       // steps per window: k x RW rows (beam 2 / 3: 32 steps, 5: 19, 8: 12), padded to whole groups of 16
    const int s_last = std::min(nd, max_new - 1);       // last step a window can hold: rows from the draft's entry s - 1; step max_new - 1 ends every search
    int* ha = m->h_pin + 12352;                         // ancestor table of the window rows, [rows][ANC_W] (.. 15424; the rows' staging follows at 15488)
    int* hb = m->h_pin + 15872;                         // ... and the slot holding each row's history before the window, [rows] (.. 15968)
    int* hw2 = m->h_pin + 16000;                        // per-window reset of the path bookkeeping: vs[2] = 0, vs[16 + j] = j
    int done_flag = 0, step_dev = 0;
    int pinv[MAX_R];                                    // real slot of draft node i at the window's start (the matching read back at the previous sync)
    for (int j = 0; j < MAX_R; ++j) pinv[j] = j;        // (first window: every slot holds the same prompt rows - any assignment is right)
    for (int s0 = 1;; ) {
      const int Rw = std::min(RW, s_last - s0 + 1);
      if (Rw >= 1) {
        std::vector<int> tok, pos, slot, ls;
        const int Mpad = fill_tree_window(hd, s0, Rw, k, P, tok, pos, slot, ls, ha);
        for (int r = 0; r < Mpad; ++r) hb[r] = pinv[ha[r * ANC_W]];      // the row's window-step-0 ancestor is draft node ha[r][0]: its earlier history sits in that node's REAL slot
        // the window's rows go through a row table of their own: the search's table (next input rows, written by the last beam step that
        // counted) must survive a window that turns out to sit behind a parked search (queued before the host has looked)
        const RowMeta rm_search = m->rm;
        m->rm = m->rm_win;
        int rc = upload_rows(m, tok, pos, slot, ls, false, 15488);      // (the prompt rows' staging copy may still be pending: own area; windows are a sync apart)
        if (!rc && (hipMemcpyAsync(m->d_anc, ha, (size_t)Mpad * ANC_W * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipMemcpyAsync(m->d_base, hb, (size_t)Mpad * 4, hipMemcpyHostToDevice, st) != hipSuccess)) { set_error("draft window: ancestor table upload failed"); rc = WIS_E_HIP; }
        TreeWin tw{m->d_anc, P - 1 + s0, ANC_W}; tw.base = m->d_base;
        if (!rc) rc = dec_forward(m, Mpad, 16, Mpad / 16, true, 1, 0, &tw);
        m->rm = rm_search;
        WIS_RET(rc);
        hw2[0] = 0; for (int j = 0; j < MAX_R; ++j) hw2[1 + j] = j;
        WIS_HIP_CHECK(hipMemcpyAsync(m->d_vstate + 2, hw2, 4, hipMemcpyHostToDevice, st));
        WIS_HIP_CHECK(hipMemcpyAsync(m->d_vstate + DRAFT_VS_BASE, hw2 + 1, MAX_R * 4, hipMemcpyHostToDevice, st));
        for (int t = 0; t < Rw; ++t) {      // replay: beam j's logits come from the row of the draft node it is matched to (rowmap); no cache traffic per step
          WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->bs.step_u, m->st_max, m->st_sum, m->st_val, m->st_idx, 1, sc, beam, 1, t * k, nullptr, m->d_vstate + DRAFT_VS_PERM));
          WIS_RET(launch_beam_step(st, m->st_max, m->st_sum, m->st_val, m->st_idx, m->bs, m->rm, 1, P, ctx, sc));
          WIS_RET(launch_draft_match(st, m->bs, m->d_draft, nd, k, m->d_vstate, 1));
        }
        // the window's paths applied to the cache at once: slot j = live beam j's history, as ordinary steps (and the next window) expect it
        WIS_RET(launch_kv_gather(st, m->kc_all, m->vc_all, m->kv_layer_stride, c.n_dec_layers, m->d_vstate, m->bs.done, beam, P - 1 + s0, ctx, c.d_model));
      }
      WIS_HIP_CHECK(hipMemcpyAsync(hv, m->d_vstate, 16 * 4, hipMemcpyDeviceToHost, st));      // steps verified, ..., the matching
      WIS_HIP_CHECK(hipMemcpyAsync(hv + 24, m->bs.done, 4, hipMemcpyDeviceToHost, st));
      WIS_HIP_CHECK(hipMemcpyAsync(hv + 25, m->bs.step_u, 4, hipMemcpyDeviceToHost, st));
      WIS_HIP_CHECK(hipStreamSynchronize(st));
      done_flag = hv[24]; step_dev = hv[25];
      if (Rw < 1 || done_flag != 0) break;
      for (int j = 0; j < k; ++j) { const int n_ = hv[DRAFT_VS_PERM + j]; if (n_ >= 0 && n_ < k) pinv[n_] = j; }
      s0 += Rw;
    }
    if (accepted) *accepted = hv[0];
    __atomic_store_n(&m->h_prog[HP_REC], 0ull, __ATOMIC_RELAXED);      // the replayed steps' progress records count launches, not steps: the pacing loop starts from `steps`
    if (done_flag == 1) {      // the replayed steps ended the search: hypotheses ranked, result written by beam_step_kernel
      beam_fin = true; steps = step_dev + 1;
    } else {                   // parked behind a step that stands (or nothing to verify): ordinary steps resume
      steps = step_dev;
      if (steps < 1) { set_error("wis_generate_draft_beam: verification completed no step"); return WIS_E_STATE; }
      int* hw = m->h_pin + 4096 + 512;
      unsigned* tk = reinterpret_cast<unsigned*>(hw);
      tk[0] = (unsigned)steps; tk[1] = m->gen; tk[2] = 0; tk[3] = 0;
      WIS_HIP_CHECK(hipMemcpyAsync(m->bs.tick, tk, 16, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemsetAsync(m->bs.done, 0, 4, st));
    }
  } else {
    // ---- verify the draft: the prompt and the draft tokens go through the decoder as teacher-forced rows, 16 positions per pass
    // (causal by position inside the utterance's KV slot, like the merged prompt pass); row i's logits are what a greedy step fed
    // seq[i] after seq[0..i-1] sees, so as long as every earlier draft token equalled the greedy pick, row P-1+g yields generated
    // token g.  The first disagreement ends the verification WITH the right token for that index (its prefix was right); the K / V
    // rows of the accepted prefix are in the cache, and the ordinary step loop continues from there.  Per pass one weight stream
    // for up to 16 tokens instead of one per token.
    const int nd = std::min(n_draft, max_new - 1);
    std::vector<int> seq(P + nd);
    for (int i = 0; i < P; ++i) seq[i] = prompt[i];
    for (int i = 0; i < nd; ++i) seq[P + i] = draft[i];
    // (r6) up to 96 positions per pass: more than 16 rows of one utterance go through the row-group form of the tree pass (a chain is a tree whose
    // every ancestor sits in slot 0: dec_self_attn_kernel<TREE> with an all-zero table is "causal by position in the slot", the cross-attention
    // takes the rows as groups of 16 over the one K / V) - the 100 rows of a 96-token draft are 2 passes (3.5 + 1.3 ms) instead of 7 x 1.35 ms.
    // WIS_DRAFT_ROWS=16: the round-5 schedule (A/B)
    static const int env_rows = getenv("WIS_DRAFT_ROWS") ? atoi(getenv("WIS_DRAFT_ROWS")) : 0;
    const int R = (env_rows >= 16 && env_rows <= MAX_ROWS) ? env_rows / 16 * 16 : MAX_ROWS;
    bool stop = false; int n_acc = 0;
    for (int t0 = 0; t0 < P + nd && !stop; t0 += R) {
      const int rows = std::min(R, P + nd - t0);
      const int f = std::max(P - 1 - t0, 0), nv = rows - f;      // rows f .. rows-1 of this pass predict generated tokens
      const int Mp = rows > 16 ? cdiv(rows, 16) * 16 : rows;      // (row groups: padded with copies of the last row - same K / V to the same place)
      std::vector<int> tok(Mp), pos(Mp), slot(Mp, 0), ls(Mp, 0);
      for (int i = 0; i < Mp; ++i) { const int ii = std::min(i, rows - 1); tok[i] = seq[t0 + ii]; pos[i] = t0 + ii; }
      WIS_RET(upload_rows(m, tok, pos, slot, ls));
      if (rows > 16) {
        WIS_HIP_CHECK(hipMemsetAsync(m->d_anc, 0, (size_t)Mp * 16 * 4, st));
        const TreeWin tw{m->d_anc, t0, 16};
        WIS_RET(dec_forward(m, Mp, 16, Mp / 16, nv > 0, 1, 0, &tw));
      } else
      WIS_RET(dec_forward(m, rows, rows, 1, nv > 0, 1, 0));
      if (nv <= 0) continue;
      int* hv = m->h_pin + 4096;
      for (int i = 0; i < nv; ++i) hv[i] = t0 + f + i - (P - 1);      // the step index of each verified row (first-step / EOT masks of logit_stats_kernel)
      WIS_HIP_CHECK(hipMemcpyAsync(m->vstep, hv, (size_t)nv * 4, hipMemcpyHostToDevice, st));
      WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->vstep, m->st_max, m->st_sum, m->st_val, m->st_idx, nv, sc, 1, 0, f));
      WIS_RET(launch_greedy_pick(st, m->st_max, m->st_sum, m->st_val, m->st_idx, nv, sc, m->pick_tok, m->pick_lp));
      int* ht = m->h_pin + 4096 + MAX_ROWS; float* hl = reinterpret_cast<float*>(m->h_pin + 4096 + 2 * MAX_ROWS);
      WIS_HIP_CHECK(hipMemcpyAsync(ht, m->pick_tok, (size_t)nv * 4, hipMemcpyDeviceToHost, st));
      WIS_HIP_CHECK(hipMemcpyAsync(hl, m->pick_lp, (size_t)nv * 4, hipMemcpyDeviceToHost, st));
      WIS_HIP_CHECK(hipStreamSynchronize(st));
      for (int i = 0; i < nv && !stop; ++i) {
        const int g = hv[i];
        spec_cum = hl[i] + spec_cum;                       // beam_step_kernel: (logit - lse) + cum
        spec_gen.push_back(ht[i]);
        const bool eos = ht[i] == c.eot, is_last = g + 1 >= max_new;
        if (eos || is_last) { spec_done = true; spec_len = eos ? g : g + 1; stop = true; }
        else if (g < nd && ht[i] == draft[g]) ++n_acc;
        else stop = true;                                  // first disagreement (or the row behind the last draft token): ht[i] is generated token g
      }
    }
    if (accepted) *accepted = n_acc;
    steps = (int)spec_gen.size();
    if (steps < 1) { set_error("wis_generate_draft: verification produced no token"); return WIS_E_STATE; }
    if (!spec_done) {
      // the search state a run of `steps` ordinary steps would have left: history, cumulative score, next input row, counters
      int* hs = m->h_pin + 4096;
      for (int t = 0; t < steps; ++t) hs[t] = spec_gen[t];
      WIS_HIP_CHECK(hipMemcpyAsync(m->bs.alive, hs, (size_t)steps * 4, hipMemcpyHostToDevice, st));
      int* hw = m->h_pin + 4096 + 512;
      hw[0] = steps; hw[1] = spec_gen.back(); hw[2] = P - 1 + steps; hw[3] = 0;
      memcpy(hw + 4, &spec_cum, 4);
      WIS_HIP_CHECK(hipMemcpyAsync(m->bs.step_u, hw, 4, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemcpyAsync(m->rm.tok, hw + 1, 4, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemcpyAsync(m->rm.pos, hw + 2, 4, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemcpyAsync(m->rm.slot, hw + 3, 4, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemcpyAsync(m->rm.lslot, hw + 3, 4, hipMemcpyHostToDevice, st));
      WIS_HIP_CHECK(hipMemcpyAsync(m->bs.cum, hw + 4, 4, hipMemcpyHostToDevice, st));
      unsigned* tk = reinterpret_cast<unsigned*>(hw + 8);
      tk[0] = (unsigned)steps; tk[1] = m->gen; tk[2] = 0; tk[3] = 0;      // the progress record counts passes: `steps` of them are done
      WIS_HIP_CHECK(hipMemcpyAsync(m->bs.tick, tk, 16, hipMemcpyHostToDevice, st));
    }
  }
  m->spin_now = spin_call;
  WIS_HIP_CHECK(hipEventRecord(m->ev[4], st));
  if (beam_fin) WIS_HIP_CHECK(hipEventRecord(m->ev[5], st));

  auto one_step = [&]() -> int {
    WIS_RET(dec_forward(m, Mrows, beam, B, true, beam, 1));
    WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->bs.step_u, m->st_max, m->st_sum, m->st_val, m->st_idx, B, sc, beam, 1, 0, WIS_TAPS ? m->d_prof + (size_t)m->cfg.n_dec_layers * 8 * 16 + 16 : nullptr));
    WIS_RET(launch_beam_step(st, m->st_max, m->st_sum, m->st_val, m->st_idx, m->bs, m->rm, B, P, ctx, sc, WIS_TAPS ? m->d_prof + (size_t)m->cfg.n_dec_layers * 8 * 16 : nullptr));
    WIS_RET(launch_kv_reorder(st, m->kc_all, m->vc_all, m->kv_layer_stride, c.n_dec_layers, m->bs, B, beam, P, ctx, c.d_model));
    return WIS_OK;
  };
  // The step graph comes in up to three forms that differ in ONE kernel argument set: how many 8-position blocks of its rows' K / V history the self-attention
  // asks for (dec_self_attn_kernel NB).  Every row of the pass that follows s beam steps has P + s positions - known HERE, by the step index, although the graph's
  // kernel arguments are frozen - so the pass is launched from the graph whose self-attention asks for 16 / 32 / 64 positions (WIS_SA_NB=0: always 64, A/B switch).
  static const bool sa_short = !(getenv("WIS_SA_NB") && atoi(getenv("WIS_SA_NB")) == 0);
  auto nb_for = [&](int passes_done) { const int len = P + passes_done; return !sa_short ? 8 : (len <= 16 ? 2 : (len <= 32 ? 4 : 8)); };
  struct NbReset { wis_model* m; ~NbReset() { m->sa_nb = 8; } } nb_reset{m};      // (everything outside this loop - prefill, verification windows, taps - asks for 64)
  auto graph_for = [&](int nb, hipGraphExec_t* out) -> int {
    GraphKey key; memset(&key, 0, sizeof(key));
    key.B = B; key.beam = beam; key.P = P; key.max_new = max_new; key.fixed_new = sc.fixed_new; key.suppress_blank = sc.suppress_blank;
    key.suppress_default = o->suppress_default; key.early_exit = sc.allow_early_exit; key.lp = sc.length_penalty; key.patience = patience; key.spin = (m->spin_now ? 1 : 0) | (sa_fuse_enabled() ? 2 : 0);
    key.sa_nb = nb;
    auto it = m->graphs.find(key);
    if (it != m->graphs.end()) { *out = it->second; return WIS_OK; }
    hipGraph_t graph = nullptr; hipGraphExec_t ge = nullptr;
    WIS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = one_step();
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { set_error("graph capture failed: %s", hipGetErrorString(e)); return WIS_E_HIP; }
    e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { set_error("graph instantiate failed: %s", hipGetErrorString(e)); return WIS_E_HIP; }
    m->graphs[key] = ge;
    *out = ge;
    return WIS_OK;
  };
  auto launch_pass = [&](int passes_done) -> int {      // the decoder pass + sampling that follows `passes_done` passes
    m->sa_nb = nb_for(passes_done);
    if (!m->use_graph) return one_step();
    hipGraphExec_t ge = nullptr;
    WIS_RET(graph_for(m->sa_nb, &ge));
    WIS_HIP_CHECK(hipGraphLaunch(ge, st));
    return WIS_OK;
  };
  int needed = 0;           // steps after which the last utterance had finished (natural termination)
  bool gave_up = false, from_host = false;
  float decode_ms_dev = -1.f;
  // with the measurement convention the step count is known: fixed_new tokens + the forced EOT
  const int known = (sc.fixed_new > 0) ? std::min(max_new, sc.fixed_new + 1) : 0;
  const int limit = known ? known : max_new;
  auto unpack = [&](unsigned long long r, int* st_, int* dn_, int* gu_) {
    if ((r >> 48) != gen) { *st_ = 0; *dn_ = 0; *gu_ = 0; return; }
    *st_ = (int)((r >> 32) & 0xFFFFu); *gu_ = (int)((r >> 16) & 0xFFFFu); *dn_ = (int)(r & 0xFFFFu);
  };
  const int base_done = drafting ? steps : 0;      // passes the draft verification stands for: done before the first progress record of this call
  const auto t_dec0 = std::chrono::steady_clock::now();
  if (spec_done || beam_fin) {
    needed = steps;
  } else if (known) {
    // every step goes out in one burst: nothing to find out from the device before the last one
    for (; steps < limit; ++steps) WIS_RET(launch_pass(steps));
    WIS_HIP_CHECK(hipEventRecord(m->ev[5], st));
    WIS_HIP_CHECK(hipStreamSynchronize(st));
    int st_, dn_, gu_; unpack(__atomic_load_n(&m->h_prog[HP_REC], __ATOMIC_ACQUIRE), &st_, &dn_, &gu_);
    gave_up = gu_ != 0;
    if (!gave_up && dn_ < B) { set_error("decode did not terminate within %d steps (done %d of %d)", steps, dn_, B); return WIS_E_STATE; }
    needed = steps;
  } else {
    // A search that ends on EOT (every real request: the reference passes no max_length, main.py:687-693).  The host keeps `depth`
    // steps enqueued beyond the last one it has seen complete (one running, one waiting behind it: the device never idles between
    // steps) and reads the search's progress from the host-mapped record beam_step_kernel writes at the end of every step - no
    // stream drain, no copy.  When the record says every utterance has finished, at most depth - 1 further steps are in the queue;
    // they run on finished utterances (every sampling workgroup returns at its `done` test, results stay as they are) and this
    // call does not wait for them: the results are already in host memory, the next call on the handle queues behind them.
    const int depth = o->queue_depth > 0 ? o->queue_depth : 2;
    auto t_last = std::chrono::steady_clock::now();
    int seen = -1, st_ = 0, dn_ = 0, gu_ = 0;
    static const bool trace = getenv("WIS_EOT_TRACE") != nullptr;      // per-step record of the pacing loop on stderr (tuning)
    struct Tr { int step; unsigned long long dev; double host_us; int launched; };
    std::vector<Tr> tr;
    const auto t_loop = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
      unpack(__atomic_load_n(&m->h_prog[HP_REC], __ATOMIC_ACQUIRE), &st_, &dn_, &gu_);
      if (st_ < base_done) st_ = base_done;
      if (dn_ >= B || gu_) break;
      if (st_ >= limit) break;                 // (cannot happen: the max_new-th step finishes every utterance)
      if (steps < limit && steps - st_ < depth) {
        const auto tl0 = std::chrono::steady_clock::now();
        WIS_RET(launch_pass(steps));
        ++steps;
        if (trace) tr.push_back({-steps, 0ull, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl0).count(), steps});      // (negative step: a launch, host_us = its duration)
        continue;
      }
      if (st_ != seen) {
        seen = st_; t_last = std::chrono::steady_clock::now(); spins = 0;
        if (trace) tr.push_back({st_, m->h_prog[HP_STAMP], std::chrono::duration<double, std::micro>(t_last - t_loop).count(), steps});
      }
      else if ((spins & 1023u) == 1023u) {
        const double idle = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count();
        if (idle > 30.0) { set_error("decode made no progress for 30 s (step %d of %d enqueued)", st_, steps); return WIS_E_HIP; }
        if (idle > 200e-6) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }      // long steps (big batches): stop burning the core
      }
      cpu_relax();
    }
    if (trace) {
      unsigned long long prev_dev = 0;
      for (size_t i = 0; i < tr.size(); ++i) {
        if (tr[i].step < 0) { fprintf(stderr, "[eot-trace]   launch of step %d took the host %.1f us\n", -tr[i].step, tr[i].host_us); continue; }
        fprintf(stderr, "[eot-trace] step %d seen by the host at %.1f us (device clock +%.1f us since the previous record), %d steps enqueued\n", tr[i].step, tr[i].host_us,
                prev_dev ? (double)(tr[i].dev - prev_dev) * 0.01 : 0.0, tr[i].launched);
        prev_dev = tr[i].dev;
      }
      fprintf(stderr, "[eot-trace] done seen at %.1f us: %d of %d utterances, %d steps enqueued\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop).count(), dn_, B, steps);
    }
    gave_up = gu_ != 0;
    if (!gave_up && dn_ < B) { set_error("decode did not terminate within %d steps (done %d of %d)", steps, dn_, B); return WIS_E_STATE; }
    if (!gave_up) {
      from_host = true;
      needed = (int)m->h_prog[HP_DONE_STEP];
      const unsigned long long s0 = m->h_prog[HP_STAMP0], s1 = m->h_prog[HP_DONE_STAMP];
      decode_ms_dev = s1 > s0 ? (float)((double)(s1 - s0) * 1e-5) : 0.f;      // 100 MHz constant clock; from the end of the first beam step (ev[4] is one kv_reorder later)
      if (drafting) decode_ms_dev = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_dec0).count();      // (no first beam step of its own: host clock)
    }
  }
  // the granule hand-off's give-up flag (a combiner's bounded spin ran out: another handle's chain held the CUs its producers needed).
  // Not an error for the caller: the handle keeps to the ticket hand-off from now on and the call is run again (wis_generate).
  if (gave_up) {
    WIS_HIP_CHECK(hipMemsetAsync(m->ca_epoch, 0, 4, st));
    WIS_HIP_CHECK(hipStreamSynchronize(st));
    m->spin_off = true; *retry = true; ++m->handoff_retries;
    fprintf(stderr, "[wis_hip] device %d: decoder cross-attention granule hand-off timed out; this handle uses the ticket hand-off from now on, the call is repeated\n", m->device);
    return WIS_OK;
  }
  // results: out_ids is [B][max_new] (beam_step_kernel indexes by the resolved max_new; the allocations are [.][256])
  if (spec_done) {      // the verification rows already contained the end of the utterance: beam_step_kernel's finalisation, on the host
    float sfin = spec_cum;
    if (sc.length_penalty != 0.f) sfin /= powf((float)spec_len, sc.length_penalty);
    out_len[0] = spec_len;
    for (int t = 0; t < max_new; ++t) out_ids[t] = t < spec_len ? spec_gen[t] : 0;
    if (out_score) out_score[0] = sfin;
    from_host = true; decode_ms_dev = 0.f;
  } else if (from_host) {
    const int* hl = hp_out_len(m->h_prog); const float* hs = hp_out_score(m->h_prog); const int* hi = hp_out_ids(m->h_prog);
    for (int b = 0; b < B; ++b) {
      int n = hl[b];
      if (n < 0 || n > max_new) { set_error("decode result of utterance %d has length %d outside [0, %d]", b, n, max_new); return WIS_E_STATE; }
      out_len[b] = n;
      for (int t = 0; t < max_new; ++t) out_ids[(size_t)b * max_new + t] = t < n ? hi[(size_t)b * max_new + t] : 0;
      if (out_score) out_score[b] = hs[b];
    }
  } else {
    std::vector<int32_t> ids((size_t)B * 256);
    WIS_HIP_CHECK(hipMemcpyAsync(ids.data(), m->bs.out_ids, (size_t)B * 256 * 4, hipMemcpyDeviceToHost, st));
    WIS_HIP_CHECK(hipMemcpyAsync(out_len, m->bs.out_len, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    std::vector<float> sc_h(B);
    WIS_HIP_CHECK(hipMemcpyAsync(sc_h.data(), m->bs.out_score, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    WIS_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      if (out_len[b] > max_new) out_len[b] = max_new;
      for (int t = 0; t < max_new; ++t) out_ids[(size_t)b * max_new + t] = t < out_len[b] ? ids[(size_t)b * max_new + t] : 0;
      if (out_score) out_score[b] = sc_h[b];
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  float ms;
  hipEventElapsedTime(&ms, m->ev[0], m->ev[1]); m->timing.logmel_ms = ms;
  hipEventElapsedTime(&ms, m->ev[1], m->ev[2]); m->timing.encoder_ms = ms;
  hipEventElapsedTime(&ms, m->ev[2], m->ev[3]); m->timing.crosskv_ms = ms;
  hipEventElapsedTime(&ms, m->ev[3], m->ev[4]); m->timing.prefill_ms = ms;
  if (from_host) m->timing.decode_ms = decode_ms_dev; else { hipEventElapsedTime(&ms, m->ev[4], m->ev[5]); m->timing.decode_ms = ms; }
  m->timing.total_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
  m->timing.decode_steps = steps;             // steps enqueued (the merged prefill + first step included)
  m->timing.decode_steps_needed = needed;     // steps after which every utterance had finished: steps - needed = over-run
  if (steps > needed) { claim.defer(st); m->overrun_left = true; }      // an over-run step is still queued: its combiners keep their share of the spin budget until it has run
  return WIS_OK;
}

extern "C" {

int wis_generate(wis_model_t* m, const float* input, int B, const int32_t* prompt, int P,
                 const wis_gen_opts_t* o, int32_t* out_ids, int32_t* out_len, float* out_score) {
  if (!m || !input || !prompt || !o || !out_ids || !out_len) { set_error("wis_generate: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_generate")
  bool retry = false;
  WIS_RET(generate_impl(m, input, B, prompt, P, o, out_ids, out_len, out_score, &retry));
  if (retry) {      // once: the second run takes the ticket hand-off (spin_off is set), which cannot time out
    WIS_RET(generate_impl(m, input, B, prompt, P, o, out_ids, out_len, out_score, &retry));
    if (retry) { set_error("wis_generate: hand-off flag raised without the granule path"); return WIS_E_STATE; }
  }
  return WIS_OK;
}

int wis_generate_draft(wis_model_t* m, const float* input, const int32_t* prompt, int P, const wis_gen_opts_t* o,
                       const int32_t* draft, int n_draft, int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* accepted) {
  if (!m || !input || !prompt || !o || !out_ids || !out_len || (n_draft > 0 && !draft) || n_draft < 0) { set_error("wis_generate_draft: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_generate_draft")
  bool retry = false;
  int acc = 0;
  WIS_RET(generate_impl(m, input, 1, prompt, P, o, out_ids, out_len, out_score, &retry, draft, n_draft, &acc));
  if (retry) {
    WIS_RET(generate_impl(m, input, 1, prompt, P, o, out_ids, out_len, out_score, &retry, draft, n_draft, &acc));
    if (retry) { set_error("wis_generate_draft: hand-off flag raised without the granule path"); return WIS_E_STATE; }
  }
  if (accepted) *accepted = acc;
  return WIS_OK;
}

int wis_generate_draft_beam(wis_model_t* m, const float* input, const int32_t* prompt, int P, const wis_gen_opts_t* o,
                            const int32_t* draft_tok, const int32_t* draft_org, int n_steps, int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* accepted_steps) {
  if (!m || !input || !prompt || !o || !out_ids || !out_len || n_steps < 0 || (n_steps > 0 && (!draft_tok || !draft_org))) { set_error("wis_generate_draft_beam: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_generate_draft_beam")
  bool retry = false;
  int acc = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    WIS_RET(generate_impl(m, input, 1, prompt, P, o, out_ids, out_len, out_score, &retry, n_steps > 0 ? draft_tok : nullptr, n_steps, &acc, n_steps > 0 ? draft_org : nullptr));
    if (!retry) break;
    if (attempt == 1) { set_error("wis_generate_draft_beam: hand-off flag raised without the granule path"); return WIS_E_STATE; }
  }
  if (accepted_steps) *accepted_steps = acc;
  return WIS_OK;
}

int wis_last_trajectory(wis_model_t* m, int b, int32_t* tok, int32_t* org, int cap_steps, int32_t* n_steps) {
  if (!m || !tok || !org || !n_steps || cap_steps < 0) { set_error("wis_last_trajectory: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_last_trajectory")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  if (b < 0 || b >= m->last_B) { set_error("wis_last_trajectory: utterance %d outside the last call's batch of %d", b, m->last_B); return WIS_E_ARG; }
  const int k = m->last_beam;
  int n = 0;
  WIS_HIP_CHECK(hipMemcpyAsync(&n, m->bs.step_u + b, 4, hipMemcpyDeviceToHost, m->st));
  WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  if (n < 0 || n > 256) { set_error("wis_last_trajectory: step counter %d out of range", n); return WIS_E_STATE; }
  if (n > cap_steps) n = cap_steps;
  std::vector<int> raw((size_t)n * MAX_R * 2);
  if (n) WIS_HIP_CHECK(hipMemcpy(raw.data(), m->bs.traj + (size_t)b * 256 * MAX_R * 2, raw.size() * 4, hipMemcpyDeviceToHost));
  for (int s = 0; s < n; ++s) for (int j = 0; j < k; ++j) { tok[s * k + j] = raw[((size_t)s * MAX_R + j) * 2]; org[s * k + j] = raw[((size_t)s * MAX_R + j) * 2 + 1]; }
  *n_steps = n;
  return WIS_OK;
}

int wis_debug_search(wis_model_t* m, const float* logits, int n_steps, int B, const wis_gen_opts_t* o,
                     int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* out_finish_step, int32_t* out_parent) {
  if (!m || !logits || !o || !out_ids || !out_len || n_steps < 1 || n_steps > 256) { set_error("wis_debug_search: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_debug_search")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  const wis_config_t& c = m->cfg;
  const int beam = o->beam_size < 1 ? 1 : o->beam_size;
  WIS_RET(check_batch(m, B, beam));
  WIS_RET(check_patience(beam, o->patience));
  const int max_new = o->max_new_tokens > 0 ? std::min(o->max_new_tokens, 256) : n_steps;
  if (max_new > n_steps) { set_error("wis_debug_search: %d steps of logits for max_new_tokens %d", n_steps, max_new); return WIS_E_ARG; }
  hipStream_t st = m->st;
  const int Mrows = B * beam, V = c.n_vocab, P = 1;
  WIS_RET(init_beam_state(m, B, beam));
  float patience;
  const SampleCfg sc = make_sample_cfg(m, o, beam, max_new, &patience);
  const float* bias_all = o->suppress_default ? m->bias_all : nullptr;
  std::vector<int> done(B, 0), fin(B, -1), par(Mrows);
  for (int s = 0; s < max_new; ++s) {
    WIS_HIP_CHECK(hipMemcpy2DAsync(m->logits, (size_t)m->n_vocab_pad * 4, logits + (size_t)s * Mrows * V, (size_t)V * 4, (size_t)V * 4, Mrows, hipMemcpyHostToDevice, st));
    // step 0 samples every beam of an utterance from ONE row (wis_generate: the last prompt row; here row b*beam), later steps row b*beam + j
    WIS_RET(launch_logit_stats(st, m->logits, bias_all, m->bias_begin, m->bs.step_u, m->st_max, m->st_sum, m->st_val, m->st_idx, B, sc, beam, s == 0 ? 0 : 1, 0));
    WIS_RET(launch_beam_step(st, m->st_max, m->st_sum, m->st_val, m->st_idx, m->bs, m->rm, B, P, c.n_text_ctx, sc));
    WIS_HIP_CHECK(hipMemcpyAsync(done.data(), m->bs.done, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    WIS_HIP_CHECK(hipMemcpyAsync(par.data(), m->bs.parent, (size_t)Mrows * 4, hipMemcpyDeviceToHost, st));
    WIS_HIP_CHECK(hipStreamSynchronize(st));
    bool all = true;
    for (int b = 0; b < B; ++b) { if (done[b] && fin[b] < 0) fin[b] = s; all = all && done[b]; }
    if (out_parent) for (int r = 0; r < Mrows; ++r) out_parent[(size_t)s * Mrows + r] = par[r];
    if (all) break;
  }
  for (int b = 0; b < B; ++b) if (fin[b] < 0) { set_error("wis_debug_search: utterance %d did not finish within %d steps", b, max_new); return WIS_E_STATE; }
  std::vector<int32_t> ids((size_t)B * 256);
  std::vector<float> sc_h(B);
  WIS_HIP_CHECK(hipMemcpyAsync(ids.data(), m->bs.out_ids, (size_t)B * 256 * 4, hipMemcpyDeviceToHost, st));
  WIS_HIP_CHECK(hipMemcpyAsync(out_len, m->bs.out_len, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  WIS_HIP_CHECK(hipMemcpyAsync(sc_h.data(), m->bs.out_score, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  WIS_HIP_CHECK(hipStreamSynchronize(st));
  for (int b = 0; b < B; ++b) {
    if (out_len[b] > max_new) out_len[b] = max_new;
    for (int t = 0; t < max_new; ++t) out_ids[(size_t)b * max_new + t] = t < out_len[b] ? ids[(size_t)b * max_new + t] : 0;
    if (out_score) out_score[b] = sc_h[b];
    if (out_finish_step) out_finish_step[b] = fin[b];
  }
  return WIS_OK;
}

int wis_debug_handoff(wis_model_t* m, int raise_flag, int* retries, int* spin_disabled) {
  if (!m) { set_error("wis_debug_handoff: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_debug_handoff")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  if (raise_flag) {      // what a combiner does when its bounded spin runs out
    const unsigned one = 1u;
    WIS_HIP_CHECK(hipMemcpyAsync(m->ca_epoch, &one, 4, hipMemcpyHostToDevice, m->st));
    WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  }
  if (retries) *retries = m->handoff_retries;
  if (spin_disabled) *spin_disabled = m->spin_off ? 1 : 0;
  return WIS_OK;
}

int wis_last_timing(const wis_model_t* m, wis_timing_t* t) {
  if (!m || !t) { set_error("wis_last_timing: bad argument"); return WIS_E_ARG; }
  *t = m->timing; return WIS_OK;
}

// rows (b): R = 1, every row reads its own KV slot
static int single_row_setup(wis_model* m, int B, const std::vector<int>& tok, int pos) {
  const int ctx = m->cfg.n_text_ctx;
  std::vector<int> ps(B, pos), slot(B), ls(B);
  for (int r = 0; r < B; ++r) { slot[r] = r; ls[r] = r; }
  return upload_rows(m, tok, ps, slot, ls);
}

int wis_detect_language(wis_model_t* m, const float* input, int input_kind, int B, float* lang_probs) {
  if (!m || !input || !lang_probs) { set_error("wis_detect_language: bad argument"); return WIS_E_ARG; }
  if (m->cfg.n_lang <= 0) { set_error("model has no lang_ids"); return WIS_E_UNSUPPORTED; }
  WIS_ENTER(m, "wis_detect_language")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, 1));
  WIS_RET(stage_input(m, input, input_kind, B));
  WIS_RET(run_encoder(m, B));
  WIS_RET(run_cross_kv(m, B));
  std::vector<int> tok(B, m->cfg.sot);
  for (int attempt = 0; attempt < 2; ++attempt) {
    SpinClaim claim(m, B);
    WIS_RET(single_row_setup(m, B, tok, 0));
    WIS_RET(dec_forward(m, B, 1, B, true, 1, 0));
    WIS_RET(launch_lang_probs(m->st, m->logits, m->n_vocab_pad, m->d_lang_ids, m->cfg.n_lang, m->d_probs, B));
    WIS_HIP_CHECK(hipMemcpyAsync(lang_probs, m->d_probs, (size_t)B * m->cfg.n_lang * 4, hipMemcpyDeviceToHost, m->st));
    bool gave_up = false;
    WIS_RET(spin_gave_up(m, &gave_up));      // (synchronises the stream)
    if (!gave_up) break;
  }
  return WIS_OK;
}

int wis_debug_encode(wis_model_t* m, const float* input, int input_kind, int B, float* enc_out) {
  if (!m || !input || !enc_out) { set_error("wis_debug_encode: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_debug_encode")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, 1));
  WIS_RET(stage_input(m, input, input_kind, B));
  WIS_RET(run_encoder(m, B));
  const int64_t n = (int64_t)B * m->cfg.n_audio_ctx * m->cfg.d_model;
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, m->st, m->mem, m->x, n);   // x is free after the encoder
  WIS_HIP_CHECK(hipMemcpyAsync(enc_out, m->x, (size_t)n * 4, hipMemcpyDeviceToHost, m->st));
  WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  return WIS_OK;
}

int wis_debug_logits(wis_model_t* m, const float* input, int input_kind, int B, const int32_t* dec_in, int T, float* logits) {
  if (!m || !input || !dec_in || !logits || T < 1 || T > m->cfg.n_text_ctx) { set_error("wis_debug_logits: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_debug_logits")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, 1));
  WIS_RET(stage_input(m, input, input_kind, B));
  WIS_RET(run_encoder(m, B));
  WIS_RET(run_cross_kv(m, B));
  const int V = m->cfg.n_vocab;
  for (int t = 0; t < T; ++t) {
    std::vector<int> tok(B);
    for (int b = 0; b < B; ++b) tok[b] = dec_in[b * T + t];
    for (int attempt = 0; attempt < 2; ++attempt) {      // (a pass whose granule hand-off gave up is repeated in the ticket form)
      SpinClaim claim(m, B);
      WIS_RET(single_row_setup(m, B, tok, t));
      WIS_RET(dec_forward(m, B, 1, B, true, 1, 0));
      for (int b = 0; b < B; ++b)
        WIS_HIP_CHECK(hipMemcpyAsync(logits + ((size_t)b * T + t) * V, m->logits + (size_t)b * m->n_vocab_pad, (size_t)V * 4, hipMemcpyDeviceToHost, m->st));
      bool gave_up = false;
      WIS_RET(spin_gave_up(m, &gave_up));
      if (!gave_up) break;
    }
  }
  return WIS_OK;
}

int wis_debug_logits_rows(wis_model_t* m, const float* input, int input_kind, int B, const int32_t* dec_in, int T, int R, float* logits) {
  if (!m || !input || !dec_in || !logits || T < 1 || T > m->cfg.n_text_ctx || R < 1 || R > 16) { set_error("wis_debug_logits_rows: bad argument (1 <= R <= 16)"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_debug_logits_rows")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, 1));
  if (B * R > MAX_ROWS) { set_error("wis_debug_logits_rows: B*R = %d exceeds %d decoder rows per pass", B * R, MAX_ROWS); return WIS_E_STATE; }
  WIS_RET(stage_input(m, input, input_kind, B));
  WIS_RET(run_encoder(m, B));
  WIS_RET(run_cross_kv(m, B));
  const int V = m->cfg.n_vocab;
  // teacher-forced in blocks of R positions: the rows (b, i) of a pass sit at positions t0 + i of utterance b's KV slot (causal by
  // position, like the merged prompt pass of wis_generate), so a pass has B * R rows - with B * R > 8 it takes the batched-row
  // route (dec_forward_frag: fragment images, partial-sum LayerNorm statistics) that wis_debug_logits' one row per utterance never
  // reaches at small B
  for (int t0 = 0; t0 < T; t0 += R) {
    const int rows = std::min(R, T - t0), M = B * rows;
    std::vector<int> tok(M), pos(M), slot(M), ls(M);
    for (int b = 0; b < B; ++b) for (int i = 0; i < rows; ++i) { const int r = b * rows + i; tok[r] = dec_in[b * T + t0 + i]; pos[r] = t0 + i; slot[r] = b; ls[r] = b; }
    for (int r = 0; r < M; ++r) if (tok[r] < 0 || tok[r] >= V) { set_error("wis_debug_logits_rows: token %d out of range", tok[r]); return WIS_E_ARG; }
    for (int attempt = 0; attempt < 2; ++attempt) {      // (a pass whose granule hand-off gave up is repeated in the ticket form)
      SpinClaim claim(m, B);
      WIS_RET(upload_rows(m, tok, pos, slot, ls));
      WIS_RET(dec_forward(m, M, rows, B, true, 1, 0));
      for (int b = 0; b < B; ++b) for (int i = 0; i < rows; ++i)
        WIS_HIP_CHECK(hipMemcpyAsync(logits + ((size_t)b * T + t0 + i) * V, m->logits + (size_t)(b * rows + i) * m->n_vocab_pad, (size_t)V * 4, hipMemcpyDeviceToHost, m->st));
      bool gave_up = false;
      WIS_RET(spin_gave_up(m, &gave_up));
      if (!gave_up) break;
    }
  }
  return WIS_OK;
}

int wis_debug_tree_logits(wis_model_t* m, const float* input, int input_kind, const int32_t* prompt, int P, int beam,
                          const int32_t* tok, const int32_t* org, int n_steps, float* logits) {
  if (!m || !input || !prompt || !tok || !org || !logits || P < 1 || P > 16 || beam < 1 || beam > MAX_R || n_steps < 1 || n_steps > std::min(32, MAX_ROWS / std::max(beam, 1))) {
    set_error("wis_debug_tree_logits: bad argument (1 <= n_steps <= min(32, %d / beam))", MAX_ROWS); return WIS_E_ARG;
  }
  WIS_ENTER(m, "wis_debug_tree_logits")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, 1, beam));
  const wis_config_t& c = m->cfg; hipStream_t st = m->st;
  const int k = beam, V = c.n_vocab;
  for (int i = 0; i < n_steps * k; ++i) if (tok[i] < 0 || tok[i] >= V || org[i] < 0 || org[i] >= k) { set_error("wis_debug_tree_logits: token / origin out of range"); return WIS_E_ARG; }
  WIS_RET(stage_input(m, input, input_kind, 1));
  WIS_RET(run_encoder(m, 1));
  WIS_RET(run_cross_kv(m, 1));
  m->spin_now = false;
  // the prompt: rows at positions 0 .. P-1 of slot 0, then every slot gets a copy (what the first step's kv_reorder does)
  std::vector<int> ptok(P), ppos(P), pslot(P, 0), pls(P, 0);
  for (int i = 0; i < P; ++i) { ptok[i] = prompt[i]; ppos[i] = i; }
  WIS_RET(upload_rows(m, ptok, ppos, pslot, pls));
  WIS_RET(dec_forward(m, P, P, 1, false, beam, 0));
  int* hs = m->h_pin + 4096;
  for (int j = 0; j < MAX_R; ++j) hs[j] = 0;
  hs[8] = 1; hs[9] = 0;
  WIS_HIP_CHECK(hipMemcpyAsync(m->bs.parent, hs, (size_t)k * 4, hipMemcpyHostToDevice, st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->bs.step_u, hs + 8, 4, hipMemcpyHostToDevice, st));
  WIS_HIP_CHECK(hipMemcpyAsync(m->bs.done, hs + 9, 4, hipMemcpyHostToDevice, st));
  WIS_RET(launch_kv_reorder(st, m->kc_all, m->vc_all, m->kv_layer_stride, c.n_dec_layers, m->bs, 1, beam, P, c.n_text_ctx, c.d_model));
  // one window: steps 1 .. n_steps, rows from trajectory entries 0 .. n_steps - 1
  std::vector<int> hd((size_t)n_steps * MAX_R * 2, 0);
  for (int s_ = 0; s_ < n_steps; ++s_) for (int j = 0; j < k; ++j) { hd[(s_ * MAX_R + j) * 2] = tok[s_ * k + j]; hd[(s_ * MAX_R + j) * 2 + 1] = org[s_ * k + j]; }
  std::vector<int> wt, wp, wsl, wls;
  int* ha = m->h_pin + 12352;
  const int Mpad = fill_tree_window(hd.data(), 1, n_steps, k, P, wt, wp, wsl, wls, ha);
  WIS_RET(upload_rows(m, wt, wp, wsl, wls, false, 15488));
  WIS_HIP_CHECK(hipMemcpyAsync(m->d_anc, ha, (size_t)Mpad * ANC_W * 4, hipMemcpyHostToDevice, st));
  const TreeWin tw{m->d_anc, P, ANC_W};
  WIS_RET(dec_forward(m, Mpad, 16, Mpad / 16, true, 1, 0, &tw));
  WIS_HIP_CHECK(hipMemcpy2DAsync(logits, (size_t)V * 4, m->logits, (size_t)m->n_vocab_pad * 4, (size_t)V * 4, (size_t)n_steps * k, hipMemcpyDeviceToHost, st));
  WIS_HIP_CHECK(hipStreamSynchronize(st));
  return WIS_OK;
}

int wis_debug_phase_cycles(wis_model_t* m, int B, int beam, int pos, uint64_t* out) {
  if (!m || !out) { set_error("wis_debug_phase_cycles: bad argument"); return WIS_E_ARG; }
  if (!WIS_TAPS) { set_error("tuning taps are not compiled in (rebuild with WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1)"); return WIS_E_UNSUPPORTED; }
  WIS_ENTER(m, "wis_debug_phase_cycles")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, beam));
  SpinClaim claim(m, B);      // the hand-off form wis_generate would take for this batch now (and not whatever the last call left behind)
  const int Mrows = B * beam, ctx = m->cfg.n_text_ctx;
  if (pos < 0 || pos >= ctx) { set_error("bad pos"); return WIS_E_ARG; }
  WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  std::vector<int> tok(Mrows, 100), ps(Mrows, pos), slot(Mrows), ls(Mrows);
  for (int r = 0; r < Mrows; ++r) { slot[r] = r; ls[r] = r; }
  WIS_RET(upload_rows(m, tok, ps, slot, ls));
  WIS_HIP_CHECK(hipMemsetAsync(m->d_prof, 0, 8 * 16 * 8, m->st));
  WIS_RET(dec_forward(m, Mrows, beam, B, false, beam, 1));   // warm
  m->prof_on = true;
  int rc = dec_forward(m, Mrows, beam, B, false, beam, 1);
  m->prof_on = false;
  WIS_RET(rc);
  // API order: QKV gemv, out-proj gemv, cross-attn, self-attn, FFN1 gemv, FFN2 gemv  <-  rows 0, 2, 4, 1, 6, 7
  // (more than 8 rows - the batched-row kernels, round 6: the fourth slot carries the cross-attention output projection (row 5); the attention
  // kernels' stamps are the one-utterance forms' and stay empty there)
  static const int rows_small[6] = {0, 2, 4, 1, 6, 7}, rows_frag[6] = {0, 2, 4, 5, 6, 7};
  const int* rows = Mrows > 8 ? rows_frag : rows_small;
  for (int i = 0; i < 6; ++i)
    WIS_HIP_CHECK(hipMemcpyAsync(out + i * 16, m->d_prof + rows[i] * 16, 16 * 8, hipMemcpyDeviceToHost, m->st));
  bool gave_up = false;
  WIS_RET(spin_gave_up(m, &gave_up));      // (synchronises the stream; a raised flag is consumed here, not by the next wis_generate)
  if (gave_up) { set_error("wis_debug_phase_cycles: the granule hand-off gave up during the tap; stamps are not valid"); return WIS_E_STATE; }
  return WIS_OK;
}

int wis_debug_sampling_cycles(wis_model_t* m, uint64_t* out) {
  if (!m || !out) { set_error("wis_debug_sampling_cycles: bad argument"); return WIS_E_ARG; }
  if (!WIS_TAPS) { set_error("tuning taps are not compiled in (rebuild with WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1)"); return WIS_E_UNSUPPORTED; }
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_HIP_CHECK(hipMemcpy(out, m->d_prof + (size_t)m->cfg.n_dec_layers * 8 * 16, 2 * 16 * 8, hipMemcpyDeviceToHost));
  return WIS_OK;
}

int wis_debug_timeline(wis_model_t* m, int B, int beam, int pos, int use_graph, uint64_t* out, int n_out) {
  if (!m || !out) { set_error("wis_debug_timeline: bad argument"); return WIS_E_ARG; }
  if (!WIS_TAPS) { set_error("tuning taps are not compiled in (rebuild with WIS_EXTRA_HIPFLAGS=-DWIS_TAPS=1)"); return WIS_E_UNSUPPORTED; }
  WIS_ENTER(m, "wis_debug_timeline")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  WIS_RET(check_batch(m, B, beam));
  SpinClaim claim(m, B);
  const int Mrows = B * beam, ctx = m->cfg.n_text_ctx, nk = m->cfg.n_dec_layers * 8;
  if (pos < 0 || pos >= ctx || n_out < nk) { set_error("wis_debug_timeline: bad pos / out size (need %d rows)", nk); return WIS_E_ARG; }
  WIS_HIP_CHECK(hipStreamSynchronize(m->st));
  std::vector<int> tok(Mrows, 100), ps(Mrows, pos), slot(Mrows), ls(Mrows);
  for (int r = 0; r < Mrows; ++r) { slot[r] = r; ls[r] = r; }
  WIS_RET(upload_rows(m, tok, ps, slot, ls));
  std::vector<unsigned long long> init((size_t)nk * 16, 0ull);
  for (int k = 0; k < nk; ++k) init[(size_t)k * 16 + 14] = ~0ull;
  m->prof_on = true; m->prof_all = true;
  int rc = WIS_OK;
  hipGraph_t g = nullptr; hipGraphExec_t gx = nullptr;
  do {
    if (use_graph) {
      if (hipStreamBeginCapture(m->st, hipStreamCaptureModeThreadLocal) != hipSuccess) { set_error("capture failed"); rc = WIS_E_HIP; break; }
      rc = dec_forward(m, Mrows, beam, B, true, beam, 1);
      if (hipStreamEndCapture(m->st, &g) != hipSuccess || rc) { if (!rc) { set_error("end capture failed"); rc = WIS_E_HIP; } break; }
      if (hipGraphInstantiate(&gx, g, nullptr, nullptr, 0) != hipSuccess) { set_error("instantiate failed"); rc = WIS_E_HIP; break; }
    }
    for (int it = 0; it < 3 && !rc; ++it) {   // the last iteration is the one reported
      if (hipMemcpyAsync(m->d_prof, init.data(), init.size() * 8, hipMemcpyHostToDevice, m->st) != hipSuccess) { rc = WIS_E_HIP; break; }
      hipStreamSynchronize(m->st);
      if (use_graph) { if (hipGraphLaunch(gx, m->st) != hipSuccess) { set_error("graph launch failed"); rc = WIS_E_HIP; } }
      else rc = dec_forward(m, Mrows, beam, B, true, beam, 1);
      hipStreamSynchronize(m->st);
    }
  } while (0);
  m->prof_on = false; m->prof_all = false;
  if (gx) hipGraphExecDestroy(gx);
  if (g) hipGraphDestroy(g);
  WIS_RET(rc);
  { bool gave_up = false; WIS_RET(spin_gave_up(m, &gave_up)); if (gave_up) { set_error("wis_debug_timeline: the granule hand-off gave up during the tap"); return WIS_E_STATE; } }
  std::vector<unsigned long long> h((size_t)nk * 16);
  WIS_HIP_CHECK(hipMemcpy(h.data(), m->d_prof, h.size() * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < nk; ++k) { out[2 * k] = h[(size_t)k * 16 + 14]; out[2 * k + 1] = h[(size_t)k * 16 + 15]; }
  return WIS_OK;
}

int wis_bench_weight_stream(wis_model_t* m, int M, int passes, float* total_ms, int* launches_per_pass, double* bytes_per_pass) {
  if (!m || M < 1 || M > MAX_ROWS || passes < 1 || !total_ms) { set_error("wis_bench_weight_stream: bad argument"); return WIS_E_ARG; }
  WIS_ENTER(m, "wis_bench_weight_stream")
  WIS_HIP_CHECK(hipSetDevice(m->device));
  const int d = m->cfg.d_model; hipStream_t st = m->st;
  int launches = 0; double bytes = 0;
  static const bool no_frag = getenv("WIS_NO_FRAG") != nullptr;
  const bool frag = M > 8 && !no_frag;          // the route dec_forward takes at this row count
  const int MBf = cdiv(M, 16);
  static const int ln_form = [] { const char* e = getenv("WIS_B1_LN"); return !e ? 0 : (!strcmp(e, "f16") ? 2 : (!strcmp(e, "partials") ? 1 : 0)); }();
  const bool ln_ok = !frag && M <= 8 && m->cq_fold && d % 64 == 0 && M * (d / 8) <= 13 * 256;      // dec_forward's choice at this row count
  const bool lnp = ln_ok && ln_form == 1 && d <= 1280, ln16 = ln_ok && ln_form == 2 && d <= 2048;
  auto pass = [&](bool count) -> int {
    for (int l = 0; l < m->cfg.n_dec_layers; ++l) {
      const DecLayerW& w = m->dec[l];
      struct { const f16* wp; const float* sc; const float* b; const float* cs; int N, K; bool ln; } mats[6] = {
        {w.p_qkv, w.s_qkv, w.b_qkv, w.c_qkv, 3 * d, d, true}, {w.p_out, w.s_out, w.b_out, nullptr, d, d, false},
        {w.p_cq, w.s_cq, w.b_cq, w.c_cq, d, d, true},        {w.p_cout, w.s_cout, w.b_cout, nullptr, d, d, false},
        {w.p_f1, w.s_f1, w.b_f1, w.c_f1, 4 * d, d, true},    {w.p_f2, w.s_f2, w.b_f2, nullptr, d, 4 * d, false}};
      for (auto& t : mats) {
        GemvP g; memset(&g, 0, sizeof(g));
        g.csum = t.cs; g.Wp = t.wp; g.wscale = t.sc; g.bias = t.b;
        g.y = m->logits; g.M = M; g.N = t.N; g.K = t.K; g.flags = (t.ln ? GV_LN : 0) | GV_OUT_F32;
        g.rows = gemv_rows_for(g.N == m->n_vocab_pad ? m->cfg.n_vocab : g.N, g.K);
        if (frag) { g.x = t.K == d ? (const void*)m->dxf : (const void*)m->dhxf; g.xmb = MBf; g.stat_in = m->dstat; WIS_RET(launch_gemv_frag(st, g)); }
        else {
          g.x = t.ln ? (const void*)m->dx : (const void*)m->dh;
          if (t.ln && lnp) { g.x = m->dxh; g.stat_in = m->dstat; g.flags = GV_LNP | GV_OUT_F32; }      // the form dec_forward launches (f16 rows + partials)
          if (t.ln && ln16) { g.x = m->dxh; g.flags = GV_LN16 | GV_OUT_F32; }
          WIS_RET(launch_ln_gemv(m, st, g));   // (more than 8 rows, round-1 route: LayerNorm runs as its own launch)
        }
        if (count) { ++launches; bytes += (double)t.N * t.K * (m->w8 ? 1 : 2); }
      }
    }
    GemvP g; memset(&g, 0, sizeof(g));
    g.csum = m->c_proj; g.bias = m->b_proj; g.Wp = m->p_proj; g.wscale = m->s_proj; g.y = m->logits; g.M = M; g.N = m->n_vocab_pad; g.K = d; g.flags = GV_LN | GV_OUT_F32;
    g.rows = gemv_rows_for(m->cfg.n_vocab, g.K);
    if (frag) { g.x = m->dxf; g.xmb = MBf; g.stat_in = m->dstat; WIS_RET(launch_gemv_frag(st, g)); }
    else {
      g.x = m->dx;
      if (lnp) { g.x = m->dxh; g.stat_in = m->dstat; g.flags = GV_LNP | GV_OUT_F32; }
      if (ln16) { g.x = m->dxh; g.flags = GV_LN16 | GV_OUT_F32; }
      WIS_RET(launch_ln_gemv(m, st, g));
    }
    if (count) { ++launches; bytes += (double)m->n_vocab_pad * d * (m->w8 ? 1 : 2); }
    return WIS_OK;
  };
  if (frag) {       // zero rows: fragment image of zeros (allocation state), partial sums of zeros
    WIS_HIP_CHECK(hipMemsetAsync(m->dxf, 0, (size_t)(d / 32) * (MAX_ROWS / 16) * 64 * 8 * 2, st));
    WIS_HIP_CHECK(hipMemsetAsync(m->dhxf, 0, (size_t)(4 * d / 32) * (MAX_ROWS / 16) * 64 * 8 * 2, st));
    WIS_HIP_CHECK(hipMemsetAsync(m->dstat, 0, (size_t)MAX_ROWS * (d / 16) * 2 * 4, st));
  }
  WIS_HIP_CHECK(hipMemsetAsync(m->dx, 0, (size_t)MAX_ROWS * d * 4, st));
  WIS_HIP_CHECK(hipMemsetAsync(m->dh, 0, (size_t)MAX_ROWS * 4 * d * 2, st));
  if (lnp || ln16) {
    WIS_HIP_CHECK(hipMemsetAsync(m->dxh, 0, (size_t)MAX_ROWS * d * 2, st));
    WIS_HIP_CHECK(hipMemsetAsync(m->dstat, 0, (size_t)MAX_ROWS * (d / 16) * 2 * 4, st));
  }
  WIS_RET(pass(true));   // warm-up pass (also counts launches / bytes)
  // the timed passes run the way the product runs these kernels: captured once into a HIP graph and replayed (wis_generate replays
  // its decode step as a graph); WIS_NO_GRAPH=1 (profilers that cannot follow a capture) falls back to eager launches
  hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
  if (m->use_graph) {
    WIS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = pass(false);
    const hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc || e != hipSuccess) { if (graph) hipGraphDestroy(graph); if (rc) return rc; set_error("weight-stream tap: graph capture failed: %s", hipGetErrorString(e)); return WIS_E_HIP; }
    if (hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess) { hipGraphDestroy(graph); set_error("weight-stream tap: graph instantiate failed"); return WIS_E_HIP; }
    hipGraphDestroy(graph);
    if (hipGraphLaunch(gexec, st) != hipSuccess) { hipGraphExecDestroy(gexec); set_error("weight-stream tap: graph launch failed"); return WIS_E_HIP; }      // untimed first replay
  }
  int rc2 = WIS_OK;
  hipError_t e2 = hipEventRecord(m->ev[6], st);
  for (int i = 0; i < passes && !rc2 && e2 == hipSuccess; ++i) { if (gexec) e2 = hipGraphLaunch(gexec, st); else rc2 = pass(false); }
  if (e2 == hipSuccess) e2 = hipEventRecord(m->ev[7], st);
  if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
  if (gexec) hipGraphExecDestroy(gexec);
  WIS_RET(rc2);
  WIS_HIP_CHECK(e2);
  WIS_HIP_CHECK(hipEventElapsedTime(total_ms, m->ev[6], m->ev[7]));
  if (launches_per_pass) *launches_per_pass = launches;
  if (bytes_per_pass) *bytes_per_pass = bytes;
  return WIS_OK;
}

// ---- raw device helpers + single-kernel entry points -----------------------------------
int wis_dev_alloc(int device, size_t bytes, void** out) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  WIS_HIP_CHECK(hipMalloc(out, bytes ? bytes : 16)); return WIS_OK;
}
int wis_dev_free(int device, void* p) { DeviceCtx* c; WIS_RET(get_ctx(device, &c)); WIS_HIP_CHECK(hipFree(p)); return WIS_OK; }
int wis_dev_h2d(int device, void* dst, const void* src, size_t bytes) { DeviceCtx* c; WIS_RET(get_ctx(device, &c)); WIS_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return WIS_OK; }
int wis_dev_d2h(int device, void* dst, const void* src, size_t bytes) { DeviceCtx* c; WIS_RET(get_ctx(device, &c)); WIS_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return WIS_OK; }
int wis_dev_sync(int device) { DeviceCtx* c; WIS_RET(get_ctx(device, &c)); WIS_HIP_CHECK(hipDeviceSynchronize()); return WIS_OK; }
int wis_dev_copy_peer(int dst_device, void* dst, int src_device, const void* src, size_t bytes) {
  if (!dst || !src) { set_error("wis_dev_copy_peer: bad argument"); return WIS_E_ARG; }
  DeviceCtx* c; WIS_RET(get_ctx(src_device, &c)); WIS_RET(get_ctx(dst_device, &c));      // both devices exist; current = dst
  if (src_device != dst_device) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst_device, src_device) == hipSuccess && can) {
      hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);                            // direct xGMI path; already-enabled is fine
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
  }
  WIS_HIP_CHECK(hipMemcpyPeer(dst, dst_device, src, src_device, bytes));
  return WIS_OK;
}

int wis_op_gemm(int device, const void* A, int lda, const void* W, const float* bias, const float* residual, void* C, int M, int N, int K, int flags) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  hipStream_t st = ctx_stream(c);
  if (flags & 8) {   // split-K = 2 path of the encoder's FFN2: requires bias, residual and fp32 output
    if (!bias || !residual || (flags & 7) != (2 | 4)) { set_error("wis_op_gemm: split-K needs bias, residual, flags 2|4|8"); return WIS_E_ARG; }
    float* scratch = nullptr;
    WIS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scratch), (size_t)2 * M * N * 4));
    int rc = launch_gemm_splitk_resid(st, gemm_plain(reinterpret_cast<const f16*>(A), lda, reinterpret_cast<const f16*>(W), M, N, K), 2, scratch, bias, residual,
                                      reinterpret_cast<float*>(C));
    hipError_t e = hipStreamSynchronize(st);
    hipFree(scratch);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("wis_op_gemm: %s", hipGetErrorString(e)); return WIS_E_HIP; }
    return WIS_OK;
  }
  WIS_RET(launch_gemm_generic(st, gemm_plain(reinterpret_cast<const f16*>(A), lda, reinterpret_cast<const f16*>(W), M, N, K), bias, residual, C, flags));
  WIS_HIP_CHECK(hipGetLastError());
  WIS_HIP_CHECK(hipStreamSynchronize(st));
  return WIS_OK;
}
int wis_op_layernorm(int device, const float* x, const float* gamma, const float* beta, void* y, int M, int d) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  WIS_RET(launch_layernorm(ctx_stream(c), x, gamma, beta, reinterpret_cast<f16*>(y), M, d));
  WIS_HIP_CHECK(hipGetLastError());
  WIS_HIP_CHECK(hipStreamSynchronize(ctx_stream(c)));
  return WIS_OK;
}
int wis_op_enc_attention(int device, const void* qk, const void* vt, void* out, int B, int T, int Tpad, int H) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  // same rule as the encoder: the split-key form (two workgroups per query tile and head, in-launch merge) at small grids
  hipStream_t st = ctx_stream(c);
  float* part = nullptr; unsigned* counters = nullptr;
  const size_t ncnt = (size_t)B * H * cdiv(T, 128);
  int rc = WIS_OK;
  if (hipMalloc(reinterpret_cast<void**>(&part), enc_attention_part_floats(B, T, H) * 4) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&counters), ncnt * 4) != hipSuccess) { set_error("wis_op_enc_attention: out of device memory"); rc = WIS_E_NOMEM; }
  f16* qk2 = nullptr;      // the lazy-reference loop wants log2(e) on Q as well (the engine folds it into the projection): a scaled private copy
  if (!rc && enc_attn_lazy()) {
    const size_t n = (size_t)B * T * 2 * H * 64;
    if (hipMalloc(reinterpret_cast<void**>(&qk2), n * 2) != hipSuccess) { set_error("wis_op_enc_attention: out of device memory"); rc = WIS_E_NOMEM; }
    else {
      hipMemcpyAsync(qk2, qk, n * 2, hipMemcpyDeviceToDevice, st);
      launch_scale_q_log2e(st, qk2, (int64_t)B * T, H * 64);
    }
  }
  if (!rc) {
    hipMemsetAsync(counters, 0, ncnt * 4, st);
    rc = launch_enc_attention(st, qk2 ? qk2 : reinterpret_cast<const f16*>(qk), reinterpret_cast<const f16*>(vt), reinterpret_cast<f16*>(out), B, T, Tpad, H, part, counters, ncnt);
  }
  hipError_t e = hipStreamSynchronize(st);
  hipFree(part); hipFree(counters); hipFree(qk2);
  if (rc) return rc;
  if (e != hipSuccess) { set_error("wis_op_enc_attention: %s", hipGetErrorString(e)); return WIS_E_HIP; }
  return WIS_OK;
}
int wis_op_gemv(int device, const void* x, const float* gamma, const float* beta, const void* W, const float* bias, void* y, int M, int N, int K, int flags) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  hipStream_t st = ctx_stream(c);
  if (flags & GV_QKV) { set_error("wis_op_gemv: flag 16 is internal"); return WIS_E_ARG; }
  const int Npad = cdiv(N, gemv_rows_for(N, K)) * gemv_rows_for(N, K);
  // (tap flags: 32 = quantise the matrix to 8 bits per weight first - NOT GV_LNP; 64 = GV_LN16: x is the F16 copy of the rows, LayerNorm folded)
  const bool w8 = flags & 32, ln16 = (flags & GV_LN16) != 0, ln = (flags & GV_LN) || ln16;
  flags &= ~32;
  if (ln16 && ((flags & GV_LN) || M > 8 || w8)) { set_error("wis_op_gemv: flag 64 (LayerNorm fold on f16 rows): <= 8 rows, f16 weights, without flag 8"); return WIS_E_ARG; }
  if (ln && (!gamma || !beta)) { set_error("wis_op_gemv: flags 8 / 64 need gamma and beta"); return WIS_E_ARG; }
  // the same preparation the model loader does: optional LayerNorm fold into a private copy of W / bias, then packing
  f16 *wp = nullptr, *wtmp = nullptr, *xn = nullptr, *xfr = nullptr; float *wsc = nullptr, *b2 = nullptr, *cs = nullptr, *stt = nullptr;
  int rc = WIS_OK;
  do {
    if (hipMalloc(reinterpret_cast<void**>(&wp), (size_t)Npad * K * 2) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&wtmp), (size_t)N * K * 2) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&b2), (size_t)Npad * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&cs), (size_t)Npad * 4) != hipSuccess ||
        (w8 && hipMalloc(reinterpret_cast<void**>(&wsc), (size_t)Npad * 4) != hipSuccess)) { set_error("wis_op_gemv: out of device memory"); rc = WIS_E_NOMEM; break; }
    hipMemcpyAsync(wtmp, W, (size_t)N * K * 2, hipMemcpyDeviceToDevice, st);
    hipMemsetAsync(b2, 0, (size_t)Npad * 4, st); hipMemsetAsync(cs, 0, (size_t)Npad * 4, st);
    if (bias) hipMemcpyAsync(b2, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, st);
    if (ln && (rc = launch_fold_ln(st, wtmp, gamma, beta, b2, cs, N, K, 0, 1.f))) break;
    const int rows = gemv_rows_for(N, K);
    rc = w8 ? launch_pack_gemv8(st, wtmp, reinterpret_cast<unsigned char*>(wp), wsc, N, Npad, K, 0, 1.f) : launch_pack_gemv(st, wtmp, wp, N, Npad, K, 0, 1.f, rows);
    if (rc) break;
    if (w8 && ln && (rc = launch_csum8(st, wtmp, wsc, cs, N, K, 0, 1.f))) break;
    GemvP g; memset(&g, 0, sizeof(g));
    g.x = x; g.csum = ln ? cs : nullptr; g.Wp = wp; g.wscale = wsc; g.bias = (bias || ln) ? b2 : nullptr; g.y = y; g.M = M; g.N = N; g.K = K; g.flags = flags; g.rows = rows;
    static const bool no_frag = getenv("WIS_NO_FRAG") != nullptr;
    if (M > 8 && !no_frag) {
      // the product's batched route (dec_forward_frag): activations as a fragment image, LayerNorm statistics as row partials
      const int MBf = cdiv(M, 16);
      if (hipMalloc(reinterpret_cast<void**>(&xfr), (size_t)(K / 32) * MBf * 64 * 8 * 2) != hipSuccess ||
          (ln && hipMalloc(reinterpret_cast<void**>(&stt), (size_t)M * (K / 16) * 2 * 4) != hipSuccess)) { set_error("wis_op_gemv: out of device memory"); rc = WIS_E_NOMEM; break; }
      hipMemsetAsync(xfr, 0, (size_t)(K / 32) * MBf * 64 * 8 * 2, st);
      if ((rc = launch_xf_pack(st, x, ln ? 0 : 1, xfr, ln ? stt : nullptr, M, K, MBf))) break;
      g.x = xfr; g.xmb = MBf; g.stat_in = stt;
      float* kp = nullptr; unsigned* kc = nullptr;
      if (!ln && K >= 4096 && (K / 32) % 16 == 0) {      // the product's rule for the K = 4d projection: four K slices per n-tile, merged in the launch
        const size_t nt = (size_t)cdiv(N, 16);
        if (hipMalloc(reinterpret_cast<void**>(&kp), nt * 4 * MBf * 64 * 16) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&kc), nt * 4) != hipSuccess) {
          hipFree(kp); set_error("wis_op_gemv: out of device memory"); rc = WIS_E_NOMEM; break; }
        hipMemsetAsync(kc, 0, nt * 4, st);
        g.ksplit = 4; g.kpart = kp; g.kcnt = kc;
        if (!(flags & GV_RESID)) rc = launch_gemv_frag(st, g);      // a first launch on the same tickets: they must re-arm themselves
      }
      if (!rc) rc = launch_gemv_frag(st, g);
      hipStreamSynchronize(st);
      hipFree(kp); hipFree(kc);
      break;
    }
    if (ln && M > 8) {      // the round-1 split path (WIS_NO_FRAG): plain normalisation, then f16 activations against the folded weights
      if (hipMalloc(reinterpret_cast<void**>(&xn), (size_t)M * K * 2) != hipSuccess) { set_error("wis_op_gemv: out of device memory"); rc = WIS_E_NOMEM; break; }
      if ((rc = launch_layernorm(st, reinterpret_cast<const float*>(x), nullptr, nullptr, xn, M, K))) break;
      g.x = xn; g.csum = nullptr; g.flags &= ~GV_LN;
    }
    rc = launch_gemv(st, g);
  } while (0);
  hipError_t e = hipStreamSynchronize(st);
  hipFree(wp); hipFree(wtmp); hipFree(b2); hipFree(cs); hipFree(wsc); hipFree(xn); hipFree(xfr); hipFree(stt);
  if (rc) return rc;
  if (e != hipSuccess) { set_error("wis_op_gemv: %s", hipGetErrorString(e)); return WIS_E_HIP; }
  return WIS_OK;
}

int wis_op_dec_self_attn(int device, const float* q, const void* kc, const void* vc, const int32_t* pos, void* out,
                         int M, int H, int ctx, int rpu, int sstride, int rmul) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  if (!q || !kc || !vc || !pos || !out || M < 1 || H < 1 || rpu < 1) { set_error("wis_op_dec_self_attn: bad argument"); return WIS_E_ARG; }
  WIS_RET(launch_dec_self_attn(ctx_stream(c), q, reinterpret_cast<const f16*>(kc), reinterpret_cast<const f16*>(vc), pos, reinterpret_cast<f16*>(out),
                               M, H, 64 * H, ctx, rpu, sstride, rmul));
  WIS_HIP_CHECK(hipGetLastError());
  WIS_HIP_CHECK(hipStreamSynchronize(ctx_stream(c)));
  return WIS_OK;
}
static int op_dec_cross_attn(int device, const float* q, const float* xres, const float* qcs, const float* qb, const void* kx, const void* vt, void* out,
                             int B, int R, int H, int T, int chunks) {
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  std::lock_guard<std::mutex> op_lock(ctx_op_mutex(c));
  if (!q || !kx || !vt || !out || B < 1 || H < 1 || T < 1) { set_error("wis_op_dec_cross_attn: bad argument"); return WIS_E_ARG; }
  hipStream_t st = ctx_stream(c);
  float* part = nullptr; unsigned* counters = nullptr; unsigned long long* gran = nullptr; unsigned* epoch = nullptr;
  int rc = WIS_OK;
  const bool small = B * H <= CA_SPIN_MAX_BH;      // the product's rule: the granule hand-off on small grids (launch_dec_cross_attn decides by chunking / rows)
  if (hipMalloc(reinterpret_cast<void**>(&part), (size_t)B * H * 16 * 16 * 66 * 4) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&counters), (size_t)B * H * 4) != hipSuccess ||
      (small && (hipMalloc(reinterpret_cast<void**>(&gran), (size_t)B * H * 6 * 8 * 66 * 8) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&epoch), ((size_t)B * H + 1) * 4) != hipSuccess))) {
    set_error("wis_op_dec_cross_attn: out of device memory"); rc = WIS_E_NOMEM; }
  unsigned flag = 0;
  if (!rc) {
    hipMemsetAsync(counters, 0, (size_t)B * H * 4, st);
    if (small) { hipMemsetAsync(gran, 0, (size_t)B * H * 6 * 8 * 66 * 8, st); hipMemsetAsync(epoch, 0, ((size_t)B * H + 1) * 4, st); }
    for (int rep = 0; rep < 3 && !rc; ++rep)      // three launches: the epochs of the granule form advance from launch to launch
      rc = launch_dec_cross_attn(st, q, reinterpret_cast<const f16*>(kx), reinterpret_cast<const f16*>(vt), reinterpret_cast<f16*>(out), part, counters,
                                 B, R, H, 64 * H, T, cdiv(T, 64) * 64, chunks, nullptr, 0, xres, qcs, qb, gran, epoch);
    if (small && !rc) hipMemcpyAsync(&flag, epoch, 4, hipMemcpyDeviceToHost, st);
  }
  hipError_t e = hipStreamSynchronize(st);
  hipFree(part); hipFree(counters); hipFree(gran); hipFree(epoch);
  if (rc) return rc;
  if (flag) { set_error("wis_op_dec_cross_attn: granule hand-off timed out"); return WIS_E_HIP; }
  if (e != hipSuccess) { set_error("wis_op_dec_cross_attn: %s", hipGetErrorString(e)); return WIS_E_HIP; }
  return WIS_OK;
}
int wis_op_dec_cross_attn(int device, const float* q, const void* kx, const void* vt, void* out, int B, int R, int H, int T, int chunks) {
  return op_dec_cross_attn(device, q, nullptr, nullptr, nullptr, kx, vt, out, B, R, H, T, chunks);
}
int wis_op_dec_cross_attn_folded(int device, const float* q_raw, const float* xres, const float* qcs, const float* qb, const void* kx, const void* vt, void* out,
                                 int B, int R, int H, int T, int chunks) {
  if (!xres || !qcs || !qb) { set_error("wis_op_dec_cross_attn_folded: bad argument"); return WIS_E_ARG; }
  return op_dec_cross_attn(device, q_raw, xres, qcs, qb, kx, vt, out, B, R, H, T, chunks);
}

}  // extern "C"
