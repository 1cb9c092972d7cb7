// kernels.hpp — launch wrappers shared between the kernel files and the model driver.
#pragma once
#include "common.hpp"

namespace wis {

// ---- encoder ---------------------------------------------------------------------------
struct GemmP {
  const f16* A; int64_t a_bs; int a_rs; int a_rpb;   // A row m -> A + (m / a_rpb)*a_bs + (m % a_rpb)*a_rs
  const f16* W;                                      // [N][K] row-major
  int M, N, K;
  int klen;                                          // split-K: K range per blockIdx.z slice (0 = whole K, gridDim.z = 1)
  // column sub-ranges (the 8-phase kernel's separate launches for plain and transposed tiles): the launch covers N columns, column c of it
  // is output feature (c / n_span) * n_period + n_phase + c % n_span (n_span = 0: the identity)
  int n_span = 0, n_period = 0, n_phase = 0;
};
static inline GemmP gemm_plain(const f16* A, int lda, const f16* W, int M, int N, int K) {
  GemmP p; p.A = A; p.a_bs = 0; p.a_rs = lda; p.a_rpb = 0x7fffffff; p.W = W; p.M = M; p.N = N; p.K = K; p.klen = 0; return p;
}
int launch_layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, f16* y, int M, int d);
int launch_gemm_generic(hipStream_t st, const GemmP& p, const float* bias, const float* resid, void* C, int flags);
// split-K GEMM for deep-K / narrow-N shapes: `splits` K-slices write fp32 partial tiles to `scratch` ([splits][M][N]), then
// one elementwise pass adds the slices + bias + residual into X (fp32, may alias resid).  Keeps dense 128x128 tiles and still
// launches >= 1 workgroup per CU.
int launch_gemm_splitk_resid(hipStream_t st, const GemmP& p, int splits, float* scratch, const float* bias, const float* resid, float* X,
                             const float* ln_gamma = nullptr, const float* ln_beta = nullptr, f16* Y = nullptr);
int launch_gemm_conv1(hipStream_t st, const GemmP& p, const float* bias, f16* C, int T);
int launch_gemm_conv2(hipStream_t st, const GemmP& p, const float* bias, const float* pos, float* X, int T);
int launch_gemm_qkv(hipStream_t st, const GemmP& p, const float* bias, f16* qk, f16* vt, int d, int T, int Tpad, int H);
int launch_gemm_crosskv(hipStream_t st, const GemmP& p, const float* bias, f16* kx, f16* vt, int d, int T, int Tpad, int H, int64_t kx_lstride, int64_t vt_lstride);
// part / counters: split-key scratch sized for `part_cap` (utterance, head, 128-query tile) triples (enc_attention_part_floats); the
// split-key form is only taken when the launch fits that capacity
int launch_enc_attention(hipStream_t st, const f16* qk, const f16* vt, f16* out, int B, int T, int Tpad, int H, float* part = nullptr, unsigned* counters = nullptr,
                         size_t part_cap = 0);
size_t enc_attention_part_floats(int B, int T, int H);
// true: the lazy-reference loop (enc_attn_lazy_kernel), which expects Q pre-multiplied by log2(e) / sqrt(64); fixed per process
bool enc_attn_lazy();
void launch_scale_q_log2e(hipStream_t st, f16* qk, int64_t rows, int d);      // (op-level entry: Q half of a [rows][2d] image times log2(e))

// ---- decoder ---------------------------------------------------------------------------
constexpr int MAX_ROWS = 96;      // decoder rows per pass: B*beam (decode) or B*P (merged prefill + first step): 16 utterances x beam 5 + slack; wis_hip/ctranslate2.py MAX_DECODER_ROWS
constexpr int MAX_R = 8;          // rows per utterance (beam or prompt prefix length)
constexpr int MAX_CAND = 2 * MAX_R;
constexpr int MAX_HYP = 3 * MAX_R;      // finished hypotheses an utterance can hold (model.hip make_sample_cfg)

// per-row decode metadata (device arrays, length MAX_ROWS)
struct RowMeta {
  int* tok;     // input token of the row
  int* pos;     // text position of the row (KV-cache index it writes)
  int* slot;    // physical KV slot the row writes
  int* lslot;   // logical KV slot of the row (prefill rows of an utterance share one)
};

// skinny GEMM (decoder): y = epi(LN?(x) . Wp^T + b), Wp packed in MFMA 16x16x32 A-fragment order
enum { GV_GELU = 1, GV_RESID = 2, GV_OUT_F32 = 4, GV_LN = 8, GV_QKV = 16,
       GV_LNP = 32,
       GV_LN16 = 64 };     // LayerNorm-folded projection on f16 rows, statistics taken from those rows in the kernel (<= 8 rows)      // LayerNorm-folded projection on f16 rows whose statistics arrive as per-16-column (sum, M2) partials in stat_in (<= 16 rows)
struct GemvP {
  const void* x;                 // f32 [M][K] (raw, un-normalised) when GV_LN else f16 [M][K]
  const float* gamma; const float* beta;   // only used by the split path (model.hip launch_ln_gemv); the kernel never reads them
  const float* csum;             // GV_LN: column sums of the gamma-folded weights (see fold_ln_kernel); bias then is b + W.beta
  const f16* Wp; const float* bias;
  const float* wscale;           // non-null: Wp is the 8-bit packed image, one dequantisation scale per output row
  void* y;                       // [M][N] f32 (GV_OUT_F32 / GV_RESID in place) or f16
  int M, N, K, flags;
  // GV_QKV epilogue: n < d -> q (f32 [M][d]); d <= n < 2d -> K cache; n >= 2d -> V cache
  float* q; f16* kc; f16* vc; const int* slot; const int* pos; int d; int ctx;   // cache [slots][ctx][d]
  unsigned long long* prof;      // optional phase stamps (workgroup 0)
  int rows;                      // weight rows per workgroup tile (16 / 8 / 4; 0 => 16): must match the packing
  const void* x2; int xsplit;    // f16 activations only: columns >= xsplit are read from x2 (row-major [M][K - xsplit]); x then is [M][xsplit]
  f16* y16;                      // GV_RESID: optional f16 row-major copy of the produced rows
  // ---- batched rows (launch_gemv_frag, M > 8): activations live in HBM in MFMA B-fragment order ("xf", xf_index below)
  int xmb;                       // 16-row blocks of the x fragment image (= ceil(M / 16))
  const float* stat_in;          // GV_LN: per-row partials of the raw fp32 rows, [M][K/16][2] = (sum x, sum (x - tile mean)^2) per 16 columns (merged Welford-style)
  float* stat_out;               // GV_RESID: the same partials of the rows this launch produces, [M][N/16][2]
  f16* y_xf;                     // GV_RESID: f16 fragment image of the produced rows (next projection's input); f16 output: the output itself
  int ymb;                       // 16-row blocks of y_xf (0: f16 output stays row-major [M][N])
  // K split over workgroups (launch_gemv_frag, ksplit > 1): workgroup (n-tile, k-slice) streams its slice, publishes the 16-column sums
  // to kpart [N/16][ksplit][xmb][64] float4 and the last-arriving slice of an n-tile (ticket kcnt[N/16], zero-initialised once,
  // re-armed by the winner) adds the slices in index order and runs the epilogue
  int ksplit; float* kpart; unsigned* kcnt;
  // M split over workgroups (launch_gemv_frag decides; set by the kernel): this workgroup's first row block and the image's row blocks
  int mb0, mbi;
  // the weight matrix as a k-step window of a wider packed image (launch_gemv_frag / launch_gemv_frag3): wks = k-steps per n-tile of
  // the image (0: K / 32, the matrix is the whole image), wk0 = first k-step of the window
  int wks, wk0;
  // self-attention fused into the QKV projection's launch (launch_gemv, GV_QKV at <= 8 decode rows; dec_kernels.hip sa_consume): sa_gran = 8-byte
  // {tag, value} slots [heads][8 rows][q | k | v][64], sa_epoch = one monotonic epoch word per head, sa_flag = the give-up flag word (the cross-
  // attention hand-off's), sa_out = attention output f16 [M][d]; slot of row m = (m / sa_rpu) * sa_sstride + m % sa_rpu; sa_first is set by the launcher
  unsigned long long* sa_gran; unsigned* sa_epoch; unsigned* sa_flag; f16* sa_out; int sa_rpu, sa_sstride, sa_first;
};
int launch_gemv(hipStream_t st, const GemvP& p);
// two f16-activation skinny GEMMs in one launch (+ optionally L2 prefetch riders for the cross-attention that follows: pf_k / pf_v = its K / V^T of utterance 0)
int launch_gemv_dual(hipStream_t st, const GemvP& pa, const GemvP& pb, const f16* pf_k = nullptr, const f16* pf_v = nullptr, int pf_T = 0, int pf_Tpad = 0, int pf_chunks = 0, int pf_H = 0);
// Batched decode rows (8 < M <= 96).  The skinny GEMM reads its activations as ready-made MFMA B fragments straight from L2
// (written in that order by the producing kernel: no per-workgroup LDS staging, no staging barrier, many workgroups per CU), and
// the pre-LN LayerNorm needs no launch of its own: every residual epilogue leaves per-16-column partial sums of the rows it
// produced, the folded projection that follows (W o gamma, b + W.beta, column sums) turns them into mean / rstd in its epilogue.
int launch_gemv_frag(hipStream_t st, const GemvP& p);
// n (2 or 3) such GEMMs of one row count in one launch (f16 weights, no K split, same K / 128 for all): see gemv_frag3_kernel
int launch_gemv_frag3(hipStream_t st, const GemvP* p, int n);
// activation fragment image: element (row m, column k) of an [M][K] matrix, MB = ceil(M / 16) row blocks:
//   [k / 32][m / 16][lane = (m % 16) + 16 * ((k / 8) % 4)][k % 8]      (one 1 KiB wave load per (k-step, row block))
__host__ __device__ static inline size_t xf_index(int m, int k, int MB) {
  return ((size_t)((k >> 5) * MB + (m >> 4)) * 64 + (m & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}
// x (fp32 or f16, row-major [M][K]) -> fragment image (+ optional per-16-column partial sums): test taps and the weight-stream tap
int launch_xf_pack(hipStream_t st, const void* x, int x_f16, f16* xf, float* stat, int M, int K, int MB);
int launch_dec_embed_xf(hipStream_t st, const f16* emb, const f16* pos_emb, const int* tok, const int* pos, float* x, f16* xf, float* stat, int M, int d, int MB);
int launch_fold_ln(hipStream_t st, f16* W, const float* gamma, const float* beta, float* bias, float* csum, int N, int K, int n_scale, float qscale);
int launch_csum8(hipStream_t st, const f16* W, const float* scale, float* csum, int N, int K, int n_scale, float qscale);
int launch_pack_gemv8(hipStream_t st, const f16* W, unsigned char* Wp, float* scale, int N, int Npad, int K, int n_scale, float qscale);
// pack W [N][K] f16 row-major -> Wp [Npad/rows][K/32][4][rows][8]; matrix rows >= N are zero; scale matrix rows
// [0, n_scale) by `scale` (folds the 1/sqrt(dh) query scaling into the projection)
int launch_pack_gemv(hipStream_t st, const f16* W, f16* Wp, int N, int Npad, int K, int n_scale, float scale, int rows = 16);
int gemv_rows_for(int N, int K);     // tile height used for an [N][K] decoder matrix

int launch_dec_embed(hipStream_t st, const f16* emb, const f16* pos_emb, const int* tok, const int* pos, float* x, int M, int d, f16* xh = nullptr, float* stat = nullptr);
// logical slot of row m = (m / rpu) * sstride + (m % rpu) * rmul
// out_mb: 0 = out is row-major f16 [M][d]; > 0 = fragment image with that many 16-row blocks (batched decode)
// anc (optional; draft verification at beam > 1, model.hip verify_beam_draft): the rows of a pass are the nodes of a beam TREE - row m reads
// the cache rows of positions < w0 from slot anc[m * aw] and position w0 + t from slot anc[m * aw + t] (its ancestor at window step t)
int launch_dec_self_attn(hipStream_t st, const float* q, const f16* kc, const f16* vc, const int* pos, f16* out,
                         int M, int H, int d, int ctx, int rpu, int sstride, int rmul, unsigned long long* prof = nullptr, int out_mb = 0,
                         const int* anc = nullptr, int w0 = 0, int aw = 0, const int* base = nullptr, int nb = 8);      // nb: 8-position blocks per pass (2 / 4 / 8; plain form only)      // base (optional): slot of the positions before w0, per row
// cross attention of R rows per utterance over the utterance's T encoder keys.
//   q f32 [B*R][d] (pre-scaled), kx f16 [B][H][8][T][8], vt f16 [B][H][64][Tpad] (zero padded) -> out f16 [B*R][d]
// gran / epoch (optional): the granule hand-off of small grids (dec_kernels.hip, SPIN): gran = 8-byte slots [B*H][6][8][66], epoch =
// ONE flag word (non-zero: a combiner's bounded spin ran out) followed by [B*H] monotonic epoch words, all zero-initialised once
constexpr int CA_SPIN_MAX_BH = 192;      // spinning combiners per launch: fewer than the chip's 256 CUs, so a producer always finds a slot (8 utterances x 20 heads = 160)
int launch_dec_cross_attn(hipStream_t st, const float* q, const f16* kx, const f16* vt, f16* out, float* part, unsigned* counters,
                          int B, int R, int H, int d, int T, int Tpad, int chunks, unsigned long long* prof = nullptr, int out_mb = 0,
                          const float* xres = nullptr, const float* qcs = nullptr, const float* qb = nullptr,   // folded query: see the kernel
                          unsigned long long* gran = nullptr, unsigned* epoch = nullptr,
                          const float* q2 = nullptr, int xres_is_stat = 0,    // batched fold: q = q + q2; xres = row partials [B*R][d/16][2] instead of the rows
                          int kv_shared = 0);      // 1: every row group b reads utterance 0's K / V (the B groups are rows of ONE utterance: draft verification)

// sampling: per-(row, chunk) masked max / sum-exp / top-2k of the logits
struct SampleCfg {
  int n_vocab, n_vocab_pad, eot, beam, n_cand, max_new, fixed_new, suppress_blank, greedy;
  float length_penalty; int max_hyp; int allow_early_exit; int max_candidates;
  int early_exit_hyps;      // finished hypotheses an early exit needs besides a finished top candidate (WIS_EARLY_EXIT_NUM_HYPOTHESES)
};
// CTranslate2's early exit (patience 1, length_penalty 0): `top_beam_finished && hypotheses.size() >= num_hypotheses` (1: the product
// returns one hypothesis) - or, with 0 here, `>= max_candidates` as rounds 1-5 had it.  Unpinned either way (no CTranslate2 offline);
// the oracle's switch is oracle/whisper_ref.py EARLY_EXIT_NEEDS, the deciding cases are in tests/golden/make_ct2_golden.py.
#ifndef WIS_EARLY_EXIT_NUM_HYPOTHESES
#define WIS_EARLY_EXIT_NUM_HYPOTHESES 1
#endif
constexpr int STAT_SUB = 64;      // sub-chunks per logits row (one wave each): statistics and top-n_cand candidates per sub-chunk
int launch_logit_stats(hipStream_t st, const float* logits, const float* bias_all, const float* bias_begin, const int* step_u,
                       float* st_max, float* st_sum, float* st_val, int* st_idx, int B, const SampleCfg& cfg,
                       int lr_b, int lr_j, int lr_off, unsigned long long* prof = nullptr,    // logits row of (b, j) = b*lr_b + j*lr_j + lr_off
                       const int* rowmap = nullptr);      // ... or b*lr_b + rowmap[j]*lr_j + lr_off (device table: draft verification of a beam search)
struct BeamState {
  int* step_u;      // [B] generated-token count so far
  int* done;        // [B]
  int* n_hyp;       // [B]
  float* cum;       // [B*beam] cumulative log-prob of live beams
  int* alive;       // [B*beam][max_new] token history of live beams
  int* parent;      // [B*beam] KV slot each live beam descends from after the last beam step (kv_reorder_kernel applies it)
  float* hyp_score; // [B][max_hyp] raw cumulative score
  int* hyp_len;     // [B][max_hyp]
  int* hyp_tok;     // [B][max_hyp][max_new]
  int* all_done;    // [1] utterances finished so far
  int* out_ids; int* out_len; float* out_score;   // final result [B][max_new], [B], [B]
  // progress of the search, published to the host without a stream round trip (model.hip generate_impl's pacing loop):
  unsigned* tick;                 // device [4]: {workgroups of beam_step_kernel that ended since init, call generation, step at which all_done reached B, -}
  const unsigned* giveup;         // word 0 of the cross-attention hand-off's epoch block (a combiner's bounded spin ran out)
  unsigned long long* host;       // device address of the HOST-mapped (fine-grained) progress block, layout HP_* below
  int* traj;                      // [B][256 steps][MAX_R][2]: (token, beam slot it continued from) of every live beam after each step - the search's
                                  // trajectory, what a later call verifies as its DRAFT (wis_last_trajectory / wis_generate_draft_beam)
};
// host-mapped progress block (uint64 words).  The workgroup of a beam step that ends LAST writes HP_REC with one system-scope
// release store after everything else of the step - finished utterances' results included - has been made visible to the host.
enum { HP_REC = 0,          // (generation << 48) | (steps completed << 32) | (give-up flag << 16) | utterances done
       HP_STAMP0 = 1,       // 100 MHz constant clock at the end of the first beam step of the call (merged prefill + first step)
       HP_STAMP = 2,        // ... at the end of the latest beam step
       HP_DONE_STEP = 3,    // steps completed when the last utterance finished (0 until then)
       HP_DONE_STAMP = 4,   // constant clock at that moment
       HP_WORDS = 8,        // results follow: int out_len[HP_MAXB], float out_score[HP_MAXB], int out_ids[HP_MAXB][256]
       HP_MAXB = MAX_ROWS };
constexpr size_t HP_BYTES = HP_WORDS * 8 + (size_t)HP_MAXB * 8 + (size_t)HP_MAXB * 256 * 4;
__host__ __device__ inline int* hp_out_len(unsigned long long* hp) { return reinterpret_cast<int*>(hp + HP_WORDS); }
__host__ __device__ inline float* hp_out_score(unsigned long long* hp) { return reinterpret_cast<float*>(hp + HP_WORDS) + HP_MAXB; }
__host__ __device__ inline int* hp_out_ids(unsigned long long* hp) { return reinterpret_cast<int*>(hp + HP_WORDS) + 2 * HP_MAXB; }
// after a beam step: every live beam's KV rows (all layers, positions < P - 1 + step) become a copy of its parent's
int launch_kv_reorder(hipStream_t st, f16* kc, f16* vc, size_t layer_stride, int L, const BeamState& bs, int B, int beam, int P, int ctx, int d);
int launch_beam_step(hipStream_t st, const float* st_max, const float* st_sum, const float* st_val, const int* st_idx,
                     const BeamState& bs, const RowMeta& rm, int B, int P, int ctx, const SampleCfg& cfg, unsigned long long* prof = nullptr);
// draft verification at beam > 1 (one utterance; dec_kernels.hip draft_match_kernel / kv_gather_kernel): behind a replayed beam step, match the live set
// it produced with the draft's entry for the step AS A SET (draft = [n_draft][MAX_R][2] in BeamState::traj's layout); vstate ints: 0 steps verified,
// 2 window steps applied, 8.. the draft node each live beam is matched to (logit_stats rowmap), 16.. / 32.. the slots holding each beam's history
// (book = 1: maintained per window step, applied to the cache by launch_kv_gather once per window).  No match: done[0] = 2 parks the search
constexpr int DRAFT_VS_INTS = 32 + MAX_R * 32, DRAFT_VS_PERM = 8, DRAFT_VS_BASE = 16;
int launch_draft_match(hipStream_t st, const BeamState& bs, const int* draft, int n_draft, int beam, int* vstate, int book);
int launch_kv_gather(hipStream_t st, f16* kc, f16* vc, size_t layer_stride, int L, const int* vstate, const int* done, int beam, int w0, int ctx, int d);
// teacher-forced rows: the token a k = 1 beam step would take from each row's statistics, and its log-probability
int launch_greedy_pick(hipStream_t st, const float* st_max, const float* st_sum, const float* st_val, const int* st_idx, int rows, const SampleCfg& cfg, int* tok_out, float* lp_out);
// language detection: softmax over lang_ids of the row's logits
int launch_lang_probs(hipStream_t st, const float* logits, int ld, const int* lang_ids, int n_lang, float* probs, int B);

}  // namespace wis
