/* wis_hip.h — C-ABI of libwis_hip.so: the MI355X-native Whisper ASR hot path for
 * Willow Inference Server (WIS).
 *
 * The reference has no FFI of its own for this path: the boundary is the set of Python
 * call sites in reference main.py / wis/audio.py.  Each entry point below
 * names the reference expression it replaces.  Everything is `extern "C"`, plain pointers
 * and sizes; the caller owns every host buffer, the library owns device memory (except
 * where a `*_dev` pointer is explicitly handed in).  All functions return WIS_OK (0) or a
 * negative WIS_E_* code; the message is available from wis_last_error() (thread-local).
 * No exceptions cross the boundary.  No CPU fallback exists: without a gfx950 device every
 * compute entry point fails with WIS_E_HIP.
 */
#ifndef WIS_HIP_H
#define WIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WIS_ABI_VERSION 1

#define WIS_OK             0
#define WIS_E_ARG         -1   /* bad argument */
#define WIS_E_FORMAT      -2   /* malformed audio container / weight index */
#define WIS_E_NOMEM       -3
#define WIS_E_CHECKSUM    -4   /* FLAC MD5 mismatch */
#define WIS_E_HIP         -5   /* HIP runtime error or no usable device */
#define WIS_E_STATE       -6   /* capacity exceeded / handle misuse */
#define WIS_E_UNSUPPORTED -7

/* audio front-end constants: reference wis/audio.py:17-25 */
#define WIS_SAMPLE_RATE 16000
#define WIS_N_FFT       400
#define WIS_HOP         160
#define WIS_N_MELS      80
#define WIS_N_SAMPLES   480000   /* 30 s window */
#define WIS_N_FRAMES    3000

typedef struct wis_model wis_model_t;

/* ---- library ------------------------------------------------------------------------ */
int         wis_version(void);              /* returns WIS_ABI_VERSION */
const char* wis_last_error(void);           /* valid until the next call on this thread */
int         wis_device_count(void);         /* replaces torch.cuda.device_count(), main.py:252-262 */
/* replaces ctranslate2.get_supported_compute_types(device), main.py:454 — writes a
 * NUL-terminated comma list ("float16,float32") */
int         wis_supported_compute_types(int device, char* out, size_t out_cap);

/* ---- a1: container decode (replaces librosa.load(audio_file, sr=16000, mono=True),
 * main.py:579).  FLAC or RIFF/WAVE bytes -> malloc'ed mono f32 PCM; free with
 * wis_audio_free.  md5_status_out: 1 verified, -1 no signature present (WAV / unsigned). */
int  wis_audio_decode(const void* bytes, size_t n_bytes, float** pcm_out,
                      int64_t* n_samples_out, int* sample_rate_out, int* md5_status_out);
void wis_audio_free(float* pcm);

/* ---- a2+a3: log-mel front-end (replaces wis.audio.log_mel_spectrogram(pad_or_trim(x)),
 * wis/audio.py:28-51,72-103; call sites main.py:606-614).
 * pcm: n_win windows, window w has n_samples[w] valid samples starting at pcm + w*stride
 * (zero-padded / truncated to 480000 on device: pad_or_trim).  mel_out: [n_win][80][3000] f32.
 * pcm_on_device / mel_on_device select host or device pointers for each side. */
int wis_logmel(int device, const float* pcm, int64_t stride, const int64_t* n_samples, int n_win,
               int pcm_on_device, float* mel_out, int mel_on_device);

/* ---- 8(f)3: incremental log-mel over arriving PCM (streaming / long-form sessions).  One handle = ONE 30 s window being
 * filled on one device.  Frames are local (400 samples around 160 t, wis/audio.py:96-101), so every 16-frame tile whose samples
 * have all arrived is transformed inside wis_melstream_feed; only the clamp at (window max - 8) and the scaling (wis/audio.py:
 * 100-102) need the whole window and run in wis_melstream_finish, together with the tail tiles against the zero padding of
 * pad_or_trim.  The result equals wis_logmel of the complete window bit for bit (same kernels, same per-tile arithmetic).
 * The features stay in HBM: finish hands back a DEVICE pointer (f32 [80][3000], valid until the next reset / destroy) that
 * wis_generate / wis_detect_language take as WIS_IN_MEL_DEV on the same device; mel_host_or_null additionally copies them out.
 * A handle is used by one thread at a time; different handles are independent (own stream and buffers). */
typedef struct wis_melstream wis_melstream_t;
int  wis_melstream_create(int device, wis_melstream_t** out);
int  wis_melstream_reset(wis_melstream_t* s);
int  wis_melstream_feed(wis_melstream_t* s, const float* pcm, int64_t n_samples);   /* host PCM, appended; beyond 480000: ignored */
int  wis_melstream_finish(wis_melstream_t* s, float* mel_host_or_null, float** mel_dev_out);
int64_t wis_melstream_samples(const wis_melstream_t* s);       /* samples received so far (<= 480000) */
int  wis_melstream_tiles_done(const wis_melstream_t* s);       /* 16-frame tiles already transformed (0..188) */
void wis_melstream_destroy(wis_melstream_t* s);

/* ---- a6: model lifecycle (replaces ctranslate2.models.Whisper(path, device=..,
 * compute_type=.., device_index=[..]), main.py:341-444).  One handle = one replica on one
 * GPU; the Python shim creates one per listed device_index entry. */
typedef struct {
  int32_t d_model, n_heads, n_enc_layers, n_dec_layers;
  int32_t n_vocab;        /* 51865 multilingual */
  int32_t n_audio_ctx;    /* 1500 */
  int32_t n_text_ctx;     /* 448 */
  int32_t n_mels;         /* 80 */
  int32_t max_batch;      /* utterances (30 s windows) per device batch */
  int32_t max_beam;       /* largest beam_size that will be requested */
  int32_t eot, sot, no_timestamps, no_speech;   /* 50257, 50258, 50363, 50362 */
  const int32_t* suppress_ids;       int32_t n_suppress;        /* CT2 config.json:suppress_ids */
  const int32_t* suppress_ids_begin; int32_t n_suppress_begin;  /* [220, 50257] */
  const int32_t* lang_ids;           int32_t n_lang;            /* 50259..50357 */
  int32_t decoder_weight_bits;  /* 0 / 16: f16 decoder weights (compute type "float16"); 8: per-row int8 weights with f16
                                 * activations ("int8_float16", the reference's GPU default main.py:242) - halves the decode
                                 * weight stream; encoder, embedding lookup and the cross K/V projection stay f16 */
} wis_config_t;

#define WIS_DT_F32 0
#define WIS_DT_F16 1
/* One weight tensor inside the arena.  Names follow the CTranslate2 WhisperSpec variable
 * names, e.g. "encoder/layer_0/self_attention/linear_0/weight". */
typedef struct {
  const char* name;
  int32_t     dtype;       /* WIS_DT_* */
  int32_t     rank;
  int64_t     shape[4];
  uint64_t    offset;      /* byte offset into the arena */
} wis_tensor_t;

/* arena: ONE contiguous block holding every tensor (host memory, or device memory on
 * `device` when arena_on_device != 0 — e.g. the buffer a RCCL broadcast just filled).
 * Weights are re-packed on the GPU into the kernels' layouts; the arena is not retained. */
int  wis_model_create(const wis_config_t* cfg, const void* arena, size_t arena_bytes,
                      int arena_on_device, const wis_tensor_t* tensors, int n_tensors,
                      int device, wis_model_t** out);
void wis_model_destroy(wis_model_t* m);
size_t wis_model_device_bytes(const wis_model_t* m);
/* Another replica on the SAME GPU that shares `parent`'s converted weights (read-only device memory, reference-counted: either
 * handle may be destroyed first) and owns its stream, activations and KV caches.  CTranslate2's `inter_threads` (reference
 * main.py:341-355: parallel batches per model) maps onto this: several device batches in flight on one GPU, each decode chain
 * filling the gaps of the others. */
int  wis_model_clone(wis_model_t* parent, wis_model_t** out);

/* ---- a7-a13: generate (replaces whisper_model.generate(features, [prompt]*B,
 * beam_size=.., return_scores=False) and results[i].sequences_ids[0], main.py:685-693,707,713;
 * decoding defaults are CTranslate2 4.1.0's because WIS passes none). */
#define WIS_IN_MEL_HOST 0   /* f32 [B][80][3000] host   (StorageView.from_array, main.py:638,685) */
#define WIS_IN_MEL_DEV  1   /* same, device memory */
#define WIS_IN_PCM_HOST 2   /* f32 [B][480000] host: log-mel runs on the GPU, mel never leaves HBM */
#define WIS_IN_PCM_DEV  3

typedef struct {
  int32_t input_kind;       /* WIS_IN_* */
  int32_t beam_size;        /* 1 = greedy (CT2 GreedySearch), >1 = beam search */
  int32_t max_new_tokens;   /* 0 => min(n_text_ctx/2, n_text_ctx - P) = 224 */
  float   length_penalty;   /* CT2 default 1 */
  float   patience;         /* CT2 default 1 */
  int32_t suppress_blank;   /* CT2 default 1: suppress_ids_begin masked at the first step */
  int32_t suppress_default; /* CT2 suppress_tokens=[-1]: mask cfg.suppress_ids every step */
  int32_t fixed_new_tokens; /* measurement convention: EOT masked until this many
                               tokens were generated, then forced.  0 = off (product default) */
  int32_t queue_depth;      /* searches that end on EOT: decode steps kept enqueued beyond the last one the host has seen
                               complete (the host polls a progress record the device writes to mapped host memory after every
                               step - no stream round trip); 0 => 2: one step running, one queued, over-run <= 1 step */
} wis_gen_opts_t;

/* out_ids: [B][max_new] (max_new = resolved max_new_tokens), out_len: [B], out_score: [B]
 * (length-normalised log-prob of the returned hypothesis) or NULL.  Blocking.  Thread safety
 * (SURVEY 8b): any number of threads may call into DIFFERENT handles; a handle itself runs one
 * compute call at a time (single-instance activations / KV caches) - a second thread entering a
 * busy handle gets WIS_E_STATE at once, nothing is corrupted.  The shim's micro-batcher feeds each
 * replica from one worker thread and coalesces concurrent requests into device batches. */
int wis_generate(wis_model_t* m, const float* input, int B, const int32_t* prompt, int P,
                 const wis_gen_opts_t* opts, int32_t* out_ids, int32_t* out_len, float* out_score);

/* ---- 8(f)3, BASELINE configs[4]: the FINAL decode of a recording that was heard while it arrived (one utterance, beam_size 1 - the
 * reference's default, settings.py:14).  `draft`: the token ids of an earlier hypothesis for (most of) the same audio, e.g. the last
 * interim transcript of a streaming session.  The encoder runs on the final window as in wis_generate; the draft is then verified
 * in teacher-forced passes of 16 positions (one decoder weight stream per 16 tokens instead of one per token), the longest prefix
 * that greedy decoding of the FINAL window reproduces is kept, and ordinary greedy steps continue behind it.  The result is the
 * greedy decode of the final window - what wis_generate returns for it (same kernels; the multi-row passes sum in a different
 * order than the one-row step, so a decision closer than ~1e-3 in logit can fall differently, as between any two batch shapes).
 * accepted: draft tokens kept (may be NULL).  n_draft == 0 is wis_generate. */
int wis_generate_draft(wis_model_t* m, const float* input, const int32_t* prompt, int P, const wis_gen_opts_t* opts,
                       const int32_t* draft, int n_draft, int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* accepted);

/* ---- the same for a BEAM SEARCH (round 6; the reference decodes every recording of 12 s or more at long_beam_size = 3, main.py:582-586,
 * settings.py:14-18 - BASELINE configs[4]'s 29 s fixture included).  The draft is the TRAJECTORY of an earlier search over (most of) the same
 * audio, as wis_last_trajectory returns it: per step the k live beams it left - draft_tok [n_steps][beam_size] their newest tokens, draft_org
 * [n_steps][beam_size] the live beam (0 .. beam_size-1 of the step before) each continued from.  While the search over the final window
 * follows the draft, up to 32 steps cost ONE decoder pass: the rows of all steps x beam_size tree nodes (<= 96) go through the decoder together,
 * the steps are replayed on their logits by the ordinary sampling kernels, and ordinary steps resume behind the first step whose live set
 * differs from the draft's AS A SET (the slot order of the live beams does not matter: near-tied candidates swap slots between two searches).
 * Every step that counts ran the engine's beam step on the logits of its true inputs: the result is the beam search of the final window,
 * what wis_generate returns for it (up to the summation order of the multi-row passes, as between any two batch shapes).
 * accepted_steps: steps whose live set equalled the draft's (may be NULL).  n_steps == 0 is wis_generate. */
int wis_generate_draft_beam(wis_model_t* m, const float* input, const int32_t* prompt, int P, const wis_gen_opts_t* opts,
                            const int32_t* draft_tok, const int32_t* draft_org, int n_steps,
                            int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* accepted_steps);
/* The trajectory of utterance b of the LAST generate call on this handle (any of the three forms; beam_size k of that call): tok / org
 * [cap_steps][k], *n_steps = steps recorded (the step that ended the search leaves no live set).  Synchronises the handle's stream. */
int wis_last_trajectory(wis_model_t* m, int b, int32_t* tok, int32_t* org, int cap_steps, int32_t* n_steps);

/* ---- a14: language detection (replaces whisper_model.detect_language(features),
 * main.py:637-643): probabilities over cfg.lang_ids, [B][n_lang]. */
int wis_detect_language(wis_model_t* m, const float* input, int input_kind, int B, float* lang_probs);

/* ---- parity taps (teacher-forced); not used by the product path -------------------- */
/* encoder output [B][1500][d] f32 */
int wis_debug_encode(wis_model_t* m, const float* input, int input_kind, int B, float* enc_out);
/* logits [B][T][n_vocab] f32 for decoder inputs dec_in [B][T] (no suppression applied) */
int wis_debug_logits(wis_model_t* m, const float* input, int input_kind, int B,
                     const int32_t* dec_in, int T, float* logits);
/* the same logits computed R (1..16) positions of every utterance per decoder pass: a pass then has B*R rows, i.e. with
 * B*R > 8 it runs the batched-row route of the decode step (the one 8 utterances x beam 5 take, main.py:685-693 with
 * concurrent requests) instead of the <= 8-row route; B*R <= 96 (MAX_ROWS) */
int wis_debug_logits_rows(wis_model_t* m, const float* input, int input_kind, int B,
                          const int32_t* dec_in, int T, int R, float* logits);
/* the decoder pass wis_generate_draft_beam verifies a window with, alone: the tree rows of steps 1 .. n_steps of a beam trajectory (tok / org
 * [n_steps][beam]: per step the live beams' newest tokens and the beam each continued from) behind `prompt`, in ONE pass (tree self-attention by
 * ancestor table, cross-attention as row groups over the utterance's one K / V); logits [n_steps][beam][n_vocab]: row (s, j) = what a step fed
 * tok[s][j] after its chain of ancestors sees - comparable with wis_debug_logits on that chain.  n_steps <= min(32, 96 / beam). */
int wis_debug_tree_logits(wis_model_t* m, const float* input, int input_kind, const int32_t* prompt, int P, int beam,
                          const int32_t* tok, const int32_t* org, int n_steps, float* logits);

/* rows a11-a13 on caller-supplied logits: the SAMPLING kernels wis_generate runs after every decoder pass (logits processors,
 * log-softmax statistics, candidate selection, CTranslate2's beam bookkeeping - dec_kernels.hip logit_stats_kernel /
 * beam_step_kernel) driven by `logits` f32 [n_steps][B*beam][n_vocab] (host) instead of the decoder's output: at step s live beam j of
 * utterance b reads row s*B*beam + b*beam + j (step 0: row b*beam for every beam, like the merged prefill + first step).  Integer
 * bookkeeping only, so the result is compared EXACTLY with the oracle's search over the same table (tests/test_gpu_search.py):
 * an EOT arriving at any step, hypotheses of unequal length, refill from the secondary candidates, patience / early exit,
 * utterances of one device batch finishing at different steps.  opts->max_new_tokens (0 => n_steps) <= n_steps <= 256.
 * out_ids [B][max_new], out_len [B], out_score [B] or NULL, out_finish_step [B] or NULL (step index at which the utterance ended),
 * out_parent [n_steps][B*beam] or NULL (KV slot every live beam continues from after each step: what kv_reorder_kernel applies). */
int wis_debug_search(wis_model_t* m, const float* logits, int n_steps, int B, const wis_gen_opts_t* opts,
                     int32_t* out_ids, int32_t* out_len, float* out_score, int32_t* out_finish_step, int32_t* out_parent);

/* the decoder cross-attention's granule hand-off (small grids; dec_kernels.hip SPIN): retries = calls on this handle that were run
 * again in the ticket form because a combiner's bounded spin ran out (a request never fails for it; expected 0 - the granule form is
 * only taken while B * heads * live handles on the GPU <= 192), spin_disabled = the handle has switched to the ticket form for good.
 * raise_flag != 0 raises the give-up flag by hand, so that the NEXT compute call exercises the repeat path (tests). */
int wis_debug_handoff(wis_model_t* m, int raise_flag, int* retries, int* spin_disabled);

/* ---- timing taps: wall/device ms of the stages of the LAST wis_generate on this handle */
typedef struct {
  float logmel_ms, encoder_ms, crosskv_ms, prefill_ms, decode_ms, total_ms;
  int32_t decode_steps;          /* decoder passes enqueued (the merged prefill + first step counts as one) */
  int32_t decode_steps_needed;   /* passes after which every utterance had finished; decode_steps - this = over-run */
} wis_timing_t;
int wis_last_timing(const wis_model_t* m, wis_timing_t* t);

/* ---- tuning tap: shader-clock stamps of the phases of decoder layer 0's kernels for one decode forward at
 * text position `pos` (out: [6][16] uint64: QKV gemv, out-proj gemv, cross-attn, self-attn, FFN1 gemv, FFN2 gemv; more than 8 rows: the
 * batched-row kernels, the fourth row is the cross-attention output projection; the cross-attention row carries two stamp sets, entries 0-5 and 7-13:
 * the first K wave and the first V wave of workgroup (0, 0, 0)) */
int wis_debug_phase_cycles(wis_model_t* m, int B, int beam, int pos, uint64_t* out);

/* ---- tuning tap: device timeline of one decode forward (B x beam rows at text position `pos`): for each of the
 * n_dec_layers * 8 layer kernels (QKV, self-attn, out-proj, cross-Q, cross-attn, cross-out, FFN1, FFN2) the 100 MHz
 * constant-clock time of its first workgroup start and last workgroup end -> out[k][2].  use_graph = 1 replays the
 * forward as a HIP graph (what wis_generate does), 0 launches it eagerly. */
int wis_debug_timeline(wis_model_t* m, int B, int beam, int pos, int use_graph, uint64_t* out, int n_out);
/* ---- tuning tap (tap builds): shader-clock phase stamps of the last beam_step_kernel launch of utterance 0 -> out[2][16]
 * (row 0 reserved for the logit statistics kernel, row 1 = beam step) */
int wis_debug_sampling_cycles(wis_model_t* m, uint64_t* out);

/* ---- roofline tap (bench.py): launch the decoder's weight-streaming skinny-GEMM kernel once over
 * EVERY decoder weight matrix of the model (6 per layer + the vocabulary projection = the weight
 * stream of one decode step, >> the 256 MiB Infinity Cache for the large sizes), `passes` times,
 * with M activation rows, bracketed by HIP events on the model's stream.  Outputs: total device
 * milliseconds, launches per pass, algorithmic (weight) bytes per pass. */
int wis_bench_weight_stream(wis_model_t* m, int M, int passes, float* total_ms, int* launches_per_pass,
                            double* bytes_per_pass);

/* ---- raw device helpers + single-kernel entry points (used by tests/, bench.py roofline
 * timing and __graft_entry__; all pointers below are DEVICE pointers on `device`) -------- */
int wis_dev_alloc(int device, size_t bytes, void** out);
int wis_dev_free(int device, void* p);
int wis_dev_h2d(int device, void* dst, const void* src, size_t bytes);
int wis_dev_d2h(int device, void* dst, const void* src, size_t bytes);
int wis_dev_sync(int device);
/* device-to-device copy between two GPUs of the node (hipMemcpyPeer: over xGMI where the devices are peers).  The replica pool
 * (ctranslate2.models.Whisper(device_index=[0..n-1]), reference main.py:295) uploads the weight arena from the host ONCE and
 * fans it out to the other replicas with this, in a doubling tree. */
int wis_dev_copy_peer(int dst_device, void* dst, int src_device, const void* src, size_t bytes);

/* C[M][N] = epilogue(A[M][K](lda) . W[N][K]^T + bias): the encoder MFMA GEMM.
 * flags: 1 = GELU, 2 = add residual (f32, [M][N]) , 4 = output f32 (else f16), 8 = split-K x2 variant (with 2|4 only) */
int wis_op_gemm(int device, const void* A_f16, int lda, const void* W_f16, const float* bias,
                const float* residual, void* C, int M, int N, int K, int flags);
/* y f16 [M][d] = LayerNorm(x f32 [M][d]) * gamma + beta, eps 1e-5 */
int wis_op_layernorm(int device, const float* x, const float* gamma, const float* beta,
                     void* y_f16, int M, int d);
/* non-causal MHA over T keys: qk f16 [B*T][2d] (Q pre-scaled | K), vt f16 [B][H][64][Tpad] with key t stored at
 * (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1) (bits 2, 3 swapped inside groups of 16: the P.V MFMA fragment order),
 * out f16 [B*T][d] */
int wis_op_enc_attention(int device, const void* qk_f16, const void* vt_f16, void* out_f16,
                         int B, int T, int Tpad, int H);
/* skinny GEMM used by the decoder: y[M][N] = epi(LN?(x)[M][K] . W[N][K]^T + bias); W is the
 * plain row-major f16 matrix (packed internally for the call).
 * flags: 1 = GELU, 2 = residual add in place into y f32, 4 = y f32 (else f16), 8 = fuse LayerNorm
 * (x is f32 [M][K], gamma/beta given); without 8, x is f16 [M][K]; 32 = quantise W to per-row int8 first (int8_float16). */
int wis_op_gemv(int device, const void* x, const float* gamma, const float* beta,
                const void* W_f16, const float* bias, void* y, int M, int N, int K, int flags);

/* decoder self-attention of ONE new token per row over its cached history (the kernel inside every decode step):
 * q f32 [M][d] (pre-scaled by 1/sqrt(64)), kc / vc f16 [slots][ctx][d] (row m reads positions 0..pos[m] of logical slot
 * (m / rpu) * sstride + (m % rpu) * rmul), pos i32 [M] -> out f16 [M][d].  64 <= ctx <= 512, d = 64 H. */
int wis_op_dec_self_attn(int device, const float* q, const void* kc_f16, const void* vc_f16, const int32_t* pos, void* out_f16,
                         int M, int H, int ctx, int rpu, int sstride, int rmul);
/* decoder cross-attention of the R (<= 16) query rows of each of B utterances over that utterance's T encoder keys, split into
 * `chunks` key chunks (6 -> 256-key, 12 -> 128-key chunks for T = 1500) with the in-launch combine:
 * q f32 [B*R][d] (pre-scaled), kx f16 [B][H][8][T][8] (element (t, 8 c + j) of head h at ((h*8 + c)*T + t)*8 + j),
 * vt f16 [B][H][64][Tpad] (V transposed, zero padded to Tpad = T rounded up to 64) -> out f16 [B*R][d]. */
int wis_op_dec_cross_attn(int device, const float* q, const void* kx_f16, const void* vt_f16, void* out_f16,
                          int B, int R, int H, int T, int chunks);
/* the same with the LayerNorm-folded query of the one-utterance decode step (model.hip fused_out_cq): q_raw f32 [B*R][d] is the folded
 * projection of the UN-normalised rows, xres f32 [B*R][d] the rows themselves; the kernel reduces mean / rstd of every row and
 * finishes q = rstd (q_raw - mean qcs) + qb (qcs, qb f32 [d]: column sums and bias of the folded projection).  R <= 8, d <= 1280. */
int wis_op_dec_cross_attn_folded(int device, const float* q_raw, const float* xres, const float* qcs, const float* qb,
                                 const void* kx_f16, const void* vt_f16, void* out_f16, int B, int R, int H, int T, int chunks);

#ifdef __cplusplus
}
#endif
#endif /* WIS_HIP_H */
