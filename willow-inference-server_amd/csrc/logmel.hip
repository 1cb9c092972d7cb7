// logmel.hip — log-mel front-end on gfx950 (SURVEY §8 rows a2+a3).
//
// Replaces reference wis/audio.py:28-51 (pad_or_trim) and :72-103 (log_mel_spectrogram):
//   hann(400, periodic) * frame  ->  400-point real DFT (centre=True, reflect padding,
//   hop 160, last frame dropped)  ->  |X|^2  ->  mel_80[80,201] @ P  ->  log10(clamp 1e-10)
//   ->  max(x, global_max - 8)  ->  (x + 4) / 4.
//
// MI355X design: the windowed DFT is one exact-f32 MFMA GEMM
//   frames[16 x 400] . D[400 x 416]     (v_mfma_f32_16x16x4_f32; D = window-folded cos | -sin,
//                                         fp64-built on the host, 8 re + 8 im columns per tile)
// per 16-frame workgroup; the frame span is staged once in LDS (bank-skewed), the 665 KB
// DFT matrix is pre-packed in MFMA B-fragment order and stays L2 resident.  Power, the
// sparse 80x201 triangular filterbank (391 non-zeros, VALU) and log10 are fused in the same
// kernel; the global max is a per-window atomicMax on an order-preserving key, applied by a
// small finalize kernel that also emits the conv1 input image (time-major f16, zero padded)
// so the mel never has to leave HBM.  All-silent 16-frame tiles (the zero padding behind a
// short utterance) skip the DFT: the DFT of zeros is exactly zero, so results are identical.
#include <math.h>
#include <stdarg.h>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace wis {

// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* get_error() { return g_err; }

constexpr int NFFT = WIS_N_FFT, HOP = WIS_HOP, NMEL = WIS_N_MELS, NFRAMES = WIS_N_FRAMES;
constexpr int NSAMP = WIS_N_SAMPLES;
constexpr int NBIN = NFFT / 2 + 1;       // 201
constexpr int FT = 16;                   // frames per workgroup
constexpr int NCT = 26;                  // column tiles: 8 bins (re|im) each -> 208 bins
constexpr int NBINP = NCT * 8;           // 208
constexpr int SPAN = (FT - 1) * HOP + NFFT;          // 2800 samples per workgroup
constexpr int SKEW(int s) { return s + 2 * (s / HOP); }  // LDS bank skew (see A-operand reads)
constexpr int SPAN_LDS = SPAN + 2 * (SPAN / HOP) + 8;
constexpr int PSTR = NBINP + 1;          // 209: power tile row stride
constexpr int CONV_C = 96;               // conv1 input channels padded 80 -> 96 (K = 288 = 9 x 32)

struct DeviceCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  float* d_dft = nullptr;     // [NCT][25][64][4]  MFMA-B-fragment-packed DFT matrix
  float* d_filt = nullptr;    // [80][201]
  int* d_frange = nullptr;    // [80][2] first / one-past-last non-zero bin
  // wis_logmel workspaces: one per in-flight call (own stream, own staging and scratch), recycled through a free list, so
  // the entry point is re-entrant: concurrent callers never share a buffer or a stream (SURVEY 8(b) conventions)
  std::mutex ws_mu;
  std::vector<struct LogmelWs*> ws_free;
  std::mutex op_mu;           // serialises the single-kernel test taps (wis_op_*), which share `stream`
};

// everything one wis_logmel call touches on the device; owned by exactly one caller between acquire and release
struct LogmelWs {
  hipStream_t stream = nullptr;
  float* d_pcm = nullptr; int64_t* d_nsamp = nullptr; float* d_melout = nullptr;
  float* d_logspec = nullptr; unsigned* d_gmax = nullptr;
  int cap = 0;                // windows
};
static int ws_acquire(DeviceCtx* c, int n_win, LogmelWs** out) {
  LogmelWs* w = nullptr;
  {
    std::lock_guard<std::mutex> lk(c->ws_mu);
    if (!c->ws_free.empty()) { w = c->ws_free.back(); c->ws_free.pop_back(); }
  }
  if (!w) {
    w = new LogmelWs();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; set_error("wis_logmel: stream create failed"); return WIS_E_HIP; }
  }
  if (n_win > w->cap) {       // grown only while exclusively owned: no buffer is ever freed under an in-flight call
    hipFree(w->d_pcm); hipFree(w->d_nsamp); hipFree(w->d_melout); hipFree(w->d_logspec); hipFree(w->d_gmax);
    w->d_pcm = nullptr; w->d_nsamp = nullptr; w->d_melout = nullptr; w->d_logspec = nullptr; w->d_gmax = nullptr; w->cap = 0;
    if (hipMalloc(&w->d_pcm, (size_t)n_win * NSAMP * 4) != hipSuccess || hipMalloc(&w->d_nsamp, (size_t)n_win * 8) != hipSuccess ||
        hipMalloc(&w->d_melout, (size_t)n_win * NMEL * NFRAMES * 4) != hipSuccess ||
        hipMalloc(&w->d_logspec, (size_t)n_win * NMEL * NFRAMES * 4) != hipSuccess || hipMalloc(&w->d_gmax, (size_t)n_win * 4) != hipSuccess) {
      set_error("wis_logmel: out of device memory for %d windows", n_win);
      std::lock_guard<std::mutex> lk(c->ws_mu); c->ws_free.push_back(w);     // cap = 0: the next owner re-allocates
      return WIS_E_NOMEM;
    }
    w->cap = n_win;
  }
  *out = w; return WIS_OK;
}
static void ws_release(DeviceCtx* c, LogmelWs* w) {
  std::lock_guard<std::mutex> lk(c->ws_mu);
  c->ws_free.push_back(w);
}
std::mutex& ctx_op_mutex(DeviceCtx* c) { return c->op_mu; }

static std::mutex g_ctx_mu;
static DeviceCtx* g_ctx[64] = {nullptr};

// Slaney mel scale, as librosa.filters.mel(sr=16000, n_fft=400, n_mels=80) which produced
// the reference asset wis/assets/mel_filters.npz (wis/audio.py:54-69); fp64 then rounded.
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}
void build_mel_filters(float* out /*[80][201]*/) {
  double melpts[NMEL + 2];
  const double mlo = hz_to_mel(0.0), mhi = hz_to_mel(WIS_SAMPLE_RATE / 2.0);
  for (int i = 0; i < NMEL + 2; ++i) melpts[i] = mel_to_hz(mlo + (mhi - mlo) * i / (NMEL + 1));
  for (int m = 0; m < NMEL; ++m) {
    const double enorm = 2.0 / (melpts[m + 2] - melpts[m]);
    for (int k = 0; k < NBIN; ++k) {
      const double f = (WIS_SAMPLE_RATE / 2.0) * k / (NBIN - 1);
      const double lower = (f - melpts[m]) / (melpts[m + 1] - melpts[m]);
      const double upper = (melpts[m + 2] - f) / (melpts[m + 2] - melpts[m + 1]);
      double w = lower < upper ? lower : upper; if (w < 0) w = 0;
      out[m * NBIN + k] = (float)(w * enorm);
    }
  }
}

int get_ctx(int device, DeviceCtx** out) {
  if (device < 0 || device >= 64) { set_error("bad device index %d", device); return WIS_E_ARG; }
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  if (g_ctx[device]) { *out = g_ctx[device]; WIS_HIP_CHECK(hipSetDevice(device)); return WIS_OK; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device >= n) {
    set_error("no HIP device %d (found %d): libwis_hip has no CPU fallback", device, n); return WIS_E_HIP;
  }
  WIS_HIP_CHECK(hipSetDevice(device));
  DeviceCtx* c = new DeviceCtx(); c->device = device;
  WIS_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // DFT matrix, fp64 with exact angle reduction; window folded in (torch.hann_window(400),
  // periodic: 0.5 - 0.5 cos(2 pi n / 400), wis/audio.py:97).
  std::vector<float> dft((size_t)NCT * 25 * 64 * 4, 0.f);
  const double PI = 3.14159265358979323846;
  for (int ct = 0; ct < NCT; ++ct)
    for (int g = 0; g < 25; ++g)
      for (int lane = 0; lane < 64; ++lane)
        for (int u = 0; u < 4; ++u) {
          const int ks = 4 * g + u, n = 4 * ks + (lane >> 4), j = lane & 15;
          const int bin = 8 * ct + (j & 7);
          float v = 0.f;
          if (bin < NBIN) {
            const double w = 0.5 - 0.5 * cos(2.0 * PI * n / NFFT);
            const double ang = 2.0 * PI * (double)((n * bin) % NFFT) / NFFT;
            v = (float)(j < 8 ? w * cos(ang) : -w * sin(ang));
          }
          dft[(((size_t)ct * 25 + g) * 64 + lane) * 4 + u] = v;
        }
  std::vector<float> filt((size_t)NMEL * NBIN);
  build_mel_filters(filt.data());
  std::vector<int> fr(NMEL * 2);
  for (int m = 0; m < NMEL; ++m) {
    int lo = NBIN, hi = 0;
    for (int k = 0; k < NBIN; ++k) if (filt[m * NBIN + k] != 0.f) { if (k < lo) lo = k; hi = k + 1; }
    if (lo > hi) lo = hi = 0;
    fr[2 * m] = lo; fr[2 * m + 1] = hi;
  }
  WIS_HIP_CHECK(hipMalloc(&c->d_dft, dft.size() * 4));
  WIS_HIP_CHECK(hipMalloc(&c->d_filt, filt.size() * 4));
  WIS_HIP_CHECK(hipMalloc(&c->d_frange, fr.size() * 4));
  WIS_HIP_CHECK(hipMemcpy(c->d_dft, dft.data(), dft.size() * 4, hipMemcpyHostToDevice));
  WIS_HIP_CHECK(hipMemcpy(c->d_filt, filt.data(), filt.size() * 4, hipMemcpyHostToDevice));
  WIS_HIP_CHECK(hipMemcpy(c->d_frange, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
  g_ctx[device] = c; *out = c;
  return WIS_OK;
}
hipStream_t ctx_stream(DeviceCtx* c) { return c->stream; }

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fkey(float x) {
  unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// grid (188, n_win), block 256
__global__ __launch_bounds__(256) void logmel_stft_kernel(
    const float* __restrict__ pcm, int64_t stride, const int64_t* __restrict__ nsamp,
    const float* __restrict__ dft, const float* __restrict__ filt, const int* __restrict__ frange,
    float* __restrict__ logspec, unsigned* __restrict__ gmax, int tile0) {
  __shared__ float s_x[SPAN_LDS];
  __shared__ float s_p[FT * PSTR];
  __shared__ float s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.y, f0 = (blockIdx.x + tile0) * FT;   // tile0: first 16-frame tile of this launch (streaming sessions)
  const float* x = pcm + (int64_t)w * stride;
  int64_t nv = nsamp[w]; if (nv > NSAMP) nv = NSAMP;   // pad_or_trim (wis/audio.py:28-51)

  // stage the 2800-sample span with torch.stft's centre/reflect padding of the 480000 window
  int nz = 0;
  for (int t = tid; t < SPAN; t += 256) {
    int g = HOP * f0 - NFFT / 2 + t;
    if (g < 0) g = -g;
    if (g >= NSAMP) g = 2 * (NSAMP - 1) - g;
    float v = (g >= 0 && g < nv) ? x[g] : 0.f;
    nz |= (v != 0.f);
    s_x[SKEW(t)] = v;
  }
  const int any = __syncthreads_or(nz);

  if (any) {
    // D[frame][col] = sum_n x[160 frame + n] * dft[n][col]; wave handles column tiles
    // {wave, wave+4, ...} two at a time (shared A fragment, two independent accumulators).
    const int arow = lane & 15, akq = lane >> 4;
    const float* xa = s_x;  // A[i = arow][k = akq] at step ks: sample index 160*arow + 4*ks + akq
    for (int ct = wave; ct < NCT; ct += 8) {
      const int ct2 = ct + 4;
      const bool has2 = ct2 < NCT;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float4* b0 = reinterpret_cast<const float4*>(dft) + ((size_t)ct * 25) * 64 + lane;
      const float4* b1 = reinterpret_cast<const float4*>(dft) + ((size_t)(has2 ? ct2 : ct) * 25) * 64 + lane;
#pragma unroll 5
      for (int g = 0; g < 25; ++g) {
        const float4 vb0 = b0[(size_t)g * 64];
        const float4 vb1 = b1[(size_t)g * 64];
        const float bb0[4] = {vb0.x, vb0.y, vb0.z, vb0.w};
        const float bb1[4] = {vb1.x, vb1.y, vb1.z, vb1.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s = HOP * arow + 4 * (4 * g + u) + akq;
          const float a = xa[SKEW(s)];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb0[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb1[u], acc1, 0, 0, 0);
        }
      }
      // acc[r]: frame 4*(lane>>4)+r, column lane&15 (0-7 re, 8-15 im of bin 8ct + (lane&7))
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p0 = acc0[r] * acc0[r]; p0 += __shfl_xor(p0, 8);
        float p1 = acc1[r] * acc1[r]; p1 += __shfl_xor(p1, 8);
        if ((lane & 15) < 8) {
          const int fr = 4 * (lane >> 4) + r;
          s_p[fr * PSTR + 8 * ct + (lane & 7)] = p0;
          if (has2) s_p[fr * PSTR + 8 * ct2 + (lane & 7)] = p1;
        }
      }
    }
  }
  __syncthreads();

  // mel filterbank (sparse rows) + log10; 80 x 16 outputs over 256 threads
  float lmax = -INFINITY;
  for (int o = tid; o < NMEL * FT; o += 256) {
    const int f = o & (FT - 1), m = o >> 4;
    float v = 0.f;
    if (any) {
      const int lo = frange[2 * m], hi = frange[2 * m + 1];
      const float* fm = filt + m * NBIN;
      const float* pr = s_p + f * PSTR;
      for (int k = lo; k < hi; ++k) v = fmaf(fm[k], pr[k], v);
    }
    const float lg = log10f(fmaxf(v, 1e-10f));
    if (f0 + f < NFRAMES) {
      logspec[((size_t)w * NMEL + m) * NFRAMES + f0 + f] = lg;
      lmax = fmaxf(lmax, lg);
    }
  }
  lmax = wave_max(lmax);
  if (lane == 0) s_red[wave] = lmax;
  __syncthreads();
  if (tid == 0) {
    const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    atomicMax(gmax + w, fkey(mx));
  }
}

// grid (ceil(3000/64), n_win), block 256: clamp to max-8, scale, write f32 [80][3000] and/or
// the conv1 input image f16 [3002][96] (row t+1 = frame t; pad rows/cols stay zero).
__global__ __launch_bounds__(256) void logmel_finalize_kernel(
    const float* __restrict__ logspec, const unsigned* __restrict__ gmax,
    float* __restrict__ mel, f16* __restrict__ conv_in) {
  __shared__ float s_t[NMEL][65];
  const int tid = threadIdx.x, w = blockIdx.y, f0 = blockIdx.x * 64;
  const float floor_ = fkey_inv(gmax[w]) - 8.0f;
  for (int o = tid; o < NMEL * 64; o += 256) {
    const int m = o >> 6, f = o & 63;
    if (f0 + f < NFRAMES) {
      const size_t idx = ((size_t)w * NMEL + m) * NFRAMES + f0 + f;
      const float v = (fmaxf(logspec[idx], floor_) + 4.0f) / 4.0f;
      if (mel) mel[idx] = v;
      s_t[m][f] = v;
    }
  }
  if (!conv_in) return;
  __syncthreads();
  for (int o = tid; o < 64 * CONV_C; o += 256) {
    const int f = o / CONV_C, c = o - f * CONV_C;
    if (f0 + f < NFRAMES)
      conv_in[((size_t)w * (NFRAMES + 2) + f0 + f + 1) * CONV_C + c] = (c < NMEL) ? (f16)s_t[c][f] : (f16)0.f;
  }
}

// The frame tiles are independent (a frame depends on 400 samples around it); only the clamp at (global max - 8) couples a
// window, and that is the finalize pass.  d_logspec [n_win][80][3000] / d_gmax [n_win] are the CALLER's scratch: nothing in
// here is shared between two in-flight calls.
int logmel_frames(DeviceCtx* c, hipStream_t stream, float* d_logspec, unsigned* d_gmax, const float* d_pcm, int64_t stride,
                  const int64_t* d_nsamp, int n_win, int tile0, int n_tiles) {
  if (n_win <= 0 || n_tiles <= 0) return WIS_OK;
  if (tile0 < 0 || tile0 + n_tiles > cdiv(NFRAMES, FT)) { set_error("logmel_frames: tiles [%d, %d) out of range", tile0, tile0 + n_tiles); return WIS_E_ARG; }
  hipLaunchKernelGGL(logmel_stft_kernel, dim3(n_tiles, n_win), dim3(256), 0, stream,
                     d_pcm, stride, d_nsamp, c->d_dft, c->d_filt, c->d_frange, d_logspec, d_gmax, tile0);
  WIS_HIP_CHECK(hipGetLastError());
  return WIS_OK;
}
int logmel_finalize(hipStream_t stream, const float* d_logspec, const unsigned* d_gmax, int n_win, float* d_mel, f16* d_conv_in) {
  if (n_win <= 0) return WIS_OK;
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3(cdiv(NFRAMES, 64), n_win), dim3(256), 0, stream, d_logspec, d_gmax, d_mel, d_conv_in);
  WIS_HIP_CHECK(hipGetLastError());
  return WIS_OK;
}
int logmel_device(DeviceCtx* c, hipStream_t stream, float* d_logspec, unsigned* d_gmax, const float* d_pcm, int64_t stride,
                  const int64_t* d_nsamp, int n_win, float* d_mel, f16* d_conv_in) {
  if (n_win <= 0) return WIS_OK;
  WIS_HIP_CHECK(hipMemsetAsync(d_gmax, 0, (size_t)n_win * 4, stream));
  WIS_RET(logmel_frames(c, stream, d_logspec, d_gmax, d_pcm, stride, d_nsamp, n_win, 0, cdiv(NFRAMES, FT)));
  return logmel_finalize(stream, d_logspec, d_gmax, n_win, d_mel, d_conv_in);
}

}  // namespace wis

// ---------------------------------------------------------------------------------------
using namespace wis;

// Re-entrant: every call runs on a workspace of its own (stream + staging + scratch) taken from the device's free list.
extern "C" int wis_logmel(int device, const float* pcm, int64_t stride, const int64_t* n_samples, int n_win,
                          int pcm_on_device, float* mel_out, int mel_on_device) {
  if (!pcm || !n_samples || !mel_out || n_win < 0 || stride < 0) { set_error("wis_logmel: bad argument"); return WIS_E_ARG; }
  if (n_win == 0) return WIS_OK;
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  LogmelWs* ws; WIS_RET(ws_acquire(c, n_win, &ws));
  hipStream_t st = ws->stream;
  auto body = [&]() -> int {
    std::vector<int64_t> ns(n_win);
    const float* dp = pcm; int64_t dstride = stride;
    for (int w = 0; w < n_win; ++w) ns[w] = n_samples[w] < 0 ? 0 : (n_samples[w] > NSAMP ? NSAMP : n_samples[w]);
    if (!pcm_on_device) {
      // copy only the valid samples of each window; the tail is treated as zero by the kernel
      for (int w = 0; w < n_win; ++w)
        if (ns[w]) WIS_HIP_CHECK(hipMemcpyAsync(ws->d_pcm + (size_t)w * NSAMP, pcm + (size_t)w * stride, (size_t)ns[w] * 4, hipMemcpyHostToDevice, st));
      dp = ws->d_pcm; dstride = NSAMP;
    }
    WIS_HIP_CHECK(hipMemcpyAsync(ws->d_nsamp, ns.data(), (size_t)n_win * 8, hipMemcpyHostToDevice, st));
    float* dm = mel_on_device ? mel_out : ws->d_melout;
    WIS_RET(logmel_device(c, st, ws->d_logspec, ws->d_gmax, dp, dstride, ws->d_nsamp, n_win, dm, nullptr));
    if (!mel_on_device)
      WIS_HIP_CHECK(hipMemcpyAsync(mel_out, dm, (size_t)n_win * NMEL * NFRAMES * 4, hipMemcpyDeviceToHost, st));
    WIS_HIP_CHECK(hipStreamSynchronize(st));      // `ns` (pageable) and the caller's buffers are free again
    return WIS_OK;
  };
  const int rc = body();
  if (rc != WIS_OK) hipStreamSynchronize(st);     // nothing of this call may still be in flight when the workspace is recycled
  ws_release(c, ws);
  return rc;
}

// ---------------------------------------------------------------------------------------
// Incremental log-mel: the window's PCM accumulates in HBM; tile j (frames 16j .. 16j+15) reads the samples
// [2560 j - 200, 2560 j + 2600) (reflected at the window edges), so it is final once 2560 j + 2600 samples have arrived - unless
// it reaches the right edge of the 30 s window, where the reflection reads samples up to 479999: those tiles wait for finish.
struct wis_melstream {
  int device = 0;
  DeviceCtx* ctx = nullptr;
  hipStream_t stream = nullptr;
  float* d_pcm = nullptr; int64_t* d_nsamp = nullptr; float* d_logspec = nullptr; unsigned* d_gmax = nullptr; float* d_mel = nullptr;
  int64_t* h_nsamp = nullptr;      // pinned
  int64_t n = 0;                   // samples received
  int tiles_done = 0;
  bool finished = false;
};
namespace {
constexpr int NTILES = (NFRAMES + FT - 1) / FT;      // 188
int tiles_final_for(int64_t n) {
  if (n >= NSAMP) return NTILES;
  int64_t t = (n - (SPAN - NFFT / 2)) / (FT * HOP) + 1;             // largest j + 1 with 2560 j + 2600 <= n
  if (n < SPAN - NFFT / 2) t = 0;
  // tiles whose span crosses the right window edge (reflection) are not final before the window is complete
  const int64_t edge = ((int64_t)NSAMP + NFFT / 2 - SPAN) / (FT * HOP) + 1;   // first tile with 2560 j + 2600 > 480000
  if (t > edge) t = edge;
  return (int)(t < 0 ? 0 : (t > NTILES ? NTILES : t));
}
}  // namespace

extern "C" int wis_melstream_create(int device, wis_melstream_t** out) {
  if (!out) { set_error("wis_melstream_create: bad argument"); return WIS_E_ARG; }
  DeviceCtx* c; WIS_RET(get_ctx(device, &c));
  wis_melstream* s = new wis_melstream();
  s->device = device; s->ctx = c;
  bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess &&
            hipMalloc(&s->d_pcm, (size_t)NSAMP * 4) == hipSuccess && hipMalloc(&s->d_nsamp, 8) == hipSuccess &&
            hipMalloc(&s->d_logspec, (size_t)NMEL * NFRAMES * 4) == hipSuccess && hipMalloc(&s->d_gmax, 4) == hipSuccess &&
            hipMalloc(&s->d_mel, (size_t)NMEL * NFRAMES * 4) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&s->h_nsamp), 64, hipHostMallocDefault) == hipSuccess;
  if (!ok) { set_error("wis_melstream_create: allocation failed"); wis_melstream_destroy(s); return WIS_E_NOMEM; }
  *out = s;
  return wis_melstream_reset(s);
}
extern "C" int wis_melstream_reset(wis_melstream_t* s) {
  if (!s) { set_error("wis_melstream_reset: bad argument"); return WIS_E_ARG; }
  WIS_HIP_CHECK(hipSetDevice(s->device));
  WIS_HIP_CHECK(hipStreamSynchronize(s->stream));
  s->n = 0; s->tiles_done = 0; s->finished = false;
  WIS_HIP_CHECK(hipMemsetAsync(s->d_gmax, 0, 4, s->stream));
  return WIS_OK;
}
static int melstream_run_tiles(wis_melstream* s, int upto) {
  if (upto <= s->tiles_done) return WIS_OK;
  *s->h_nsamp = s->n;                                  // the kernel treats samples >= n as zero (pad_or_trim)
  WIS_HIP_CHECK(hipMemcpyAsync(s->d_nsamp, s->h_nsamp, 8, hipMemcpyHostToDevice, s->stream));
  WIS_RET(logmel_frames(s->ctx, s->stream, s->d_logspec, s->d_gmax, s->d_pcm, NSAMP, s->d_nsamp, 1, s->tiles_done, upto - s->tiles_done));
  WIS_HIP_CHECK(hipStreamSynchronize(s->stream));      // h_nsamp is reused by the next call; a feed is not latency critical
  s->tiles_done = upto;
  return WIS_OK;
}
extern "C" int wis_melstream_feed(wis_melstream_t* s, const float* pcm, int64_t n_samples) {
  if (!s || (!pcm && n_samples > 0) || n_samples < 0) { set_error("wis_melstream_feed: bad argument"); return WIS_E_ARG; }
  if (s->finished) { set_error("wis_melstream_feed: window already finished (reset first)"); return WIS_E_STATE; }
  WIS_HIP_CHECK(hipSetDevice(s->device));
  int64_t take = n_samples;
  if (s->n + take > NSAMP) take = NSAMP - s->n;        // pad_or_trim: a window holds 30 s
  if (take > 0) {
    WIS_HIP_CHECK(hipMemcpyAsync(s->d_pcm + s->n, pcm, (size_t)take * 4, hipMemcpyHostToDevice, s->stream));
    WIS_HIP_CHECK(hipStreamSynchronize(s->stream));    // the caller's buffer is free again
    s->n += take;
  }
  return melstream_run_tiles(s, tiles_final_for(s->n));
}
extern "C" int wis_melstream_finish(wis_melstream_t* s, float* mel_host_or_null, float** mel_dev_out) {
  if (!s) { set_error("wis_melstream_finish: bad argument"); return WIS_E_ARG; }
  WIS_HIP_CHECK(hipSetDevice(s->device));
  if (!s->finished) {
    WIS_RET(melstream_run_tiles(s, NTILES));           // the tail against the zero padding
    WIS_RET(logmel_finalize(s->stream, s->d_logspec, s->d_gmax, 1, s->d_mel, nullptr));
    s->finished = true;
  }
  if (mel_host_or_null) WIS_HIP_CHECK(hipMemcpyAsync(mel_host_or_null, s->d_mel, (size_t)NMEL * NFRAMES * 4, hipMemcpyDeviceToHost, s->stream));
  WIS_HIP_CHECK(hipStreamSynchronize(s->stream));
  if (mel_dev_out) *mel_dev_out = s->d_mel;
  return WIS_OK;
}
extern "C" int64_t wis_melstream_samples(const wis_melstream_t* s) { return s ? s->n : 0; }
extern "C" int wis_melstream_tiles_done(const wis_melstream_t* s) { return s ? s->tiles_done : 0; }
extern "C" void wis_melstream_destroy(wis_melstream_t* s) {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) { hipStreamSynchronize(s->stream); hipStreamDestroy(s->stream); }
  hipFree(s->d_pcm); hipFree(s->d_nsamp); hipFree(s->d_logspec); hipFree(s->d_gmax); hipFree(s->d_mel);
  if (s->h_nsamp) hipHostFree(s->h_nsamp);
  delete s;
}
