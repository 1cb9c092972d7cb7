"""Drop-in for the reference's `wis/audio.py` (same public names, reference main.py:52-57), with the
arithmetic moved to the GPU:

    from wis_hip.audio import log_mel_spectrogram, pad_or_trim, chunk_iter, find_longest_common_sequence

`log_mel_spectrogram` runs the hand-written gfx950 kernels of csrc/logmel.hip through the C-ABI
(`wis_logmel`); it raises if libwis_hip.so or a GPU is missing (no CPU fallback).  `pad_or_trim`,
`chunk_iter` and `find_longest_common_sequence` are integer/host logic and keep the reference's
semantics exactly (wis/audio.py:28-51, 119-134, 139-159).  `load_audio` replaces the
`librosa.load(audio_file, sr=16000, mono=True)` call of main.py:579 (container decode in C, channel mix, resampling to 16 kHz).
"""
import ctypes as C
import threading
import math

import numpy as np

from . import _lib

SAMPLE_RATE = 16000
N_FFT = 400
N_MELS = 80
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH

chunk_length_s = 22
stride_length_s = [4, 4]
chunk_len = chunk_length_s * SAMPLE_RATE
stride_left = stride_length_s[0] * SAMPLE_RATE
stride_right = stride_length_s[1] * SAMPLE_RATE


def load_audio(audio_file):
    """FLAC / WAV path, bytes or file-like -> (float32 mono PCM in [-1, 1), sample_rate).

    Container decode is host C code inside libwis_hip.so (wis_audio_decode); FLAC streams are MD5-verified."""
    if isinstance(audio_file, (bytes, bytearray, memoryview)):
        data = bytes(audio_file)
    elif hasattr(audio_file, "read"):
        if hasattr(audio_file, "seek"):
            audio_file.seek(0)
        data = audio_file.read()
    else:
        with open(audio_file, "rb") as f:
            data = f.read()
    lib = _lib.load()
    p, n, sr, md = C.POINTER(C.c_float)(), C.c_int64(), C.c_int(), C.c_int()
    _lib.check(lib.wis_audio_decode(data, len(data), C.byref(p), C.byref(n), C.byref(sr), C.byref(md)))
    try:
        pcm = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        lib.wis_audio_free(p)
    if sr.value != SAMPLE_RATE:         # librosa.load(..., sr=16000) resamples whatever the container holds (main.py:579)
        pcm = resample(pcm, sr.value, SAMPLE_RATE)
    return pcm, SAMPLE_RATE


def _resample_filter(up, down, half_width=32, beta=14.769656459379492, rolloff=0.945):
    """Kaiser-windowed sinc prototype low-pass of a polyphase up/down resampler, designed at the `up`-times oversampled
    rate: cut-off at rolloff x the lower Nyquist, `half_width` zero crossings each side (at the lower rate), fp64."""
    q = max(up, down)
    n = half_width * q
    t = np.arange(-n, n + 1, dtype=np.float64)
    fc = rolloff / q                                     # cycles per oversampled sample, relative to Nyquist = 1
    h = fc * np.sinc(fc * t) * np.kaiser(2 * n + 1, beta)
    return h * (up / h.sum())


def resample(pcm, sr_in, sr_out=SAMPLE_RATE):
    """Band-limited sample-rate conversion sr_in -> sr_out (host, numpy fp64 accumulation; polyphase Kaiser-windowed sinc,
    64 zero crossings, stop band below -100 dB).  Replaces the resampling half of `librosa.load(audio_file, sr=16000)`
    (main.py:579; librosa's default there is soxr_hq - a different filter of the same class, so the result agrees with it to
    pass-band ripple / transition-band shape, not bit for bit: the reference fixtures are all 16 kHz and never take this path).
    Output length = ceil(n * sr_out / sr_in), as librosa."""
    x = np.ascontiguousarray(pcm, np.float64).reshape(-1)
    sr_in, sr_out = int(sr_in), int(sr_out)
    if sr_in <= 0 or sr_out <= 0:
        raise ValueError(f"bad sample rate {sr_in} -> {sr_out}")
    if sr_in == sr_out or x.shape[0] == 0:
        return x.astype(np.float32)
    g = math.gcd(sr_in, sr_out)
    up, down = sr_out // g, sr_in // g
    h = _resample_filter(up, down)
    half = (h.shape[0] - 1) // 2
    n_out = -(-x.shape[0] * up // down)
    # output m sits at oversampled index m * down; y[m] = sum_k x[k] h[m * down - k * up + half]
    # per output phase r = m mod up:  m = r + j * up  ->  (m * down) = r * down + j * up * down:
    #   k ranges over a window that advances by `down` input samples per j, taps = h[(r * down - k0 * up + half) :: -up]
    taps_per_phase = -(-h.shape[0] // up)
    pad = taps_per_phase + down + 2
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad + taps_per_phase * 2)])
    out = np.zeros(n_out, np.float64)
    for r in range(min(up, n_out)):
        c = r * down                                          # oversampled position of the phase's first output
        k_hi = (c + half) // up                               # last input index with a tap for j = 0
        t0 = c + half - k_hi * up                             # tap index for k_hi (0 <= t0 < up)
        taps = h[t0::up]                                      # taps for k = k_hi, k_hi - 1, ...
        nt = taps.shape[0]
        n_r = (n_out - r + up - 1) // up                      # outputs of this phase
        start = k_hi - (nt - 1) + pad                         # window start in xp for j = 0 (ascending k)
        win = np.lib.stride_tricks.as_strided(xp[start:], shape=(n_r, nt), strides=(xp.strides[0] * down, xp.strides[0]))
        out[r::up] = win @ taps[::-1]
    return out.astype(np.float32)


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Zero-pad on the right / truncate `axis` to `length` samples (numpy arrays)."""
    array = np.asarray(array)
    n = array.shape[axis]
    if n > length:
        array = np.take(array, np.arange(length), axis=axis)
    elif n < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - n)
        array = np.pad(array, widths)
    return array


class MelFeatures:
    """What the reference's `log_mel_spectrogram(...)` result is used for: `.numpy()` (main.py:608,614)."""

    def __init__(self, arr):
        self._arr = arr

    def numpy(self):
        return self._arr

    @property
    def shape(self):
        return self._arr.shape

    def __array__(self, dtype=None, copy=None):
        return self._arr if dtype is None else self._arr.astype(dtype)


def log_mel_spectrogram(audio, n_mels: int = N_MELS, device=None):
    """float32 [480000] (or [n, 480000]) -> MelFeatures wrapping float32 [80, 3000] (or [n, 80, 3000]).
    `device`: the GPU to run on (callers that own a model pass one of ITS replicas' devices, wis_hip.whisper.do_whisper); None =
    device 0 - the call never touches a GPU it was not pointed at.  Re-entrant: concurrent calls never share a stream or a
    buffer (csrc/logmel.hip)."""
    assert n_mels == 80, f"Unsupported n_mels: {n_mels}"
    x = np.ascontiguousarray(np.asarray(audio, dtype=np.float32))
    single = x.ndim == 1
    if single:
        x = x[None]
    if x.ndim != 2 or x.shape[1] != N_SAMPLES:
        raise ValueError(f"log_mel_spectrogram expects pad_or_trim'ed audio of {N_SAMPLES} samples, got {x.shape}")
    _lib.require_gpu()
    device = 0 if device is None else int(device)
    n = x.shape[0]
    out = np.empty((n, N_MELS, N_FRAMES), np.float32)
    ns = (C.c_int64 * n)(*([N_SAMPLES] * n))
    _lib.check(_lib.load().wis_logmel(device, _lib.ptr(x), N_SAMPLES, ns, n, 0, _lib.ptr(out), 0))
    return MelFeatures(out[0] if single else out)


class MelStream:
    """Incremental log-mel of ONE 30 s window on one GPU (C-ABI wis_melstream_*): `feed` PCM as it arrives - every
    16-frame tile whose samples are complete is transformed right away - and `finish` clamps / scales (the only step that needs
    the whole window).  Bit-identical to `log_mel_spectrogram(pad_or_trim(everything fed))`.  After `finish`, `device_ptr` is
    the address of the f32 [80, 3000] features in HBM (valid until reset / close): `Whisper.generate_from_device` consumes it
    without the features ever visiting the host."""

    # idle native handles per device: creating one costs five device allocations, a stream and a pinned block, destroying one as
    # many (synchronising) frees - milliseconds each, and a streaming session opens a window every 14 s of audio and closes it on the
    # latency path of stop().  close() therefore parks the handle here (reset) and the next MelStream on that GPU takes it over.
    _idle, _idle_lock, _IDLE_MAX = {}, threading.Lock(), 16

    def __init__(self, device=0):
        _lib.require_gpu()
        self.device = int(device)
        self._h = None
        with MelStream._idle_lock:
            pool = MelStream._idle.get(self.device)
            if pool:
                self._h = pool.pop()
        if self._h is None:
            self._h = C.c_void_p()
            _lib.check(_lib.load().wis_melstream_create(self.device, C.byref(self._h)))
        self.device_ptr = None

    def feed(self, samples):
        x = np.ascontiguousarray(samples, np.float32).reshape(-1)
        if x.shape[0]:
            _lib.check(_lib.load().wis_melstream_feed(self._h, _lib.ptr(x), x.shape[0]))

    def finish(self, to_host=True):
        out = np.empty((N_MELS, N_FRAMES), np.float32) if to_host else None
        dev = C.c_void_p()
        _lib.check(_lib.load().wis_melstream_finish(self._h, _lib.ptr(out) if to_host else None, C.byref(dev)))
        self.device_ptr = dev.value
        return out

    def reset(self):
        _lib.check(_lib.load().wis_melstream_reset(self._h))
        self.device_ptr = None

    @property
    def samples(self):
        return int(_lib.load().wis_melstream_samples(self._h))

    @property
    def tiles_done(self):
        return int(_lib.load().wis_melstream_tiles_done(self._h))

    def close(self):
        """Give the window up: its features (device_ptr) are no longer valid.  The native handle is reset and parked for reuse."""
        h, self._h = self._h, None
        self.device_ptr = None
        if not h:
            return
        try:
            if _lib.load().wis_melstream_reset(h) == 0:
                with MelStream._idle_lock:
                    pool = MelStream._idle.setdefault(self.device, [])
                    if len(pool) < MelStream._IDLE_MAX:
                        pool.append(h)
                        return
        except Exception:
            pass
        _lib.load().wis_melstream_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def chunk_iter(inputs):
    """22 s windows with 4 s of context each side, stepping 14 s; yields (chunk, (len, left, right))."""
    assert isinstance(inputs, np.ndarray), "chunk_iter only takes numpy array"
    total = inputs.shape[0]
    step = chunk_len - stride_left - stride_right
    start = 0
    while start < total:
        piece = inputs[start:start + chunk_len]
        left = stride_left if start else 0
        right = 0 if start + step + stride_left >= total else stride_right
        if piece.shape[0] > left:
            yield piece, (piece.shape[0], left, right)
        start += step


def _overlap_hits(tail, head):
    """`np.sum(np.array(tail) == np.array(head))` as the reference's pinned numpy 1.23.5 evaluates it
    (wis/audio.py:152, requirements.txt:58): equal lengths compare element by element, a one-element side
    BROADCASTS against the other (a running sequence holding a single token is compared with every
    head token), any other length mismatch is the scalar False of the deprecated elementwise-comparison
    fallback, i.e. 0 (newer numpy raises there instead)."""
    if len(tail) == len(head):
        return sum(1 for a, b in zip(tail, head) if a == b)
    if len(tail) == 1:
        return sum(1 for b in head if b == tail[0])
    if len(head) == 1:
        return sum(1 for a in tail if a == head[0])
    return 0


def find_longest_common_sequence(sequences, tokenizer):
    """Stitch per-window token id lists: for each next window choose the overlap length i that maximises
    matches/i + i/10000 with matches > 1 and append the remainder.  `tokenizer.all_special_ids` are dropped first.
    Bit-exact with wis/audio.py:139-159 under its pinned numpy (see _overlap_hits)."""
    special = set(tokenizer.all_special_ids)
    merged = [t for t in sequences[0][0] if t not in special]
    for entry in sequences[1:]:
        cand = [t for t in entry[0] if t not in special]
        cut, best = 0, 0.0
        for i in range(1, len(cand) + 1):
            hits = _overlap_hits(merged[-i:], cand[:i])
            score = hits / i + i / 10000.0
            if hits > 1 and score > best:
                cut, best = i, score
        merged.extend(cand[cut:])
    return np.array(merged)
