"""ORACLE (test infrastructure, NOT product code): CPU restatement of the reference's audio
front-end, reference file `wis/audio.py`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (willow-inference-server_amd/wis_hip) never does.

Pinning: checked against golden vectors produced by importing the REAL reference
(`/root/reference/wis/audio.py`) in the build container — tests/golden/make_golden.py, fixtures
tests/golden/logmel_*.npz, chunker_lcs.json — see tests/test_oracle_audio.py.
"""
import numpy as np

SAMPLE_RATE = 16000   # wis/audio.py:17
N_FFT = 400           # wis/audio.py:18
N_MELS = 80           # wis/audio.py:19
HOP_LENGTH = 160      # wis/audio.py:20
N_SAMPLES = 480000    # wis/audio.py:22
N_FRAMES = 3000       # wis/audio.py:23-25


def pad_or_trim(array, length=N_SAMPLES, axis=-1):
    """wis/audio.py:28-51 (numpy branch): truncate, then zero-pad on the right."""
    array = np.asarray(array)
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filters(n_mels=N_MELS):
    """The matrix stored in wis/assets/mel_filters.npz (wis/audio.py:54-69): librosa.filters.mel(sr=16000,
    n_fft=400, n_mels=80) = Slaney-scale triangles with Slaney area normalisation.  f32 [80, 201]."""
    assert n_mels == 80
    fftfreqs = np.linspace(0, SAMPLE_RATE / 2, 1 + N_FFT // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(SAMPLE_RATE / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + N_FFT // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def log_mel_spectrogram(audio, dtype=np.float32):
    """wis/audio.py:72-103.  audio: [480000] float32 -> [80, 3000].

    torch.stft(audio, 400, 160, window=hann(400), return_complex=True) defaults: center=True, pad_mode='reflect',
    onesided -> 201 bins x 3001 frames; the last frame is dropped (:99), power (:99), mel matmul (:102),
    log10 of clamp(1e-10) (:104), floor at global max - 8 (:105), (x + 4) / 4 (:106).
    `dtype=np.float64` gives the high-precision variant used to measure noise floors."""
    x = np.asarray(audio, dtype=dtype)
    n = np.arange(N_FFT)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(dtype)          # torch.hann_window(400): periodic
    y = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    frames = np.lib.stride_tricks.sliding_window_view(y, N_FFT)[::HOP_LENGTH]      # [3001, 400]
    stft = np.fft.rfft(frames * window, axis=1)                                     # numpy >= 2 keeps float32 precision
    mag = (np.abs(stft[:-1]) ** 2).T.astype(dtype)                                  # [201, 3000]
    mel = mel_filters().astype(dtype) @ mag
    log_spec = np.log10(np.maximum(mel, dtype(1e-10)))
    log_spec = np.maximum(log_spec, log_spec.max() - dtype(8.0))
    return ((log_spec + dtype(4.0)) / dtype(4.0)).astype(dtype)


# wis/audio.py:106-115
CHUNK_LEN = 22 * SAMPLE_RATE
STRIDE_LEFT = 4 * SAMPLE_RATE
STRIDE_RIGHT = 4 * SAMPLE_RATE


def chunk_iter(inputs):
    """wis/audio.py:119-134: 22 s windows stepping 14 s; yields (chunk, (len, stride_left, stride_right))."""
    n = inputs.shape[0]
    step = CHUNK_LEN - STRIDE_LEFT - STRIDE_RIGHT
    for i in range(0, n, step):
        chunk = inputs[i:i + CHUNK_LEN]
        left = 0 if i == 0 else STRIDE_LEFT
        right = 0 if i + step + STRIDE_LEFT >= n else STRIDE_RIGHT
        if chunk.shape[0] > left:
            yield chunk, (chunk.shape[0], left, right)


def np123_eq_sum(a, b):
    """np.sum(np.array(a) == np.array(b)) under numpy 1.23.5 (the reference's pin, requirements.txt:58): equal lengths ->
    elementwise; one side of length 1 -> broadcast; any other mismatch -> scalar False (DeprecationWarning there, ValueError
    from numpy 1.25 on) -> 0."""
    la, lb = len(a), len(b)
    if la == lb:
        return int(sum(u == v for u, v in zip(a, b)))
    if la == 1:
        return int(sum(v == a[0] for v in b))
    if lb == 1:
        return int(sum(u == b[0] for u in a))
    return 0


def find_longest_common_sequence(sequences, special_ids):
    """wis/audio.py:139-159: stitch per-window token lists by the overlap length i maximising matches/i + i/10000
    (matches > 1 required); strides are carried but ignored, as in the reference.  The match count follows the reference's
    numpy broadcasting (wis/audio.py:152): a running sequence of ONE token is compared against every head token."""
    special = set(special_ids)
    seq = [t for t in sequences[0][0] if t not in special]
    for new in sequences[1:]:
        new_seq = [t for t in new[0] if t not in special]
        index, best = 0, 0.0
        for i in range(1, len(new_seq) + 1):
            matches = np123_eq_sum(seq[-i:], new_seq[:i])
            score = matches / i + i / 10000.0
            if matches > 1 and score > best:
                index, best = i, score
        seq.extend(new_seq[index:])
    return np.array(seq)
