"""not gpu: pin the CPU oracle of the audio front-end (oracle/audio_ref.py) against golden vectors generated
by the REAL reference module /root/reference/wis/audio.py (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import audio_ref


def _decode(golden_dir, clip):
    from wis_hip import audio
    pcm, sr = audio.load_audio(os.path.join(golden_dir, "clips", clip + ".flac"))
    assert sr == 16000
    return pcm


def test_mel_filters_match_reference_asset(golden_dir):
    ref = np.load(os.path.join(golden_dir, "mel_filters.npz"))["mel_80"]
    mine = audio_ref.mel_filters()
    assert mine.shape == ref.shape == (80, 201) and mine.dtype == np.float32
    assert np.abs(mine - ref).max() <= 2e-9          # 1 ulp on <1 % of entries
    assert ((mine != 0) == (ref != 0)).all() and (ref != 0).sum() == 391


@pytest.mark.parametrize("clip", ["3sec", "10sec", "30sec"])
def test_logmel_oracle_vs_reference_golden(golden_dir, clip):
    pcm = _decode(golden_dir, clip)
    ref = np.load(os.path.join(golden_dir, f"logmel_{clip}.npz"))["mel"]
    got = audio_ref.log_mel_spectrogram(audio_ref.pad_or_trim(pcm))
    assert got.shape == (80, 3000) and got.dtype == np.float32
    assert np.abs(got - ref).max() <= 5e-5
    # min = max - 2 exactly (floor at global max - 8, then /4): SURVEY Appendix A.2
    assert abs((ref.max() - ref.min()) - 2.0) < 1e-6 and abs((got.max() - got.min()) - 2.0) < 1e-6


def test_logmel_oracle_noise(golden_dir):
    g = np.load(os.path.join(golden_dir, "logmel_noise.npz"))
    rng = np.random.default_rng(1234)
    for n in (61440, 480000):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        got = audio_ref.log_mel_spectrogram(audio_ref.pad_or_trim(x))
        assert np.abs(got - g[f"mel_{n}"]).max() <= 5e-5


def test_chunker_and_lcs_vs_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "chunker_lcs.json")))
    for n, strides in cases["chunk_iter"].items():
        got = [list(s) for _, s in audio_ref.chunk_iter(np.zeros(int(n), np.float32))]
        assert got == strides, n
    assert len(cases["chunk_iter"]["2880000"]) == 13      # 180 s -> 13 windows
    for c in cases["lcs"]:
        seqs = [(s, (1, 0, 0)) for s in c["seqs"]]
        if isinstance(c["out"], str):
            continue
        assert audio_ref.find_longest_common_sequence(seqs, c["special"]).tolist() == c["out"]


def test_pad_or_trim():
    x = np.arange(10, dtype=np.float32)
    assert audio_ref.pad_or_trim(x, 6).tolist() == [0, 1, 2, 3, 4, 5]
    y = audio_ref.pad_or_trim(x, 12)
    assert y.shape == (12,) and y[10:].tolist() == [0, 0]
    assert audio_ref.pad_or_trim(np.zeros(0, np.float32), 4).shape == (4,)
