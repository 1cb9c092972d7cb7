"""Generate the golden fixtures under tests/golden/ by running the REAL reference code.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What it pins:
  * clips/*.flac            — the reference's own test clips (client/*.flac), copied as input fixtures
                              (data, not code); their STREAMINFO MD5 is the decoder's known-answer test.
  * logmel_{3,10,30}sec.npz — `wis.audio.log_mel_spectrogram(pad_or_trim(pcm))` of the REAL reference
                              module (imported from /root/reference) on each clip, float16-packed residual
                              free: stored as float32, zlib-compressed.
  * logmel_noise.npz        — the same on seeded noise (SURVEY §8d synthetic audio: default_rng(1234), 0.1*N(0,1),
                              drawn in the order 61440 then 480000 samples) incl. a full 30 s window; inputs are
                              regenerated from the seed by the tests.
  * mel_filters.npz         — the reference asset wis/assets/mel_filters.npz (80x201 float32).
  * chunker_lcs.json        — `chunk_iter` strides for several lengths and `find_longest_common_sequence` cases (incl. first
                              windows of 0/1/2 tokens and windows shorter than the overlap), evaluated by the reference's own
                              code under its pinned numpy's comparison semantics (np123_shim.py).  `--lcs-only` rewrites just
                              this file.
The decoder used to read the FLAC clips here is this repo's own C decoder, verified by MD5.
"""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "willow-inference-server_amd"))


def decode(path):
    so = "/tmp/_wis_audio_io.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "willow-inference-server_amd", "csrc", "audio_io.c"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.wis_audio_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_float)),
                                     ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    b = open(path, "rb").read()
    p, n, sr, md = ctypes.POINTER(ctypes.c_float)(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    rc = lib.wis_audio_decode(b, len(b), ctypes.byref(p), ctypes.byref(n), ctypes.byref(sr), ctypes.byref(md))
    assert rc == 0 and md.value == 1 and sr.value == 16000, (rc, md.value, sr.value)
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def main():
    from wis.audio import chunk_iter, find_longest_common_sequence, log_mel_spectrogram, pad_or_trim  # the reference
    sys.path.insert(0, HERE)
    lcs_only = "--lcs-only" in sys.argv      # regenerate chunker_lcs.json only (the npz fixtures stay byte-identical)

    os.makedirs(os.path.join(HERE, "clips"), exist_ok=True)
    meta = {}
    for clip in ("3sec", "10sec", "30sec"):
        src = os.path.join(REF, "client", clip + ".flac")
        shutil.copyfile(src, os.path.join(HERE, "clips", clip + ".flac"))
        pcm = decode(src)
        i16 = np.round(pcm * 32768.0).astype("<i2")
        meta[clip] = dict(samples=int(pcm.shape[0]), pcm_md5=hashlib.md5(i16.tobytes()).hexdigest(),
                          first8=[int(v) for v in i16[:8]])
        mel = log_mel_spectrogram(pad_or_trim(pcm)).numpy()
        assert mel.shape == (80, 3000) and mel.dtype == np.float32
        if not lcs_only:
            np.savez_compressed(os.path.join(HERE, f"logmel_{clip}.npz"), mel=mel)
        meta[clip]["mel_sum"] = float(mel.astype(np.float64).sum())
        print(clip, meta[clip])
    # seeded noise
    rng = np.random.default_rng(1234)
    out = {}
    for n in (() if lcs_only else (61440, 480000)):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        out[f"mel_{n}"] = log_mel_spectrogram(pad_or_trim(x)).numpy()
    if not lcs_only:
        np.savez_compressed(os.path.join(HERE, "logmel_noise.npz"), **out)
    shutil.copyfile(os.path.join(REF, "wis", "assets", "mel_filters.npz"), os.path.join(HERE, "mel_filters.npz"))

    # chunker + LCS cases from the real reference
    cases = {"chunk_iter": {}, "lcs": []}
    for n in (100, 352000, 480001, 700000, 2880000):
        cases["chunk_iter"][str(n)] = [list(map(int, s)) for _, s in chunk_iter(np.zeros(n, np.float32))]

    class Tok:
        all_special_ids = [50257, 50258, 50259, 50359, 50363]
    from np123_shim import numpy_1_23_semantics
    import wis.audio as ref_audio

    def ref_lcs(lists):
        seqs = [(s, (1, 0, 0)) for s in lists]
        with numpy_1_23_semantics(ref_audio):       # the reference's pinned numpy (see np123_shim.py)
            return [int(v) for v in find_longest_common_sequence(seqs, Tok)]

    rr = np.random.default_rng(7)
    lcs_lists = []
    for trial in range(12):
        base = [int(v) for v in rr.integers(0, 50, size=60)]
        a = base[:35] + [50257]
        b = [50258] + base[25 - (trial % 4):55]
        c = base[48:60]
        lcs_lists.append([a, b, c])
    # first windows of 0, 1 and 2 tokens (near-silent opening window), windows shorter than the overlap, specials only
    lcs_lists += [
        [[], [3, 2, 2, 3, 0]], [[50257], [1, 1, 2]], [[2], [3, 2, 2, 3, 0]], [[2], [2, 2]], [[2], [2]], [[7], [1, 7, 7, 7, 2]],
        [[2, 3], [2, 3, 4, 5]], [[2, 3], [3, 2, 3, 9]], [[5, 5], [5, 5, 5, 5, 5]], [[1, 2, 3, 4, 5, 6], [5, 6]],
        [[1, 2, 3, 4, 5, 6], [6]], [[1, 2, 3, 4, 5, 6], []], [[9], [9, 9, 9], [9, 9]], [[], [], [4, 4]], [[4], [], [4, 4, 1]],
        [[50258, 8, 50257], [50258, 8, 8, 8, 50257], [8, 8, 3]],
    ]
    for n_first in (0, 1, 2, 3):
        for trial in range(20):
            k = int(rr.integers(1, 4))
            lists = [[int(v) for v in rr.integers(0, 3, size=n_first)]]
            for _ in range(k):
                lists.append([int(v) for v in rr.integers(0, 3, size=int(rr.integers(0, 7)))])
            lcs_lists.append(lists)
    for lists in lcs_lists:
        cases["lcs"].append(dict(seqs=lists, special=Tok.all_special_ids, out=ref_lcs(lists)))
    cases["clips"] = meta
    with open(os.path.join(HERE, "chunker_lcs.json"), "w") as f:
        json.dump(cases, f)
    print("lcs outs:", [c["out"] if isinstance(c["out"], str) else len(c["out"]) for c in cases["lcs"]])


if __name__ == "__main__":
    main()
