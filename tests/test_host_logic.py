"""not gpu: the host side of the boundary — the C-ABI library loads and exports every symbol include/wis_hip.h
declares, container decode (FLAC MD5 known-answer tests), the wis.audio-compatible host logic, the weight
arena, and loud failure without a GPU."""
import hashlib
import io
import math
import json
import os
import re
import struct
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(lib):
    from wis_hip import _lib
    hdr = open(os.path.join(ROOT, "include", "wis_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wis_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.wis_version() == 1


def test_library_exports_only_the_c_abi_and_passes_the_isa_lint(lib):
    """(round-4 review items 2b / 8) nothing but the wis_* C symbols leaves libwis_hip.so (csrc/exports.map: no C++ helper, no kernel stub
    can collide with another HIP library in the process), and the built code objects obey the packed-f32 / scratch rule that build()
    enforces (tools/isa_lint.py) - checked here on the artefacts the tests run on."""
    import glob
    import importlib.util
    import subprocess
    from wis_hip import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and all(n.startswith("wis_") for n in names), [n for n in names if not n.startswith("wis_")][:5]
    assert set(names) == {name for name, _, _ in _lib.SYMBOLS}
    spec = importlib.util.spec_from_file_location("wis_isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    objs = sorted(glob.glob(os.path.join(ROOT, "willow-inference-server_amd", "build", "*.hip.o")))
    assert len(objs) == 4
    bad, rows = mod.lint(objs)
    assert len(rows) > 150 and not bad and not [r for r in rows if r[6]]
    assert sum(1 for r in rows if r[4]) > 100          # (the scan really saw the MFMA kernels)


@pytest.mark.parametrize("clip,md5,n", [("3sec", "ad790df21d4d9d223d3f34227b5cfedd", 61440),
                                        ("10sec", "c5b99673d012d9a8f5d19dd68874a121", 171008),
                                        ("30sec", "3a541ad6463fe6e884e5995212b518fa", 467968)])
def test_flac_decode_known_answer(golden_dir, clip, md5, n):
    """STREAMINFO MD5 of the decoded PCM = the reference clips' signature."""
    from wis_hip import audio
    pcm, sr = audio.load_audio(os.path.join(golden_dir, "clips", clip + ".flac"))
    assert sr == 16000 and pcm.shape == (n,) and pcm.dtype == np.float32
    i16 = np.round(pcm * 32768.0).astype("<i2")
    assert hashlib.md5(i16.tobytes()).hexdigest() == md5
    meta = json.load(open(os.path.join(golden_dir, "chunker_lcs.json")))["clips"][clip]
    assert i16[:8].tolist() == meta["first8"]
    # file-like and bytes inputs (reference hands do_whisper a BytesIO, main.py:1297-1310)
    raw = open(os.path.join(golden_dir, "clips", clip + ".flac"), "rb").read()
    assert np.array_equal(audio.load_audio(io.BytesIO(raw))[0], pcm)
    assert np.array_equal(audio.load_audio(raw)[0], pcm)


def test_flac_corruption_is_detected(golden_dir):
    from wis_hip import _lib, audio
    raw = bytearray(open(os.path.join(golden_dir, "clips", "3sec.flac"), "rb").read())
    raw[len(raw) // 2] ^= 0x40
    with pytest.raises(_lib.WisError):
        audio.load_audio(bytes(raw))
    with pytest.raises(_lib.WisError):
        audio.load_audio(b"not audio at all")


def test_wav_decode_roundtrip():
    """PCM-in-WAV as write_stream_wav produces it (reference main.py:98-105): mono / stereo, 16-bit."""
    from wis_hip import audio
    rng = np.random.default_rng(0)
    x = rng.integers(-20000, 20000, size=(1600, 2)).astype("<i2")
    for ch in (1, 2):
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(ch); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes(x[:, :ch].tobytes())
        pcm, sr = audio.load_audio(buf.getvalue())
        exp = (x[:, :ch].astype(np.float32) / 32768.0).mean(axis=1)
        assert sr == 16000 and np.allclose(pcm, exp, atol=1e-7)
    # any other rate is resampled to 16 kHz, as librosa.load(sr=16000) does (reference main.py:579)
    for sr_in in (8000, 44100, 48000):
        t = np.arange(sr_in) / sr_in
        tone = (0.4 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr_in); w.writeframes((tone * 32767).astype("<i2").tobytes())
        pcm, sr = audio.load_audio(buf.getvalue())
        assert sr == 16000 and pcm.shape[0] == 16000 and pcm.dtype == np.float32
        exp = 0.4 * np.sin(2 * np.pi * 1000.0 * np.arange(16000) / 16000)
        assert np.abs(pcm[500:-500] - exp[500:-500]).max() < 1e-3          # 16-bit quantisation of the input dominates


def test_resampler_against_analytic_and_scipy():
    import scipy.signal as ss
    from wis_hip import audio
    for sr_in in (8000, 11025, 22050, 32000, 44100, 48000):
        n = 2 * sr_in
        t = np.arange(n) / sr_in
        x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t + 0.3)).astype(np.float32)
        y = audio.resample(x, sr_in, 16000)
        assert y.shape[0] == -(-n * 16000 // sr_in) and y.dtype == np.float32
        to = np.arange(y.shape[0]) / 16000
        exp = 0.5 * np.sin(2 * np.pi * 440 * to) + 0.2 * np.sin(2 * np.pi * 3000 * to + 0.3)
        assert np.abs(y[400:-400] - exp[400:-400]).max() < 1e-6, sr_in
        g = math.gcd(sr_in, 16000)
        z = ss.resample_poly(x.astype(np.float64), 16000 // g, sr_in // g)       # an independent polyphase resampler (different filter)
        assert np.abs(y[400:-400] - z[:y.shape[0]][400:-400]).max() < 5e-3, sr_in
    t = np.arange(48000) / 48000
    assert np.sqrt((audio.resample(np.sin(2 * np.pi * 10000 * t), 48000)[400:-400] ** 2).mean()) < 1e-5     # above the new Nyquist: rejected
    assert audio.resample(np.zeros(0, np.float32), 44100).shape == (0,)
    assert np.array_equal(audio.resample(np.arange(5, dtype=np.float32), 16000), np.arange(5, dtype=np.float32))


def test_audio_host_logic_matches_reference_golden(golden_dir):
    from wis_hip import audio
    cases = json.load(open(os.path.join(golden_dir, "chunker_lcs.json")))
    for n, strides in cases["chunk_iter"].items():
        assert [list(s) for _, s in audio.chunk_iter(np.zeros(int(n), np.float32))] == strides

    class Tok:
        all_special_ids = cases["lcs"][0]["special"]
    for c in cases["lcs"]:
        if isinstance(c["out"], str):
            continue
        seqs = [(s, (1, 0, 0)) for s in c["seqs"]]
        assert audio.find_longest_common_sequence(seqs, Tok).tolist() == c["out"]
    x = np.arange(5, dtype=np.float32)
    assert audio.pad_or_trim(x, 3).tolist() == [0, 1, 2] and audio.pad_or_trim(x, 7).shape == (7,)
    assert audio.N_SAMPLES == 480000 and audio.N_FRAMES == 3000 and audio.chunk_len == 352000


def test_weight_arena_and_layout():
    from wis_hip import weights as W
    w = W.synthetic_weights("tiny", seed=7)
    w2 = W.synthetic_weights("tiny", seed=7, threads=1)
    assert all(np.array_equal(w[k], w2[k]) for k in w)                  # thread-count independent
    arena, index = W.build_arena(w)
    idx2, total = W.synthetic_layout("tiny")
    assert idx2 == index and total == arena.nbytes
    e = next(i for i in index if i["name"] == "decoder/layer_3/attention/linear_1/bias")
    v = arena[e["offset"]:e["offset"] + 2 * 768].view(np.float16)
    assert np.array_equal(v, w[e["name"]]) and not v[:384].any()        # k_proj has no bias
    a = W.arch("large")
    n = sum(int(np.prod(s)) for s, _ in W.tensor_shapes(a["d_model"], a["n_layers"]).values())
    assert abs(n / 1e6 - 1543.3) < 1.0                                   # large-v2: 1540.8 M params + both position tables
    assert len(W.SUPPRESS_IDS) == 88 and W.SUPPRESS_IDS_BEGIN == [220, 50257] and len(W.LANG_IDS) == 99


def test_ct2_model_bin_reader_roundtrip(tmp_path):
    """model.bin layout per SURVEY Appendix C (binary version 6)."""
    from wis_hip import weights as W

    def wstr(f, s):
        b = s.encode() + b"\0"
        f.write(struct.pack("<H", len(b))); f.write(b)
    vars_ = {"encoder/conv1/bias": np.arange(4, dtype=np.float16), "decoder/embeddings/weight": np.ones((3, 2), np.float32)}
    p = tmp_path / "model.bin"
    with open(p, "wb") as f:
        f.write(struct.pack("<I", 6)); wstr(f, "WhisperSpec"); f.write(struct.pack("<I", 3)); f.write(struct.pack("<I", len(vars_)))
        for name, v in vars_.items():
            wstr(f, name); f.write(struct.pack("<B", v.ndim)); f.write(struct.pack("<" + "I" * v.ndim, *v.shape))
            f.write(struct.pack("<B", 4 if v.dtype == np.float16 else 0)); f.write(struct.pack("<I", v.nbytes)); f.write(v.tobytes())
        f.write(struct.pack("<I", 1)); wstr(f, "decoder/projection/weight"); wstr(f, "decoder/embeddings/weight")
    out = W.read_ct2_model_bin(str(p))
    assert np.array_equal(out["encoder/conv1/bias"], vars_["encoder/conv1/bias"])
    assert np.array_equal(out["decoder/projection/weight"], vars_["decoder/embeddings/weight"])


def test_product_path_fails_loudly_without_gpu(lib):
    """No CPU fallback: compute entry points must raise when no HIP device is visible."""
    from wis_hip import _lib, audio, ctranslate2 as ct2
    if lib.wis_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.WisError):
        audio.log_mel_spectrogram(np.zeros(480000, np.float32))
    with pytest.raises(_lib.WisError):
        ct2.Whisper("synthetic:tiny")
    with pytest.raises(ValueError):
        ct2.Whisper("synthetic:tiny", device="cpu")


def test_no_oracle_import_in_product_code():
    """The product package must never import the oracle (or the reference)."""
    pkg = os.path.join(ROOT, "willow-inference-server_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".c", ".h")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "/root/reference" not in src, fn


def test_all_special_ids_come_from_the_checkpoint_tokenizer(tmp_path):
    """reference wis/audio.py:141-146 strips `tokenizer.all_special_ids` (HF: the added tokens flagged special) before the LCS
    merge - for Whisper that list holds <|endoftext|> .. <|notimestamps|> but NOT the <|0.00|>.. timestamp tokens.  With a
    tokenizer.json present the list must come from it; without one (synthetic weights) everything >= <|endoftext|> is special."""
    from tokenizers import AddedToken, Tokenizer, models as tk_models
    from wis_hip import audio, weights as W
    from wis_hip.whisper import _Tokenizer
    tok = Tokenizer(tk_models.WordLevel({f"w{i}": i for i in range(W.EOT)}, unk_token="w0"))
    specials = ["<|endoftext|>", "<|startoftranscript|>"] + [f"<|l{i}|>" for i in range(99)] + ["<|translate|>", "<|transcribe|>", "<|startoflm|>",
                "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    tok.add_special_tokens([AddedToken(t, special=True) for t in specials])                       # ids 50257 .. 50363
    tok.add_tokens([AddedToken(f"<|{i * 0.02:.2f}|>", special=False) for i in range(1501)])       # timestamps: added, not special
    tok.save(str(tmp_path / "tokenizer.json"))
    t = _Tokenizer(str(tmp_path))
    assert t.all_special_ids == list(range(W.EOT, W.NO_TIMESTAMPS + 1))
    assert _Tokenizer(None).all_special_ids == list(range(W.EOT, W.N_VOCAB))
    # a timestamp id survives the stitch with the checkpoint's list, and is dropped by the id-only fallback
    ts = W.NO_TIMESTAMPS + 5
    seqs = [([10, 11, ts, 12, 13], (0, 0, 0)), ([12, 13, 14, W.EOT], (0, 0, 0))]
    assert list(audio.find_longest_common_sequence(seqs, t)) == [10, 11, ts, 12, 13, 14]
    assert list(audio.find_longest_common_sequence(seqs, _Tokenizer(None))) == [10, 11, 12, 13, 14]


def test_isa_lint_scalar_load_rule():
    """tools/isa_lint.py rule 2 (round 6): a destination SGPR of a scalar load named before the s_waitcnt that retires it is a build error -
    the hand-issued `s_load_dword` of csrc/common.hpp uniform_load_issue_* is only safe while hipcc keeps its hands off that register."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wis_isa_lint_t", os.path.join(ROOT, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ok = """
0000000000001000 <k_ok>:
	s_load_dword s20, s[4:5], 0x0                              // 000000001000: C0020502
	v_mov_b32_e32 v1, s7                                       // 000000001008: 7E020207
	s_waitcnt lgkmcnt(0)                                       // 00000000100C: BF8CC07F
	v_mov_b32_e32 v2, s20                                      // 000000001010: 7E040214
	s_endpgm                                                   // 000000001014: BF810000
"""
    bad = ok.replace("v_mov_b32_e32 v1, s7 ", "v_mov_b32_e32 v1, s20").replace("<k_ok>", "<k_bad>")
    joined = """
0000000000002000 <k_join>:
	s_load_dword s20, s[4:5], 0x0                              // 000000002000: C0020502
	s_cbranch_scc1 2                                           // 000000002008: BF850002 <k_join+0x14>
	s_waitcnt lgkmcnt(0)                                       // 00000000200C: BF8CC07F
	s_mov_b32 s21, s20                                         // 000000002010: BE950014
	s_mov_b32 s22, s20                                         // 000000002014: BE960014
	s_endpgm                                                   // 000000002018: BF810000
"""
    assert mod.smem_hazards_text(ok) == []
    hz = mod.smem_hazards_text(bad)
    assert len(hz) == 1 and hz[0][0] == "k_bad" and "s20" in hz[0][2]
    assert mod.smem_hazards_text(joined) == []          # (basic blocks only: what is pending at a join is not known)
    spill = ok.replace("v_mov_b32_e32 v1, s7 ", "v_writelane_b32 v9, s20, 3")
    assert len(mod.smem_hazards_text(spill)) == 1       # an SGPR spill of the in-flight register is the advisor's scenario
