"""-m gpu: model-level parity of the HIP path (through the C-ABI / the ctranslate2-compatible shim)
against the CPU oracle on the same seeded synthetic weights (no checkpoint exists offline).

Tolerances (written here as the test bar):
  * encoder output rel-L2 <= 2e-3 (f16 MFMA path vs fp32 oracle)
  * teacher-forced logits: max abs error <= 5e-2 and rel-L2 <= 5e-3
  * greedy / beam token ids identical wherever the oracle's per-step DECISION margin (oracle/whisper_ref.py generate: the
    k / k+1 survival boundary, every adjacent gap of the top-(2k+1) candidates on hypothesis-finishing steps, the final
    ranking; greedy: top-1 / top-2) exceeds MARGIN; otherwise the returned hypothesis score must agree within 1e-2.
"""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so: one HIP runtime per process)

pytestmark = pytest.mark.gpu
MARGIN = 0.02
PROMPT = [50258, 50259, 50359, 50363]


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def mels(golden_dir):
    return np.stack([np.load(os.path.join(golden_dir, f"logmel_{c}.npz"))["mel"] for c in ("3sec", "10sec")]).astype(np.float32)


def _make(size, **kw):
    from oracle.whisper_ref import WhisperRef
    from wis_hip import ctranslate2 as ct2, weights as W
    w = W.synthetic_weights(size, seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    a = W.arch(size)
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=4, max_beam=5, **kw)
    ref = WhisperRef(w, a["d_model"], a["n_layers"], a["n_heads"])
    return model, ref, w, a


@pytest.fixture(scope="module")
def tiny():
    return _make("tiny")


@pytest.fixture(scope="module")
def base():
    return _make("base")


def _handle(model):
    return model._replicas[0].handle


@pytest.mark.parametrize("which", ["tiny", "base"])
def test_encoder_parity(which, request, mels, lib):
    import ctypes as C
    from wis_hip import _lib
    model, ref, w, a = request.getfixturevalue(which)
    B = 2
    out = np.zeros((B, 1500, a["d_model"]), np.float32)
    _lib.check(lib.wis_debug_encode(_handle(model), _lib.ptr(mels), _lib.WIS_IN_MEL_HOST, B, out.ctypes.data_as(C.POINTER(C.c_float))))
    exp = ref.encode(mels).numpy()
    e = _relerr(out, exp)
    print(f"encoder {which}: rel-L2 {e:.3e}, max abs {np.abs(out - exp).max():.3e}")
    assert e <= 2e-3


def test_encoder_parity_eight_utterances(tiny, mels, lib):
    """The batched encoder takes other code paths than one or two utterances (256x256 GEMM tiles with the early staging order, the
    unsplit FFN2, attention workgroup form by grid size): eight different feature windows against the oracle."""
    import ctypes as C
    from wis_hip import _lib, ctranslate2 as ct2
    _, ref, w, a = tiny
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=8, max_beam=5)      # the module fixture holds 4 utterances
    B = 8
    m8 = np.ascontiguousarray(np.stack([np.roll(mels[i % 2], 53 * i, axis=-1) for i in range(B)]))
    out = np.zeros((B, 1500, a["d_model"]), np.float32)
    _lib.check(lib.wis_debug_encode(_handle(model), _lib.ptr(m8), _lib.WIS_IN_MEL_HOST, B, out.ctypes.data_as(C.POINTER(C.c_float))))
    exp = ref.encode(m8).numpy()
    e = max(_relerr(out[i], exp[i]) for i in range(B))
    print(f"encoder tiny, 8 utterances: worst rel-L2 {e:.3e}, max abs {np.abs(out - exp).max():.3e}")
    assert e <= 2e-3


@pytest.mark.parametrize("which", ["tiny", "base"])
def test_teacher_forced_logits(which, request, mels, lib):
    import ctypes as C
    from wis_hip import _lib
    model, ref, w, a = request.getfixturevalue(which)
    B, T = 2, 7
    rng = np.random.default_rng(3)
    dec_in = np.concatenate([np.tile(np.array(PROMPT, np.int32), (B, 1)), rng.integers(0, 50000, size=(B, T - 4)).astype(np.int32)], axis=1)
    dec_in = np.ascontiguousarray(dec_in)
    out = np.zeros((B, T, a["n_vocab"]), np.float32)
    _lib.check(lib.wis_debug_logits(_handle(model), _lib.ptr(mels), _lib.WIS_IN_MEL_HOST, B, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T,
                                    out.ctypes.data_as(C.POINTER(C.c_float))))
    mem = ref.encode(mels)
    exp = ref.decode_logits(dec_in, mem).numpy()
    e, mx = _relerr(out, exp), np.abs(out - exp).max()
    print(f"logits {which}: rel-L2 {e:.3e}, max abs {mx:.3e}, logit std {exp.std():.3f}")
    assert mx <= 5e-2 and e <= 5e-3


def _check_generate(model, ref, mel, beam, fixed_new, max_new=0):
    """-> number of utterances whose ids are IDENTICAL to the oracle's (asserted wherever the margin rule forces them)."""
    from wis_hip import ctranslate2 as ct2, weights as W
    feats = ct2.StorageView.from_array(mel)
    kw = dict(beam_size=beam, fixed_new_tokens=fixed_new)
    res = model.generate(feats, [PROMPT] * mel.shape[0], **kw)
    n_exact = 0
    for b in range(mel.shape[0]):
        ids, score, trace = ref.generate(mel[b], PROMPT, beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN,
                                         fixed_new=fixed_new, max_new_tokens=max_new, return_trace=True)
        got, gscore = res[b].sequences_ids[0], res[b].scores[0]
        margin = min(trace) if trace else 1.0
        print(f"  utt {b} beam {beam}: oracle len {len(ids)} score {score:.5f} decision margin {margin:.4f} (all-gaps {min(ref.last_trace_full):.4f}) | "
              f"hip len {len(got)} score {gscore:.5f} | identical {got == ids}")
        if margin > MARGIN or min(ref.last_trace_full) > MARGIN:
            assert got == ids, (got, ids)          # every decision of the oracle's search is forced: ids must be identical
        n_exact += got == ids
        assert abs(gscore - score) <= 1e-2
        assert all(0 <= t < 51865 for t in got) and W.EOT not in got
    return n_exact


def test_generate_greedy_fixed(tiny, mels):
    model, ref, w, a = tiny
    assert _check_generate(model, ref, mels, 1, 8) == 2          # observed on MI355X: both identical (margins 0.14 / 0.19 force them)


def test_generate_beam5_fixed(tiny, mels):
    model, ref, w, a = tiny
    assert _check_generate(model, ref, mels, 5, 8) >= 1          # observed on MI355X: 2 of 2; the oracle's decision margins here are 0.005 / 0.0009, so one may flip


def _oracle_rescore(ref, memory, ids, fixed_new, suppress_blank=True):
    """Mean log-prob the ORACLE assigns to `ids` (teacher-forced over the prompt + ids, logits processors applied per step):
    what CT2 would report as the score of that hypothesis (length_penalty 1)."""
    import torch
    from wis_hip import weights as W
    lg = ref.decode_logits(np.array([PROMPT + list(ids)[:-1]]) if len(ids) else np.array([PROMPT]), torch.as_tensor(memory)[None])[0]
    total = 0.0
    for t, tok in enumerate(ids):
        row = ref.apply_processors(lg[len(PROMPT) - 1 + t][None].double(), t, W.SUPPRESS_IDS, W.SUPPRESS_IDS_BEGIN, suppress_blank, fixed_new)
        total += float(torch.log_softmax(row, dim=-1)[0, tok])
    return total / max(len(ids), 1)


@pytest.mark.parametrize("beam,fixed_new", [(1, 100), (1, 0), (5, 100), (3, 150), (5, 0)])
def test_generate_long_histories(tiny, mels, beam, fixed_new):
    """Decodes far beyond 64 cache positions - 100 / 150 fixed tokens and the natural max_new = 224 run, the reference's own
    limit (main.py:687-692 defaults) - against the ORACLE: the self-attention kernel's online-softmax continuation and
    kv_reorder over long histories.
      * ids identical whenever every decision of the oracle's search is forced (decision margin > MARGIN; the 100-token greedy
        run is), and ALSO whenever the stricter all-adjacent-gaps margin of the round-1 review holds;
      * always: the score the engine reports for ITS ids equals the oracle's teacher-forced score of those same ids (a wrong
        cache row / beam reorder / long-history softmax would show here even when a near-tie made the searches diverge);
      * always: under the oracle's model the engine's hypothesis is not worse than the oracle's by more than 1e-2."""
    from wis_hip import ctranslate2 as ct2, weights as W
    model, ref, w, a = tiny
    feats = ct2.StorageView.from_array(mels[:1])
    res = model.generate(feats, [PROMPT], beam_size=beam, fixed_new_tokens=fixed_new)[0]
    memory = ref.encode(mels[:1])[0].numpy()
    ids, score, trace = ref.generate(None, PROMPT, beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN,
                                     fixed_new=fixed_new, memory=memory, return_trace=True)
    got, gscore = res.sequences_ids[0], res.scores[0]
    rescored = _oracle_rescore(ref, memory, got, fixed_new)
    n_same = next((i for i, (x, y) in enumerate(zip(got, ids)) if x != y), min(len(got), len(ids)))
    print(f"long decode beam {beam} fixed {fixed_new}: oracle len {len(ids)} score {score:.5f} decision margin {min(trace):.5f} "
          f"(all-gaps {min(ref.last_trace_full):.5f}) | hip len {len(got)} score {gscore:.5f}, oracle rescoring of the hip ids {rescored:.5f} | "
          f"common prefix {n_same}, identical {got == ids}")
    assert len(ids) >= (fixed_new if fixed_new else 200) and len(got) >= (fixed_new if fixed_new else 200)
    assert all(0 <= t < 51865 for t in got) and W.EOT not in got
    assert abs(gscore - rescored) <= 3e-3, (gscore, rescored)
    assert rescored >= score - 1e-2, (rescored, score)
    if min(trace) > MARGIN or min(ref.last_trace_full) > MARGIN:
        assert got == ids, (n_same, got[n_same:n_same + 3], ids[n_same:n_same + 3])
    if beam == 1 and fixed_new == 100:
        assert min(trace) > MARGIN and got == ids          # the forced case must stay forced (seeded weights, CPU oracle)


def test_teacher_forced_logits_full_context(tiny, mels, lib):
    """Teacher-forced logits over the WHOLE text context (448 positions, one utterance): every history length the decoder
    self-attention can see, vs the oracle (bar: max abs 5e-2, rel-L2 5e-3)."""
    import ctypes as C
    from wis_hip import _lib
    model, ref, w, a = tiny
    T = 448
    rng = np.random.default_rng(11)
    dec_in = np.ascontiguousarray(np.concatenate([np.array(PROMPT, np.int32), rng.integers(0, 50000, size=T - 4).astype(np.int32)])[None])
    out = np.zeros((1, T, a["n_vocab"]), np.float32)
    _lib.check(lib.wis_debug_logits(_handle(model), _lib.ptr(mels[:1]), _lib.WIS_IN_MEL_HOST, 1, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T,
                                    out.ctypes.data_as(C.POINTER(C.c_float))))
    exp = ref.decode_logits(dec_in, ref.encode(mels[:1])).numpy()
    worst = max(range(T), key=lambda t: np.abs(out[0, t] - exp[0, t]).max())
    e, mx = _relerr(out, exp), np.abs(out - exp).max()
    print(f"logits over 448 positions: rel-L2 {e:.3e}, max abs {mx:.3e} (at position {worst}); positions >= 64: max abs {np.abs(out[0, 64:] - exp[0, 64:]).max():.3e}")
    assert mx <= 5e-2 and e <= 5e-3


def test_generate_beam3_base(base, mels):
    model, ref, w, a = base
    _check_generate(model, ref, mels[:1], 3, 6)


def test_generate_natural_termination(tiny, mels):
    """No measurement convention: the run ends on the max-length step (random weights rarely emit EOT);
    exercises is_last handling and the hypothesis bookkeeping."""
    from wis_hip import ctranslate2 as ct2, weights as W
    model, ref, w, a = tiny
    feats = ct2.StorageView.from_array(mels[:1])
    res = model.generate(feats, [PROMPT], beam_size=5, max_length=2 * 10)     # max_new = min(10, 20 - 4) = 10
    ids, score, trace = ref.generate(mels[0], PROMPT, beam_size=5, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN,
                                     max_new_tokens=10, return_trace=True)
    got = res[0].sequences_ids[0]
    print(f"natural: oracle {ids} ({score:.5f}) hip {got} ({res[0].scores[0]:.5f}) margin {min(trace):.4f}")
    assert len(got) <= 10
    if min(trace) > MARGIN:
        assert got == ids
    assert abs(res[0].scores[0] - score) <= 1e-2


def test_batch_invariance_and_determinism(tiny, mels):
    from wis_hip import ctranslate2 as ct2
    model, ref, w, a = tiny
    m3 = np.ascontiguousarray(np.stack([mels[0], mels[1], mels[0]]))
    r1 = model.generate(ct2.StorageView.from_array(m3), [PROMPT] * 3, beam_size=5, fixed_new_tokens=6)
    r2 = model.generate(ct2.StorageView.from_array(m3), [PROMPT] * 3, beam_size=5, fixed_new_tokens=6)
    single = model.generate(ct2.StorageView.from_array(mels[:1]), [PROMPT], beam_size=5, fixed_new_tokens=6)
    assert [r.sequences_ids for r in r1] == [r.sequences_ids for r in r2]          # bit-deterministic replay
    assert r1[0].sequences_ids == r1[2].sequences_ids == single[0].sequences_ids    # batch composition does not change an utterance


def test_batch_of_eight_matches_single_utterances(tiny, mels):
    """40 decoder rows (the fragment-image path at three 16-row blocks) and the batched encoder against one-utterance calls: the
    length-normalised beam scores agree to rounding and the ids are those of the single calls wherever the oracle-free margin
    allows (seeded weights: a flip needs a near tie; at most one utterance of eight may differ)."""
    from wis_hip import ctranslate2 as ct2
    _, ref, w, a = tiny
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=8, max_beam=5)
    m8 = np.ascontiguousarray(np.stack([np.roll(mels[i % 2], 53 * i, axis=-1) for i in range(8)]))
    r8 = model.generate(ct2.StorageView.from_array(m8), [PROMPT] * 8, beam_size=5, fixed_new_tokens=8)
    r8b = model.generate(ct2.StorageView.from_array(m8), [PROMPT] * 8, beam_size=5, fixed_new_tokens=8)
    assert [r.sequences_ids for r in r8] == [r.sequences_ids for r in r8b]
    differ = 0
    for i in range(8):
        one = model.generate(ct2.StorageView.from_array(m8[i:i + 1]), [PROMPT], beam_size=5, fixed_new_tokens=8)[0]
        same = one.sequences_ids == r8[i].sequences_ids
        differ += not same
        if same:
            assert abs(one.scores[0] - r8[i].scores[0]) <= 2e-3, (i, one.scores, r8[i].scores)
    print(f"batch of eight vs single calls: {differ} of 8 utterances differ")
    assert differ <= 1


def test_detect_language(tiny, mels):
    from wis_hip import ctranslate2 as ct2, weights as W
    model, ref, w, a = tiny
    out = model.detect_language(ct2.StorageView.from_array(mels[:1]))
    exp = ref.detect_language(mels[0], W.LANG_IDS)
    probs = {k: v for k, v in out[0]}
    from wis_hip.languages import LANGUAGE_CODES
    got = np.array([probs[f"<|{c}|>"] for c in LANGUAGE_CODES])
    print(f"detect_language: max abs prob err {np.abs(got - exp).max():.3e}, top {out[0][0]}")
    assert abs(sum(probs.values()) - 1.0) < 1e-3
    assert np.abs(got - exp).max() < 2e-3
    assert out[0][0][1] == max(probs.values())


def test_pcm_input_matches_mel_input(tiny, golden_dir, lib):
    """WIS_IN_PCM_HOST (log-mel fused on device) must give the same tokens as the mel boundary."""
    from wis_hip import _lib, audio, ctranslate2 as ct2
    model, ref, w, a = tiny
    pcm, _ = audio.load_audio(os.path.join(golden_dir, "clips", "3sec.flac"))
    x = np.ascontiguousarray(audio.pad_or_trim(pcm)[None])
    mel = audio.log_mel_spectrogram(x[0]).numpy()[None]
    r_mel = model.generate(ct2.StorageView.from_array(np.ascontiguousarray(mel)), [PROMPT], beam_size=5, fixed_new_tokens=6)
    r_pcm = model._generate_chunk(model._replicas[0], x, [PROMPT], 4, 5, 224, 1.0, 1.0, True, True, 6, _lib.WIS_IN_PCM_HOST)
    assert r_mel[0].sequences_ids == r_pcm[0].sequences_ids


def test_do_whisper_orchestrator(golden_dir):
    """The do_whisper mirror (reference main.py:554-770): 6-tuple result, per-request model/beam selection, the long-audio
    beam switch and the > 30 s chunking + LCS merge path (3 windows for 40 s)."""
    from wis_hip import audio, whisper
    from wis_hip.settings import APISettings
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch = 4
    models = whisper.WhisperModels(settings=s, device_index=[0])
    clip = os.path.join(golden_dir, "clips", "3sec.flac")
    res = whisper.do_whisper(clip, "tiny", 5, "transcribe", False, "en", models=models, fixed_new_tokens=6)
    language, text, infer_ms, translation, speedup, duration = res
    assert language == "en" and duration == 3840 and translation is None and infer_ms > 0 and speedup == int(3840 // infer_ms)
    assert len(res.tokens) == 6 and text == " ".join(str(t) for t in res.tokens)
    again = whisper.do_whisper(open(clip, "rb").read(), "tiny", 5, models=models, fixed_new_tokens=6)       # bytes input, same model handle
    assert again.tokens == res.tokens
    g = whisper.do_whisper(clip, "tiny", 1, models=models, fixed_new_tokens=6)                              # per-request beam selection
    assert len(g.tokens) == 6
    with pytest.raises(ValueError):
        whisper.do_whisper(clip, "tiny", 1, force_language="xx", models=models)
    # 40 s of seeded noise: long mode (beam := long_beam_size) and chunking into 3 windows, merged by LCS
    rng = np.random.default_rng(0)
    long_pcm = (0.05 * rng.standard_normal(40 * 16000)).astype(np.float32)
    assert len(list(audio.chunk_iter(long_pcm))) == 3
    out = whisper.do_whisper(long_pcm, "tiny", 5, models=models, fixed_new_tokens=5)
    assert out[5] == 40000 and 1 <= len(out.tokens) <= 15
    det = whisper.do_whisper(clip, "tiny", 1, detect_language=True, models=models, fixed_new_tokens=4)
    assert det[0] in __import__("wis_hip.languages", fromlist=["LANGUAGES"]).LANGUAGES


def test_model_from_device_resident_arena(mels):
    """Multi-GPU load path: the weight arena arrives in DEVICE memory (the buffer an RCCL broadcast filled, here a torch CUDA
    tensor) and is handed to wis_model_create(arena_on_device=1) by raw pointer; results must equal the host-arena model."""
    import torch
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    arena, index = W.build_arena(w)
    buf = torch.from_numpy(arena).to("cuda:0")
    torch.cuda.synchronize()
    h = ct2.create_handle(a, None, index, 0, max_batch=2, max_beam=5, arena_device_ptr=(buf.data_ptr(), arena.nbytes))
    del buf
    torch.cuda.empty_cache()
    host = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5)
    dev = ct2.Whisper.from_handles([(h, 0)], a, max_batch=2, max_beam=5)
    f = ct2.StorageView.from_array(mels)
    r_host = host.generate(f, [PROMPT] * 2, beam_size=5, fixed_new_tokens=6)
    r_dev = dev.generate(f, [PROMPT] * 2, beam_size=5, fixed_new_tokens=6)
    assert [r.sequences_ids for r in r_host] == [r.sequences_ids for r in r_dev]


def test_int8_float16_compute_type(mels, lib):
    """SURVEY §8(f)4 / reference main.py:242: per-row int8 decoder weights with f16 activations.  The oracle runs on the
    de-quantised weights (wis_hip.weights.quantize_decoder_weights restates the engine's quantiser in numpy), so the bars are
    the f16 ones: the 8-bit path adds no arithmetic error of its own (int8 values are exact in f16, row scales are applied in fp32)."""
    import ctypes as C
    from oracle.whisper_ref import WhisperRef
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=4, max_beam=5, compute_type="int8_float16")
    assert model.compute_type == "int8_float16"
    f16_model = ct2.Whisper("unused", weights=w, arch=a, max_batch=4, max_beam=5)
    assert lib.wis_model_device_bytes(model._replicas[0].handle) < lib.wis_model_device_bytes(f16_model._replicas[0].handle)
    ref = WhisperRef(W.quantize_decoder_weights(w), a["d_model"], a["n_layers"], a["n_heads"])
    B, T = 2, 7
    rng = np.random.default_rng(3)
    dec_in = np.ascontiguousarray(np.concatenate([np.tile(np.array(PROMPT, np.int32), (B, 1)), rng.integers(0, 50000, size=(B, T - 4)).astype(np.int32)], axis=1))
    out = np.zeros((B, T, a["n_vocab"]), np.float32)
    _lib.check(lib.wis_debug_logits(_handle(model), _lib.ptr(mels), _lib.WIS_IN_MEL_HOST, B, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T,
                                    out.ctypes.data_as(C.POINTER(C.c_float))))
    exp = ref.decode_logits(dec_in, ref.encode(mels)).numpy()
    e, mx = _relerr(out, exp), np.abs(out - exp).max()
    print(f"int8_float16 logits: rel-L2 {e:.3e}, max abs {mx:.3e}")
    assert mx <= 5e-2 and e <= 5e-3
    # and it is a DIFFERENT model from the f16 one (the quantisation is really in effect)
    out16 = np.zeros_like(out)
    _lib.check(lib.wis_debug_logits(_handle(f16_model), _lib.ptr(mels), _lib.WIS_IN_MEL_HOST, B, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T,
                                    out16.ctypes.data_as(C.POINTER(C.c_float))))
    assert _relerr(out16, exp) > 3 * e
    assert _check_generate(model, ref, mels, 1, 8) == 2          # observed on MI355X: both identical (margins 0.14 / 0.19 force them)
    assert _check_generate(model, ref, mels, 5, 8) >= 1          # observed on MI355X: 2 of 2; the oracle's decision margins here are 0.005 / 0.0009, so one may flip
    with pytest.raises(ValueError):
        ct2.Whisper("unused", weights=w, arch=a, compute_type="bfloat16")


def test_roofline_tap_runs_at_every_row_count(tiny, lib):
    """bench.py's roofline tap (wis_bench_weight_stream) at single-utterance and batched row counts: above 8 rows the LayerNorm
    runs as its own launch, the tap must take the same route as dec_forward."""
    import ctypes as C
    from wis_hip import _lib
    model, ref, w, a = tiny
    for rows in (1, 5, 8, 20, 40):
        ms, nl, nb = C.c_float(), C.c_int(), C.c_double()
        _lib.check(lib.wis_bench_weight_stream(_handle(model), rows, 1, C.byref(ms), C.byref(nl), C.byref(nb)))
        assert nl.value == 6 * a["n_layers"] + 1 and nb.value > 0 and ms.value > 0


def test_replica_pool_from_one_host_upload(mels):
    """`Whisper(device_index=[...])` (reference main.py:295): ONE host upload, the other replicas are filled device-to-device
    (wis_dev_copy_peer, doubling tree) and created from the device-resident arena.  A 1-GPU box can only list device 0 several
    times - the same code path with the peer copy degenerating to a local one; every replica must return what a lone model returns,
    and concurrent calls must spread over the replicas."""
    import threading
    from wis_hip import ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    lone = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5)
    pool = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5, device_index=[0, 0, 0])
    assert len(pool._replicas) == 3
    f = ct2.StorageView.from_array(mels)
    exp = [r.sequences_ids for r in lone.generate(f, [PROMPT] * 2, beam_size=5, fixed_new_tokens=6)]
    for r in pool._replicas:                      # every replica, addressed directly
        got = pool._generate_chunk(r, mels, [PROMPT] * 2, 4, 5, 224, 1.0, 1.0, True, True, 6, 0)
        assert [x.sequences_ids for x in got] == exp
    out = {}

    def client(i):
        out[i] = pool.generate(ct2.StorageView.from_array(mels[i % 2:i % 2 + 1]), [PROMPT], beam_size=5, fixed_new_tokens=6)[0].sequences_ids

    th = [threading.Thread(target=client, args=(i,)) for i in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    used = {idx for idx, _ in pool._batcher.batches}
    print("replicas used:", sorted(used))
    assert len(used) >= 2
    single = [lone.generate(ct2.StorageView.from_array(mels[k:k + 1]), [PROMPT], beam_size=5, fixed_new_tokens=6)[0].sequences_ids for k in range(2)]
    flips = sum(out[i] != single[i % 2] for i in range(12))
    assert flips <= 2
    pool.close(); lone.close()


def test_generate_refuses_concurrent_entry_on_one_handle(base, mels):
    """the boundary must be safe under threads.  Concurrency comes from replicas and the micro-batcher; a handle
    itself runs one call at a time - a second thread that enters `wis_generate` on a BUSY handle (bypassing the shim's lock) is
    refused with WIS_E_STATE at once instead of corrupting the first call's KV caches, and the first call's result is unharmed."""
    import threading
    import time
    from wis_hip import _lib
    model = base[0]
    r = model._replicas[0]
    x = np.ascontiguousarray(mels[:1])
    want = model._generate_chunk(r, x, [PROMPT], 4, 5, 224, 1.0, 1.0, True, True, 120, 0)[0].sequences_ids
    refused = 0
    for attempt in range(6):
        oks, errs = [], []

        def call():
            try:
                oks.append(model._generate_chunk(r, x, [PROMPT], 4, 5, 224, 1.0, 1.0, True, True, 120, 0)[0].sequences_ids)
            except _lib.WisError as e:
                errs.append((e.code, str(e)))

        a, b = threading.Thread(target=call), threading.Thread(target=call)
        a.start(); time.sleep(0.003); b.start()
        a.join(); b.join()
        assert len(oks) >= 1 and all(o == want for o in oks), "the call that held the handle was disturbed"
        assert all(code == -6 and "another call is running" in msg for code, msg in errs), errs
        refused += len(errs)
    print(f"concurrent entry refused {refused} times in 6 attempts")
    assert refused >= 1
    assert model._generate_chunk(r, x, [PROMPT], 4, 5, 224, 1.0, 1.0, True, True, 120, 0)[0].sequences_ids == want


def test_sixteen_utterances_per_device_batch_vs_oracle(mels, lib):
    """Device batches beyond 48 decoder rows (csrc/kernels.hpp MAX_ROWS = 96: 16 utterances x beam 5 = 80 rows, 5 row blocks of the
    fragment-image skinny GEMM; the merged prompt pass has 64 rows): teacher-forced logits at 64 / 80 / 96 rows per pass and a
    16-utterance beam-5 generate, every utterance against the oracle's answer for ITS features."""
    import ctypes as C
    from oracle.whisper_ref import WhisperRef
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    model = ct2.Whisper("unused", weights=w, arch=a, max_batch=16, max_beam=5)
    ref = WhisperRef(w, a["d_model"], a["n_layers"], a["n_heads"])
    mem = ref.encode(mels)
    rng = np.random.default_rng(9)
    for B, R in ((4, 16), (5, 16), (6, 16)):          # 64, 80, 96 rows per pass
        T = 16
        m = np.ascontiguousarray(mels[[i % 2 for i in range(B)]])
        dec_in = np.ascontiguousarray(np.concatenate([np.tile(np.array(PROMPT, np.int32), (B, 1)), rng.integers(0, 50000, size=(B, T - 4)).astype(np.int32)], axis=1))
        out = np.zeros((B, T, a["n_vocab"]), np.float32)
        _lib.check(lib.wis_debug_logits_rows(_handle(model), _lib.ptr(m), _lib.WIS_IN_MEL_HOST, B, dec_in.ctypes.data_as(C.POINTER(C.c_int32)), T, R,
                                             out.ctypes.data_as(C.POINTER(C.c_float))))
        exp = ref.decode_logits(dec_in, mem[[i % 2 for i in range(B)]]).numpy()
        e, mx = _relerr(out, exp), np.abs(out - exp).max()
        print(f"logits at {B * R} rows per pass: rel-L2 {e:.3e}, max abs {mx:.3e}")
        assert mx <= 5e-2 and e <= 5e-3
    S = 10
    order = [i % 2 for i in range(16)]
    res = model.generate(ct2.StorageView.from_array(np.ascontiguousarray(mels[order])), [PROMPT] * 16, beam_size=5, fixed_new_tokens=S)
    want = {}
    for c in (0, 1):
        ids, score, trace = ref.generate(None, PROMPT, beam_size=5, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN, fixed_new=S,
                                         memory=mem[c].numpy(), return_trace=True)
        want[c] = (ids, score, min(trace))
    exact = 0
    for i, r in enumerate(res):
        ids, score, margin = want[order[i]]
        got, gscore = r.sequences_ids[0], r.scores[0]
        rescored = _oracle_rescore(ref, mem[order[i]].numpy(), got, S)
        assert len(got) == S and abs(gscore - rescored) <= 3e-3 and rescored >= score - 1e-2, (i, gscore, rescored, score)
        if margin > MARGIN:
            assert got == ids, (i, got, ids)
        exact += got == ids
    print(f"16 utterances x beam 5: {exact} of 16 identical to the oracle (decision margins {want[0][2]:.4f} / {want[1][2]:.4f})")
    assert exact >= 8
    model.close()


def test_replicas_per_device_share_one_weight_copy(mels, lib):
    """`inter_threads` (reference main.py:341-355: batches a CTranslate2 model runs in parallel) -> replicas PER GPU that share one
    weight copy (wis_model_clone) and own their stream, activations and KV caches: every replica answers like a lone model,
    concurrent calls spread over them, and the shared weights live until the LAST replica is destroyed (whichever goes first)."""
    import ctypes as C
    import threading
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    w = W.synthetic_weights("tiny", seed=1234, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    lone = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5)
    pool = ct2.Whisper("unused", weights=w, arch=a, max_batch=2, max_beam=5, inter_threads=3)
    assert len(pool._replicas) == 3 and len({r.device for r in pool._replicas}) == 1
    own = [lib.wis_model_device_bytes(r.handle) for r in pool._replicas]
    assert own[1] == own[2] and own[1] < own[0]            # a clone carries buffers only
    f = ct2.StorageView.from_array(mels)
    exp = [r.sequences_ids for r in lone.generate(f, [PROMPT] * 2, beam_size=5, fixed_new_tokens=6)]
    for r in pool._replicas:
        got = pool._generate_chunk(r, mels, [PROMPT] * 2, 4, 5, 224, 1.0, 1.0, True, True, 6, 0)
        assert [x.sequences_ids for x in got] == exp
    out = {}

    def client(i):
        out[i] = pool.generate(ct2.StorageView.from_array(mels[i % 2:i % 2 + 1]), [PROMPT], beam_size=5, fixed_new_tokens=6)[0].sequences_ids

    th = [threading.Thread(target=client, args=(i,)) for i in range(18)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len({idx for idx, _ in pool._batcher.batches}) >= 2
    single = [lone.generate(ct2.StorageView.from_array(mels[k:k + 1]), [PROMPT], beam_size=5, fixed_new_tokens=6)[0].sequences_ids for k in range(2)]
    assert sum(out[i] != single[i % 2] for i in range(18)) <= 2
    # the parent goes first: its clones keep the weights alive
    pool.close()
    parent = pool._replicas[0]
    lib.wis_model_destroy(parent.handle); parent.handle = None
    for r in pool._replicas[1:]:
        got = ct2._generate_chunk(r, mels, [PROMPT] * 2, 4, 5, 224, 1.0, 1.0, True, True, 6, 0)
        assert [x.sequences_ids for x in got] == exp
    lone.close()
