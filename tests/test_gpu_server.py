"""-m gpu: the re-hosted REST endpoints over the HIP path and the dynamic micro-batcher under concurrent load
(load shape of client/jmeter-asr.jmx:53-90 - N clients POSTing 3sec.flac with model=…&beam_size=5)."""
import asyncio
import os
import threading

import time

import numpy as np
import pytest
import torch  # noqa: F401  (before libwis_hip.so)

pytestmark = pytest.mark.gpu
PROMPT = [50258, 50259, 50359, 50363]


def _multipart(data):
    b = "wisBoundary7"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() \
        + data + f"\r\n--{b}--\r\n".encode()
    return body, {"content-type": f"multipart/form-data; boundary={b}"}


@pytest.fixture(scope="module")
def served():
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch = 8
    models = WhisperModels(s, device_index=[0])
    return create_app(models=models), models


def test_concurrent_generate_calls_batch_and_match_serial_results(golden_dir):
    """16 threads call generate() at once: the micro-batcher must form multi-utterance device batches and every caller must
    get what a lone call gets (up to the batch-composition rounding documented in test_gpu_fullsize: equal scores, ids
    equal unless a near-tie flips)."""
    from wis_hip import ctranslate2 as ct2
    model = ct2.Whisper("synthetic:base", max_batch=8, max_beam=5)
    m3 = np.load(os.path.join(golden_dir, "logmel_3sec.npz"))["mel"].astype(np.float32)
    m10 = np.load(os.path.join(golden_dir, "logmel_10sec.npz"))["mel"].astype(np.float32)
    feats = {0: np.ascontiguousarray(m3[None]), 1: np.ascontiguousarray(m10[None])}
    kw = dict(beam_size=5, fixed_new_tokens=10)
    lone = {k: model.generate(ct2.StorageView.from_array(v), [PROMPT], **kw)[0] for k, v in feats.items()}
    n0 = len(model._batcher.batches)
    out, start = {}, threading.Barrier(16)

    def client(i):
        start.wait()
        beam = 5 if i % 4 else 1                      # a different batch key mixed in: must never share a device batch
        out[i] = (beam, model.generate(ct2.StorageView.from_array(feats[i % 2]), [PROMPT], beam_size=beam, fixed_new_tokens=10)[0])

    th = [threading.Thread(target=client, args=(i,)) for i in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sizes = [n for _, n in model._batcher.batches[n0:]]
    print("device batches formed:", sizes)
    assert sum(sizes) == 16 and max(sizes) > 1 and max(sizes) <= 8
    flips = 0
    for i, (beam, r) in out.items():
        if beam != 5:
            continue
        ref = lone[i % 2]
        assert abs(r.scores[0] - ref.scores[0]) <= 2e-3
        flips += r.sequences_ids != ref.sequences_ids
    assert flips <= 2
    model.close()


def test_asr_and_willow_endpoints(served, golden_dir):
    import httpx
    from wis_hip import audio
    from wis_hip.whisper import do_whisper
    app, models = served
    clip = os.path.join(golden_dir, "clips", "3sec.flac")
    flac = open(clip, "rb").read()
    direct = do_whisper(clip, "tiny", 5, "transcribe", False, None, models=models)
    pcm, _ = audio.load_audio(clip)
    s16 = np.round(pcm * 32768.0).astype("<i2").tobytes()

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis", timeout=120) as c:
            body, hdr = _multipart(flac)
            r = await c.post("/api/asr?model=tiny&beam_size=5&detect_language=False", content=body, headers=hdr)
            assert r.status_code == 200, r.text
            j = r.json()
            assert set(j) == {"infer_time", "infer_speedup", "audio_duration", "language", "text"}
            assert j["audio_duration"] == 3840 and j["language"] == "en" and j["text"] == direct[1]
            assert j["infer_speedup"] == int(3840 // j["infer_time"])
            # Willow device path: raw 16-bit PCM + x-audio-* headers; same samples -> same transcript
            hw = {"x-audio-sample-rate": "16000", "x-audio-bits": "16", "x-audio-channel": "1", "x-audio-codec": "pcm", "x-willow-id": "test"}
            r = await c.post("/api/willow?model=tiny&beam_size=5", content=s16, headers=hw)
            assert r.status_code == 200 and r.json() == {"language": "en", "text": direct[1]}
            r = await c.post("/api/willow?model=tiny&beam_size=5&stats=true", content=flac, headers={"x-audio-codec": "flac"})
            assert r.status_code == 200 and r.json()["text"] == direct[1] and "infer_time" in r.json()
            r = await c.post("/api/willow?model=tiny&beam_size=5&detect_language=true", content=s16, headers=hw)
            assert r.status_code == 200 and len(r.json()["language"]) in (2, 3)
            # jmeter-asr.jmx shape: concurrent identical POSTs; all answers equal, device batches > 1 formed
            mdl = models.get("tiny")
            n0 = len(mdl._batcher.batches)
            rs = await asyncio.gather(*[c.post("/api/asr?model=tiny&beam_size=5&detect_language=False", content=body, headers=hdr) for _ in range(24)])
            assert all(x.status_code == 200 for x in rs)
            texts = [x.json()["text"] for x in rs]
            sizes = [n for _, n in mdl._batcher.batches[n0:]]
            print("REST load: device batches", sizes)
            assert sum(sizes) == 24 and max(sizes) > 1
            assert sum(t != direct[1] for t in texts) <= 2       # near-tie flips only (batch composition changes rounding)

    asyncio.run(go())


def test_streaming_session_equals_offline(golden_dir):
    """BASELINE configs[4] shape: audio arrives in 0.5 s frames; > 30 s recordings are transcribed window by window while
    the audio is still arriving, and stop() returns exactly the offline do_whisper result."""
    from wis_hip import audio
    from wis_hip.settings import APISettings
    from wis_hip.streaming import StreamingSession
    from wis_hip.whisper import WhisperModels, do_whisper
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.concurrent_gpu_chunks = 1           # offline batches of one window, like the streaming schedule: bit-identical arithmetic
    models = WhisperModels(s, device_index=[0])
    # (1) the 29.2 s reference clip: one window, the whole pipeline runs at stop()
    pcm, _ = audio.load_audio(os.path.join(golden_dir, "clips", "30sec.flac"))
    off = do_whisper(pcm, "tiny", 5, models=models, fixed_new_tokens=12)
    sess = StreamingSession("tiny", 5, models=models, fixed_new_tokens=12)
    for i in range(0, pcm.shape[0], 8000):
        sess.feed((pcm[i:i + 8000] * 32768.0).astype("<i2").tobytes(), 2)       # int16 frames, as a WebRTC track delivers them
    mid = sess.interim()
    assert mid[5] == off[5] == 29248 and mid.tokens == off.tokens
    fin = sess.stop()
    assert fin.tokens == off.tokens and fin[1] == off[1] and sess.eager_windows == 0
    assert sess.front_windows == 1        # the final decode took its features from the incremental front-end (HBM), not from host PCM
    # (2) 64 s of seeded noise: 5 windows, the first three complete (and get transcribed) before the recording ends
    rng = np.random.default_rng(11)
    long_pcm = (0.05 * rng.standard_normal(64 * 16000)).astype(np.float32)
    off = do_whisper(long_pcm, "tiny", 5, models=models, fixed_new_tokens=6)
    sess = StreamingSession("tiny", 5, models=models, fixed_new_tokens=6)
    for i in range(0, long_pcm.shape[0], 8000):
        sess.feed(long_pcm[i:i + 8000])
    assert sess.eager_windows == 4        # starts 0, 14, 28, 42 s have their 22 s; the tail window (56 s) does not
    fin = sess.stop()
    assert fin.tokens == off.tokens and fin[5] == off[5] == 64000
    assert sess.front_windows == 5        # every window (4 eager + the tail) was decoded from incrementally computed features
    with pytest.raises(RuntimeError):
        sess.feed(long_pcm[:10])


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_streaming_speculation_beam1_final_equals_offline(golden_dir, size):
    """BASELINE configs[4] for a recording of up to 30 s (ONE window: its encoder needs the whole audio): while the audio arrives the
    session decodes what it has heard every 2 s (beam 1); stop() verifies the last such hypothesis against the final window 16 tokens
    per decoder pass (wis_generate_draft) and only decodes token by token behind the accepted prefix.  The answer must be the
    offline do_whisper answer."""
    from wis_hip import audio
    from wis_hip.settings import APISettings
    from wis_hip.streaming import StreamingSession
    from wis_hip.whisper import WhisperModels, do_whisper
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.beam_size, s.long_beam_size = 1, 1          # beam 1 for every length (the reference's default long-audio beam is 3: a beam search has no single chain to verify)
    s.stream_speculate_s = 2.0
    models = WhisperModels(s, device_index=[0])
    for clip, S in (("30sec", 48), ("10sec", 24)):
        pcm, _ = audio.load_audio(os.path.join(golden_dir, "clips", clip + ".flac"))
        off = do_whisper(pcm, size, 1, models=models, fixed_new_tokens=S)
        sess = StreamingSession(size, 1, models=models, fixed_new_tokens=S)
        for i in range(0, pcm.shape[0], 8000):
            sess.feed((pcm[i:i + 8000] * 32768.0).astype("<i2").tobytes(), 2)
            if sess._spec_job is not None:
                sess._spec_job.result()           # real time: an interim decode (tens of ms) is long done before the next 2 s of audio exist
        runs = sess.spec_runs
        t0 = time.perf_counter()
        fin = sess.stop()
        dt = 1e3 * (time.perf_counter() - t0)
        print(f"{size} {clip}: {runs} interim decodes while the audio arrived; stop() -> result {dt:.1f} ms (offline call {off[2]:.1f} ms), the final decode kept "
              f"{sess.accepted_draft_tokens} of the last hypothesis' {S} tokens; identical to offline: {fin.tokens == off.tokens}")
        assert runs >= pcm.shape[0] // (2 * 16000) - 1
        assert sess.accepted_draft_tokens is not None
        assert fin.tokens == off.tokens and fin[1] == off[1]
        # speculation off: the same answer, no draft
        sess = StreamingSession(size, 1, models=models, fixed_new_tokens=S, speculate_every_s=0)
        for i in range(0, pcm.shape[0], 8000):
            sess.feed(pcm[i:i + 8000])
        fin0 = sess.stop()
        assert fin0.tokens == off.tokens and sess.accepted_draft_tokens is None and sess.spec_runs == 0
    models.get(size).close()


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_streaming_speculation_beam_search_final_equals_offline(golden_dir, size):
    """The same at the reference's OWN settings (request beam 5, long_beam_size 3 from 12 s on, main.py:582-586) with beam-search speculation switched
    on (settings.stream_speculate_beam_search): the session searches what it has heard every 2 s at the beam the final call would use at that length,
    every interim search drafted by the previous one's trajectory, and stop() replays the last trajectory against the final window
    (wis_generate_draft_beam).  Whatever part of the draft the final search follows, the answer must be the offline do_whisper answer."""
    from wis_hip import audio
    from wis_hip.settings import APISettings
    from wis_hip.streaming import StreamingSession
    from wis_hip.whisper import WhisperModels, do_whisper
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.stream_speculate_s, s.stream_speculate_beam_search = 2.0, True
    assert s.long_beam_size == 3 and s.long_beam_size_threshold == 12000          # the reference's defaults (settings.py:14-18)
    models = WhisperModels(s, device_index=[0])
    for clip, S in (("30sec", 48), ("10sec", 24)):
        pcm, _ = audio.load_audio(os.path.join(golden_dir, "clips", clip + ".flac"))
        off = do_whisper(pcm, size, 5, models=models, fixed_new_tokens=S)
        sess = StreamingSession(size, 5, models=models, fixed_new_tokens=S)
        beams = []
        for i in range(0, pcm.shape[0], 8000):
            sess.feed((pcm[i:i + 8000] * 32768.0).astype("<i2").tobytes(), 2)
            if sess._spec_job is not None:
                sess._spec_job.result()
                beams.append(sess._spec_latest[1])
        runs = sess.spec_runs
        t0 = time.perf_counter()
        fin = sess.stop()
        dt = 1e3 * (time.perf_counter() - t0)
        final_beam = 3 if pcm.shape[0] >= 12 * 16000 else 5
        print(f"{size} {clip}: {runs} interim searches (beams {sorted(set(beams))}) while the audio arrived; stop() -> result {dt:.1f} ms at beam {final_beam} (offline call {off[2]:.1f} ms), "
              f"the final search followed the last trajectory for {sess.accepted_draft_tokens} steps; identical to offline: {fin.tokens == off.tokens}")
        assert runs >= pcm.shape[0] // (2 * 16000) - 1 and beams[0] == 5 and beams[-1] == final_beam
        assert sess.accepted_draft_tokens is not None          # a draft of the right beam size was handed over
        assert fin.tokens == off.tokens and fin[1] == off[1]
    models.get(size).close()


def test_logmel_is_reentrant_across_threads(golden_dir):
    """SURVEY 8(b) conventions / round-1 review: wis_logmel called from 32 threads at once with DIFFERENT audio (the three
    reference clips, batches of 1 and 2 windows interleaved so workspaces of different sizes are recycled) - every caller must
    get exactly (bit for bit) what a lone call gets.  The round-1 entry point shared one staging buffer and one stream per
    device and failed this."""
    from wis_hip import audio
    clips = [audio.pad_or_trim(audio.load_audio(os.path.join(golden_dir, "clips", c + ".flac"))[0]) for c in ("3sec", "10sec", "30sec")]
    serial = [audio.log_mel_spectrogram(c, device=0).numpy() for c in clips]
    assert not np.array_equal(serial[0], serial[1]) and not np.array_equal(serial[1], serial[2])
    bad, start = [], threading.Barrier(32)

    def client(i):
        start.wait()
        for rep in range(6):
            k = (i + rep) % 3
            if (i + rep) % 4 == 0:        # a two-window call: grows / recycles workspaces under the other callers
                got = audio.log_mel_spectrogram(np.stack([clips[k], clips[(k + 1) % 3]]), device=0).numpy()
                ok = np.array_equal(got[0], serial[k]) and np.array_equal(got[1], serial[(k + 1) % 3])
            else:
                ok = np.array_equal(audio.log_mel_spectrogram(clips[k], device=0).numpy(), serial[k])
            if not ok:
                bad.append((i, rep, k))

    th = [threading.Thread(target=client, args=(i,)) for i in range(32)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:8]


@pytest.mark.parametrize("fuse", [True, False])
def test_rest_interleaved_clips_from_32_clients(golden_dir, fuse):
    """32 concurrent POSTs of the 3 s / 10 s / 30 s clips interleaved (load shape of client/jmeter-asr.jmx:53-90; the 30 s clip
    switches to the long-audio beam: a second batch key in flight): every request is answered, with its own duration, and with
    the answer a lone request gets.  fuse=False takes the reference's two-step form (wis_logmel from the request threads, then
    generate on host features).

    The served model carries the healthy-margin seeded weights of the parity tests (emb_std 0.06, LayerNorm jitter 0.1: SURVEY 7)
    and the decode length is the measurement convention's 16 tokens, so a greedy decision is not a coin toss: the ORACLE's
    decision margin of each clip's serial answer is computed here, and a response to a clip whose margin exceeds 0.02 (twenty
    times the rounding a different batch composition causes) must be identical to the serial answer.  A response that does
    differ (only possible on a sub-threshold margin) must still be a near-tie of its own clip - the oracle's teacher-forced
    score of its ids within 1e-2 of the serial answer's - and at most 2 of the 32 may differ at all."""
    import httpx
    import torch
    from oracle.whisper_ref import WhisperRef
    from wis_hip import ctranslate2 as ct2, weights as W
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels, do_whisper
    S = 16
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch, s.fuse_logmel, s.fixed_new_tokens = 8, fuse, S
    models = WhisperModels(s, device_index=[0])
    w = W.synthetic_weights("tiny", seed=1234, std=0.02, emb_std=0.06, ln_jitter=0.1)
    a = W.arch("tiny")
    models._models["tiny"] = ct2.Whisper("unused", weights=w, arch=a, max_batch=8, max_beam=5)      # served in place of the default seeded weights
    ref = WhisperRef(w, a["d_model"], a["n_layers"], a["n_heads"])
    app = create_app(models=models)
    names = ("3sec", "10sec", "30sec")
    blobs = {c: open(os.path.join(golden_dir, "clips", c + ".flac"), "rb").read() for c in names}
    serial = {c: do_whisper(os.path.join(golden_dir, "clips", c + ".flac"), "tiny", 1, "transcribe", False, None, models=models) for c in names}
    assert [serial[c][5] for c in names] == [3840, 10688, 29248]
    mem = {c: ref.encode(np.load(os.path.join(golden_dir, f"logmel_{c}.npz"))["mel"][None].astype(np.float32))[0] for c in names}

    def rescore(c, ids):        # oracle: mean teacher-forced log-prob of `ids` for clip c, and the smallest top-1 / top-2 gap on the way
        lg = ref.decode_logits(np.array([PROMPT + list(ids)[:-1]]), mem[c][None])[0]
        total, margin = 0.0, np.inf
        for t, tok in enumerate(ids):
            row = ref.apply_processors(lg[len(PROMPT) - 1 + t][None].double(), t, W.SUPPRESS_IDS, W.SUPPRESS_IDS_BEGIN, True, S)
            lp = torch.log_softmax(row, dim=-1)[0]
            total += float(lp[tok])
            top = torch.topk(lp, 2).values
            margin = min(margin, float(top[0] - top[1]))
        return total / len(ids), margin

    base = {}
    for c in names:
        ids = serial[c].tokens
        assert len(ids) == S
        base[c] = rescore(c, ids)
        print(f"serial {c}: oracle score of the served answer {base[c][0]:.5f}, smallest top-1/top-2 gap along it {base[c][1]:.4f}")

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis", timeout=300) as c:
            reqs = []
            for i in range(32):
                body, hdr = _multipart(blobs[names[i % 3]])
                reqs.append(c.post("/api/asr?model=tiny&beam_size=1&detect_language=False", content=body, headers=hdr))
            return await asyncio.gather(*reqs)

    rs = asyncio.run(go())
    assert all(r.status_code == 200 for r in rs), [r.text for r in rs if r.status_code != 200][:2]
    wrong = 0
    for i, r in enumerate(rs):
        c = names[i % 3]
        j, exp = r.json(), serial[c]
        assert j["audio_duration"] == exp[5] and j["language"] == "en"
        got = [int(t) for t in j["text"].split()]
        assert len(got) == S
        if got != exp.tokens:
            wrong += 1
            sc, _ = rescore(c, got)
            print(f"  response {i} ({c}) differs from its serial answer: oracle score {sc:.5f} vs {base[c][0]:.5f} (margin {base[c][1]:.4f})")
            # greedy answers (3 s / 10 s clips) are forced when every step's gap is wide; the 30 s clip decodes with the long-audio beam
            assert c == "30sec" or base[c][1] <= 0.02, "a forced greedy decision came back different"
            assert abs(sc - base[c][0]) <= 1e-2, "not a near-tie of its own clip: another request's audio or state leaked in"
    print(f"interleaved REST (fuse_logmel={fuse}): {wrong} of 32 responses differ from their serial answer")
    assert wrong <= 2
    models._models["tiny"].close()


def test_concurrent_pcm_requests_keep_their_own_audio(golden_dir):
    """The fused input path under concurrency: 24 threads call generate() with the PCM of three DIFFERENT clips (device batches
    mix them); the length-normalised beam score is a continuous function of the audio, so every caller must get the score of
    ITS OWN clip (within the batch-composition rounding, 2e-3) - and the three clips' scores are far apart."""
    from wis_hip import _lib, audio, ctranslate2 as ct2
    model = ct2.Whisper("synthetic:base", max_batch=8, max_beam=5)
    names = ("3sec", "10sec", "30sec")
    pcm = {c: np.ascontiguousarray(audio.pad_or_trim(audio.load_audio(os.path.join(golden_dir, "clips", c + ".flac"))[0])[None]) for c in names}
    kw = dict(beam_size=5, fixed_new_tokens=10, input_kind=_lib.WIS_IN_PCM_HOST)
    lone = {c: model.generate(ct2.StorageView.from_array(pcm[c]), [PROMPT], **kw)[0] for c in names}
    gaps = [abs(lone[a].scores[0] - lone[b].scores[0]) for a in names for b in names if a < b]
    print("lone scores", {c: round(lone[c].scores[0], 5) for c in names})
    assert min(gaps) > 4e-3                     # (measured: 6.5e-3 .. 1.4e-2 between the three clips; the batch-composition rounding is < 1e-3)
    out, start = {}, threading.Barrier(24)
    n0 = len(model._batcher.batches)

    def client(i):
        start.wait()
        out[i] = model.generate(ct2.StorageView.from_array(pcm[names[i % 3]]), [PROMPT], **kw)[0]

    th = [threading.Thread(target=client, args=(i,)) for i in range(24)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sizes = [n for _, n in model._batcher.batches[n0:]]
    print("device batches formed:", sizes)
    assert sum(sizes) == 24 and max(sizes) > 1
    for i, r in out.items():
        own = abs(r.scores[0] - lone[names[i % 3]].scores[0])
        other = min(abs(r.scores[0] - lone[c].scores[0]) for c in names if c != names[i % 3])
        assert own <= 2e-3 and own < other, (i, names[i % 3], r.scores[0], {c: lone[c].scores[0] for c in names})
    model.close()


def test_translate_and_tokenizer_branches(tmp_path, golden_dir):
    """do_whisper's translate branch (reference main.py:729-748: a second generate with the <|translate|> prompt) and the
    text path through a real `tokenizer.json` in the model directory (main.py:329-334, 714), on a CTranslate2-layout model
    directory written here (tiny architecture, seeded weights, a word-level vocabulary covering every id)."""
    import json
    from tokenizers import Tokenizer, decoders, models as tk_models
    from wis_hip import ctranslate2 as ct2, weights as W
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels, do_whisper
    mdir = tmp_path / "tovera-wis-whisper-tiny"
    os.makedirs(mdir)
    w = W.synthetic_weights("tiny", seed=77, emb_std=0.06, ln_jitter=0.1)
    W.write_ct2_model_bin(str(mdir / "model.bin"), w, aliases={"decoder/projection/weight": "decoder/embeddings/weight"})
    with open(mdir / "config.json", "w") as f:
        json.dump(dict(suppress_ids=W.SUPPRESS_IDS, suppress_ids_begin=W.SUPPRESS_IDS_BEGIN, lang_ids=W.LANG_IDS), f)
    s = APISettings()
    s.whisper_model_path = str(tmp_path / "tovera-wis-whisper-{size}")
    models = WhisperModels(s, device_index=[0])
    with pytest.raises(FileNotFoundError, match="tokenizer"):       # a real checkpoint without its tokenizer is refused
        models.get("tiny")
    tok = Tokenizer(tk_models.WordLevel({f"w{i}": i for i in range(W.N_VOCAB)}, unk_token="w0"))
    tok.decoder = decoders.Fuse()
    tok.save(str(mdir / "tokenizer.json"))
    models = WhisperModels(s, device_index=[0])
    clip = os.path.join(golden_dir, "clips", "3sec.flac")
    res = do_whisper(clip, "tiny", 5, "transcribe", False, "en", translate=True, models=models, fixed_new_tokens=7)
    language, text, _, translation, _, duration = res
    assert language == "en" and duration == 3840 and len(res.tokens) == 7 and len(res.translation_tokens) == 7
    assert text == "".join(f"w{t}" for t in res.tokens)                       # decoded by the directory's tokenizer.json, not id strings
    assert translation == "".join(f"w{t}" for t in res.translation_tokens)
    # the translation is what generate returns for the <|translate|> prompt, and differs from the transcription prompt's result
    from wis_hip import audio
    x = np.ascontiguousarray(audio.pad_or_trim(audio.load_audio(clip)[0])[None])
    direct = models.get("tiny").generate(ct2.StorageView.from_array(x), [[W.SOT, W.LANG_IDS[0], W.TRANSLATE, W.NO_TIMESTAMPS]], beam_size=5,
                                         fixed_new_tokens=7, input_kind=ct2._lib.WIS_IN_PCM_HOST)[0]
    assert direct.sequences_ids[0] == res.translation_tokens
    # (the seeded random model barely listens to its prompt, so the ids of the two tasks may coincide: check the prompts themselves)
    seen = []
    mdl = models.get("tiny")
    real_generate = mdl.generate
    mdl.generate = lambda f, prompts, **kw: (seen.append([list(p) for p in prompts]), real_generate(f, prompts, **kw))[1]
    try:
        do_whisper(clip, "tiny", 5, "transcribe", False, "en", translate=True, models=models, fixed_new_tokens=3)
    finally:
        mdl.generate = real_generate
    assert seen == [[[W.SOT, W.LANG_IDS[0], W.TRANSCRIBE, W.NO_TIMESTAMPS]], [[W.SOT, W.LANG_IDS[0], W.TRANSLATE, W.NO_TIMESTAMPS]]]
    # task="translate" puts the translate token into the main prompt (main.py:656-663)
    t2 = do_whisper(clip, "tiny", 5, "translate", False, "en", models=models, fixed_new_tokens=7)
    assert t2.tokens == res.translation_tokens and t2[3] is None
    # the two-step (host features) form gives the same ids as the fused form
    s.fuse_logmel = False
    again = do_whisper(clip, "tiny", 5, "transcribe", False, "en", models=models, fixed_new_tokens=7)
    assert again.tokens == res.tokens and again[1] == text


def test_melstream_incremental_equals_batch_logmel(golden_dir):
    """wis_melstream_* (SURVEY 8(f)3): PCM fed in ragged pieces - single samples, pieces shorter than a frame, pieces spanning many
    tiles - must give, bit for bit, the log-mel of the complete padded window; tiles are transformed as soon as their samples are
    complete (tiles_done follows 2560 j + 2600 <= n), the right window edge waits for finish, 30 s windows are trimmed."""
    from wis_hip import audio
    rng = np.random.default_rng(21)
    st = audio.MelStream(0)
    for clip in ("3sec", "10sec", "30sec", "noise31"):
        if clip == "noise31":
            pcm = (0.1 * rng.standard_normal(31 * 16000)).astype(np.float32)       # longer than the window: trimmed (pad_or_trim)
        else:
            pcm = audio.load_audio(os.path.join(golden_dir, "clips", clip + ".flac"))[0]
        want = audio.log_mel_spectrogram(audio.pad_or_trim(pcm), device=0).numpy()
        st.reset()
        i, seen = 0, []
        sizes = [1, 199, 200, 2400, 1, 159, 7001]
        while i < pcm.shape[0]:
            k = sizes[len(seen) % len(sizes)] if len(seen) < 40 else int(rng.integers(1, 40000))
            st.feed(pcm[i:i + k])
            i += k
            n = min(i, pcm.shape[0], 480000)
            exp_tiles = 188 if n >= 480000 else max(0, min((n - 2600) // 2560 + 1 if n >= 2600 else 0, (480000 + 200 - 2800) // 2560 + 1))
            assert st.samples == n and st.tiles_done == exp_tiles, (clip, n, st.tiles_done, exp_tiles)
            seen.append(st.tiles_done)
        got = st.finish()
        assert st.device_ptr and np.array_equal(got, want), (clip, float(np.abs(got - want).max()))
        assert np.array_equal(st.finish(), want)          # idempotent
        with pytest.raises(Exception):
            st.feed(pcm[:10])                             # a finished window takes no more audio
    # an empty window is the log-mel of silence
    st.reset()
    assert np.array_equal(st.finish(), audio.log_mel_spectrogram(np.zeros(480000, np.float32), device=0).numpy())
    st.close()


def test_two_server_processes_behind_one_port_on_the_gpu(golden_dir):
    """`python -m wis_hip.server --workers-per-node 2` with the REAL engine (tiny, seeded weights): two server processes - both pinned to GPU 0 here, the
    box has one; a node gives each its own (`HIP_VISIBLE_DEVICES=i`) - share the port (SO_REUSEPORT); requests over real sockets come back with the same
    transcript whichever process the kernel hands the connection to, and SIGTERM drains the node.  (tests/test_server_cpu.py covers supervision and
    batching with the fake engine.)"""
    import json
    import signal
    import socket
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s0:
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "willow-inference-server_amd"), os.environ.get("PYTHONPATH", "")]),
               WHISPER_MODEL_PATH="synthetic:{size}", FIXED_NEW_TOKENS="12", MAX_BATCH="4", REPLICAS_PER_GPU="1")
    sup = subprocess.Popen([sys.executable, "-m", "wis_hip.server", "--host", "127.0.0.1", "--port", str(port), "--workers-per-node", "2", "--devices", "0,0",
                            "--log-level", "warning", "--graceful-timeout", "5"], env=env)
    clip = open(os.path.join(golden_dir, "clips", "3sec.flac"), "rb").read()
    b = "wisTestBoundary"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() + clip + f"\r\n--{b}--\r\n".encode()
    req = (f"POST /api/asr?task=transcribe&output=json&model=tiny&beam_size=3&detect_language=False HTTP/1.1\r\nHost: x\r\nContent-Type: multipart/form-data; boundary={b}\r\n"
           f"Content-Length: {len(body)}\r\nConnection: close\r\n\r\n").encode() + body

    def post():
        with socket.create_connection(("127.0.0.1", port), timeout=60) as c:
            c.sendall(req)
            data = b""
            while True:
                chunk = c.recv(65536)
                if not chunk:
                    break
                data += chunk
        head, _, payload = data.partition(b"\r\n\r\n")
        assert head.startswith(b"HTTP/1.1 200"), head[:200]
        return json.loads(payload.decode())

    def ping():
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1) as c:
                c.sendall(b"GET /api/ping HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n")
                return b"200" in c.recv(64)
        except OSError:
            return False

    try:
        t_end = time.time() + 120
        while time.time() < t_end and not all(ping() for _ in range(6)):
            time.sleep(0.3)
        assert ping()
        texts = [post()["text"] for _ in range(12)]            # 12 connections: both listeners get some (the kernel hashes them)
        assert len(set(texts)) == 1 and texts[0]
        print("two server processes, one port, GPU 0: 12 requests, one transcript:", texts[0][:60])
        # ... and under concurrency: 16 client threads x 4 requests each, both processes batching on the shared GPU
        import threading
        got, errs = [], []

        def client():
            try:
                for _ in range(4):
                    got.append(post()["text"])
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
        ts = [threading.Thread(target=client) for _ in range(16)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        el = time.perf_counter() - t0
        assert not errs and len(got) == 64 and set(got) == {texts[0]}, (errs[:2], len(got))
        print(f"  64 concurrent requests in {el:.2f} s ({64 / el:.0f} requests/s through two processes on one GPU), all with the same transcript")
        sup.send_signal(signal.SIGTERM)
        assert sup.wait(30) == 0
    finally:
        if sup.poll() is None:
            sup.kill()
