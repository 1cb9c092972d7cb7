"""not gpu: the micro-batcher and the REST re-host's request handling - everything that does not need the
GPU: batch formation / routing / error propagation with a fake device, multipart + x-audio-* parsing, the reference's 400s."""
import asyncio
import io
import threading
import time
import wave

import numpy as np
import pytest


def test_microbatcher_coalesces_equal_keys_and_routes_results():
    from wis_hip.batching import MicroBatcher
    seen = []

    def run(ctx, key, payloads):
        seen.append((ctx, key, list(payloads)))
        time.sleep(0.03)                       # the "GPU" is busy: later submissions queue up meanwhile
        return [(key, p * 10) for p in payloads]

    mb = MicroBatcher(["gpu0"], run, lambda key: 4 if key == "a" else 2)
    out = {}

    def client(i):
        key = "a" if i % 3 else "b"
        out[i] = mb.submit(key, [i, i + 100])

    th = [threading.Thread(target=client, args=(i,)) for i in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    mb.close()
    for i in range(12):
        key = "a" if i % 3 else "b"
        assert out[i] == [(key, i * 10), (key, (i + 100) * 10)]             # right caller, payload order kept
    assert all(len({k for _, k, _ in [b]}) == 1 for b in seen)
    for ctx, key, payloads in seen:
        assert len(payloads) <= (4 if key == "a" else 2)                      # capacity respected per key
    assert max(len(p) for _, _, p in seen) > 1                                # batching happened
    assert sum(len(p) for _, _, p in seen) == 24
    assert sum(n for _, n in mb.batches) == 24


def test_microbatcher_two_workers_and_error_propagation():
    from wis_hip.batching import MicroBatcher

    def run(ctx, key, payloads):
        time.sleep(0.02)
        if key == "boom":
            raise RuntimeError("device fault")
        return [f"{ctx}:{p}" for p in payloads]

    mb = MicroBatcher(["g0", "g1"], run, lambda key: 2)
    res, errs = {}, []

    def client(i):
        try:
            res[i] = mb.submit("boom" if i == 5 else "k", [i])
        except RuntimeError as e:
            errs.append((i, str(e)))

    th = [threading.Thread(target=client, args=(i,)) for i in range(10)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs == [(5, "device fault")]
    assert sorted(res) == [0, 1, 2, 3, 4, 6, 7, 8, 9] and all(v[0].endswith(f":{i}") for i, v in res.items())
    assert {v[0].split(":")[0] for v in res.values()} == {"g0", "g1"}         # both replicas were fed
    mb.close()
    with pytest.raises(RuntimeError):
        mb.submit("k", [1])


def test_affinity_routes_device_resident_rows_to_their_replica():
    """Rows whose features already live on one GPU (streaming sessions: WIS_IN_MEL_DEV) may only be run by THAT replica's
    worker, and still coalesce with each other; rows without affinity go to whichever worker is free."""
    from wis_hip.batching import MicroBatcher
    g0, g1 = object(), object()
    gate = threading.Event()
    seen = []

    def run(ctx, key, payloads):
        gate.wait(2)
        seen.append((0 if ctx is g0 else 1, key, list(payloads)))
        return list(payloads)

    mb = MicroBatcher([g0, g1], run, lambda k: 4)
    res = {}
    th = [threading.Thread(target=lambda i=i: res.__setitem__(i, mb.submit("dev", [i], affinity=g1))) for i in range(6)]
    th += [threading.Thread(target=lambda i=i: res.__setitem__(i, mb.submit("host", [i]))) for i in range(6, 9)]
    for t in th:
        t.start()
    time.sleep(0.1)
    gate.set()
    for t in th:
        t.join()
    assert sorted(res) == list(range(9)) and all(res[i] == [i] for i in range(9))
    assert all(w == 1 for w, k, _ in seen if k == "dev")                     # never on the other GPU
    assert max(len(p) for _, k, p in seen if k == "dev") > 1                 # and they did form a batch
    mb.close()


def test_lone_request_is_not_delayed():
    from wis_hip.batching import MicroBatcher
    mb = MicroBatcher(["g"], lambda c, k, p: list(p), lambda k: 8)
    t0 = time.perf_counter()
    for i in range(50):
        assert mb.submit("k", [i]) == [i]
    assert (time.perf_counter() - t0) / 50 < 5e-3        # no batching timer on the latency path
    mb.close()


class _FakeModels:
    """Stands in for WhisperModels so request handling can be exercised without a GPU; reaching the model is an error here."""

    def __init__(self):
        from wis_hip.settings import APISettings
        from wis_hip.whisper import _Tokenizer
        self.settings = APISettings()
        self.tokenizer = _Tokenizer(None)

    def tokenizer_for(self, size):
        return self.tokenizer

    def get(self, size):
        if size not in ("tiny", "base", "small", "medium", "large"):
            raise ValueError(f"unknown model {size!r}")
        return object()


def _client(app):
    import httpx
    return httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis")


def _multipart(data, field="audio_file"):
    b = "xYzBoundary123"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"{field}\"; filename=\"a.flac\"\r\nContent-Type: application/octet-stream\r\n\r\n").encode() \
        + data + f"\r\n--{b}--\r\n".encode()
    return body, {"content-type": f"multipart/form-data; boundary={b}"}


def test_multipart_parser_roundtrip():
    from wis_hip.server import parse_multipart
    payload = bytes(range(256)) * 40 + b"\r\n--not-the-boundary\r\n"
    body, hdr = _multipart(payload)
    assert parse_multipart(body, hdr["content-type"]) == payload
    with pytest.raises(ValueError):
        parse_multipart(body, "application/json")
    with pytest.raises(ValueError):
        parse_multipart(_multipart(payload, field="other")[0], hdr["content-type"])


def test_write_stream_wav_is_a_wav_our_decoder_reads():
    from wis_hip import audio
    from wis_hip.server import write_stream_wav
    pcm = (np.sin(np.arange(1600) * 0.05) * 12000).astype("<i2")
    f = write_stream_wav(pcm.tobytes(), 16000, 16, 1)
    with wave.open(io.BytesIO(f.getvalue())) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 1600)
    x, sr = audio.load_audio(f)                          # host C decoder inside libwis_hip.so: no GPU needed
    assert sr == 16000 and np.array_equal(x, pcm.astype(np.float32) / 32768.0)


def test_endpoints_reference_error_behaviour():
    from wis_hip.server import create_app
    app = create_app(models=_FakeModels())

    async def go():
        async with _client(app) as c:
            r = await c.get("/api/ping")
            assert r.status_code == 200 and r.json() == {"message": "pong"}
            body, hdr = _multipart(b"this is not audio")
            r = await c.post("/api/asr?model=tiny&force_language=klingon", content=body, headers=hdr)
            assert r.status_code == 400 and r.json() == {"error": "Invalid force_language"}
            r = await c.post("/api/asr?model=tiny", content=body, headers=hdr)
            assert r.status_code == 400 and r.json() == {"error": "Invalid audio"}
            r = await c.post("/api/asr?model=huge", content=body, headers=hdr)
            assert r.status_code == 400 and "unknown model" in r.json()["error"]
            r = await c.post("/api/willow?model=tiny&force_language=xx", content=b"\0" * 64, headers={"x-audio-codec": "pcm"})
            assert r.status_code == 400 and r.json() == {"error": "Invalid force_language"}
            r = await c.post("/api/willow?model=tiny", content=b"\0" * 64, headers={"x-audio-codec": "opus", "x-audio-sample-rate": "16000"})
            assert r.status_code == 400 and r.json() == {"error": "Invalid audio"}
            r = await c.post("/api/willow?model=tiny", content=b"\0" * 64, headers={"x-audio-codec": "pcm"})   # missing rate/bits/channel headers
            assert r.status_code == 400 and r.json() == {"error": "Invalid audio"}
            r = await c.post("/api/willow?model=tiny", content=b"RIFFjunk", headers={"x-audio-codec": "wav"})
            assert r.status_code == 400 and r.json() == {"error": "Invalid audio"}
            r = await c.post("/api/willow?model=tiny&voice_auth=true", content=b"", headers={"x-audio-codec": "pcm"})
            assert r.status_code == 400
            # a beam the engine cannot serve (legal in the reference, main.py:1180) is a 400 with a message, never a 500
            for q in ("beam_size=10", "beam_size=0", "beam_size=five"):
                r = await c.post(f"/api/asr?model=tiny&{q}", content=body, headers=hdr)
                assert r.status_code == 400 and "beam_size" in r.json()["error"], (q, r.status_code, r.text)
                r = await c.post(f"/api/willow?model=tiny&{q}", content=b"\0" * 64, headers={"x-audio-codec": "pcm"})
                assert r.status_code == 400 and "beam_size" in r.json()["error"]

    asyncio.run(go())


def test_datachannel_protocol_state_machine():
    """reference main.py:906-996: ping/start/stop -> pong/log/infer/error, per-message model / beam overrides."""
    import json
    from wis_hip.streaming import DataChannelProtocol
    made = []

    class FakeSession:
        def __init__(self, model, beam_size, task, detect_language, force_language, models=None):
            self.args, self.fed = (model, beam_size, task, detect_language), 0
            made.append(self)

        def feed(self, frame, width):
            self.fed += len(frame) // width

        def stop(self):
            return ("en", f"heard {self.fed} samples", 12.5, None, 80, int(self.fed / 16))

    p = DataChannelProtocol(models=_FakeModels(), model="medium", beam_size=1, session_factory=FakeSession)
    dec = lambda xs: [json.loads(x) for x in xs]
    assert dec(p.on_message("not json")) == [{"type": "error", "message": "could not parse message", "obj": None}]
    assert dec(p.on_message(json.dumps({"type": "ping", "message": "hi"}))) == [{"type": "pong", "message": "hi", "obj": None}]
    assert dec(p.on_message(json.dumps({"type": "stop"})))[0]["message"] == "Recording not yet started"
    p.on_audio(b"\0\0" * 100)                                   # before "start": dropped
    assert dec(p.on_message(json.dumps({"type": "start"})))[0]["type"] == "log"
    for _ in range(10):
        p.on_audio(b"\1\0" * 1600)
    out = dec(p.on_message(json.dumps({"type": "stop", "obj": {"model": "large", "beam_size": 5}})))
    assert made[-1].args == ("large", 5, "transcribe", False) and made[-1].fed == 16000
    assert [m["type"] for m in out] == ["log", "infer", "log", "log", "log"]
    assert out[1]["obj"] == {"text": "heard 16000 samples"} and "beam size 5" in out[0]["message"]
    assert out[3]["message"] == "ASR Audio Duration: 1000 ms" and out[4]["message"] == "ASR Speedup: 80x faster than realtime"
    assert dec(p.on_message(json.dumps({"type": "nope"})))[0]["message"] == 'unknown message type "nope"'
    assert dec(p.on_message(json.dumps({"type": "stop"})))[0]["type"] == "error"      # recorder was consumed


def test_streaming_session_schedule_without_gpu():
    """The window schedule of a streaming session (which windows are transcribed before stop(), what stop() merges) with the
    model replaced by a deterministic stand-in: must equal the offline chunk_iter + LCS computation on the complete audio."""
    from wis_hip import audio
    from wis_hip.streaming import StreamingSession

    def fake_ids(piece):                       # "transcript" of a window: a few ids derived from its content
        q = np.round(np.abs(piece[::16000]) * 1000).astype(int) % 50000
        return [int(v) for v in q]

    class Sess(StreamingSession):
        def __init__(self, models):
            self.models = models
            s = models.settings
            self.model_name, self.task, self.beam_size = "tiny", "transcribe", s.beam_size
            self.detect_language = self.force_language = None
            self.fixed_new_tokens = 0
            self._whisper = None
            self._chunks, self._n = [], 0
            import threading
            from concurrent.futures import ThreadPoolExecutor
            self._lock, self._pool = threading.Lock(), ThreadPoolExecutor(max_workers=2)
            self._windows, self._language_job, self._closed, self.eager_windows = {}, None, False, 0
            self.calls, self.detect_calls = [], []

        def _detect(self, first_window):
            self.detect_calls.append(first_window.shape[0])
            return "en"

        def _window_tokens(self, piece, beam, language):
            if hasattr(language, "result"):
                language = language.result()
            assert language == "en"
            self.calls.append((piece.shape[0], beam))
            return fake_ids(piece)

        def _window_decode(self, piece, beam, language, stream=None, draft=None, want_traj=False):
            return self._window_tokens(piece, beam, language), None, None

    models = _FakeModels()
    rng = np.random.default_rng(5)
    pcm = rng.standard_normal(75 * 16000).astype(np.float32)
    s = Sess(models)
    for i in range(0, pcm.shape[0], 4000):
        s.feed(pcm[i:i + 4000])
    # 75 s: windows start every 14 s; complete 22 s windows: starts 0, 14, 28, 42 (42 + 22 <= 75) ; 56 s and 70 s windows are tails
    assert s.eager_windows == 4 and all(n == audio.chunk_len and b == models.settings.long_beam_size for n, b in s.calls)
    out = s.stop()
    expect = audio.find_longest_common_sequence([(fake_ids(p), st) for p, st in audio.chunk_iter(pcm)], models.tokenizer)
    assert out.tokens == [int(t) for t in expect] and out[5] == 75000
    assert len(s.calls) == 6                   # the two tail windows were transcribed at stop(), nothing twice
    assert s.detect_calls == [audio.chunk_len]  # the language was resolved ONCE, from window 0 (no race between windows)
    # short recording: no eager work, one window, request beam below the long-audio threshold
    s2 = Sess(models)
    s2.feed((pcm[:5 * 16000] * 32768 * 0.01).astype("<i2").tobytes(), 2)
    assert s2.buffered_ms == 5000 and s2.eager_windows == 0
    out2 = s2.stop()
    assert s2.calls == [(5 * 16000, models.settings.beam_size)] and len(out2.tokens) == 5
    with pytest.raises(RuntimeError):
        s2.feed(pcm[:10])


def test_streaming_speculation_schedule_without_gpu():
    """The interim decodes of a recording that is still one window (<= 30 s) with the model replaced by a stand-in: one every
    `speculate_every_s` of NEW audio and never two at once, at the beam the final call would use at that length, each drafted by the
    previous hypothesis OF THAT BEAM (token chain at beam 1, trajectory at beam > 1), the latest handed to the final decode at stop();
    none for a language that must be detected on the final audio, past 30 s, with the interval set to 0 or while the GPU has no replica to
    spare - and the final answer never depends on any of it."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from wis_hip.streaming import StreamingSession

    class FakeWhisper:
        def __init__(self):
            self.queued, self.running, self._replicas = 0, 0, [None] * 4

        def load(self, device=None):
            return self.queued, self.running

        def replicas_on(self, device):
            return 4

    class Sess(StreamingSession):
        def __init__(self, models, beam=1, every=2.0, detect=False, gate=None):
            self.models = models
            self.model_name, self.task, self.beam_size = "tiny", "transcribe", beam
            self.detect_language, self.force_language = detect, None
            self.fixed_new_tokens = 0
            self._whisper, self._replica = FakeWhisper(), None
            self._chunks, self._n = [], 0
            self._lock, self._pool = threading.Lock(), ThreadPoolExecutor(max_workers=2)
            self._windows, self._language_job, self._closed, self.eager_windows = {}, None, False, 0
            self._spec_every, self._spec_n, self._spec_job, self._spec_latest, self._spec_busy = every, 0, None, None, 0.5
            self.spec_runs, self.spec_skipped, self.spec_ms, self.accepted_draft_tokens = 0, 0, 0.0, None
            self.calls, self.gate = [], gate

        def _detect(self, first_window):
            return "en"

        def _window_decode(self, piece, beam, language, stream=None, draft=None, want_traj=False):
            if self.gate is not None:
                self.gate.wait(5)
            ids = [int(piece.shape[0] // 16000), 7, 8]                    # "transcript": seconds heard, then two fixed ids
            d = dict(draft or {})
            self.calls.append((piece.shape[0], beam, d))
            acc = None
            if "draft_tokens" in d:
                acc = sum(1 for a, b in zip(d["draft_tokens"], ids) if a == b)
            if "draft_trajectory" in d:
                acc = len(d["draft_trajectory"][0])
            traj = (np.full((3, beam), ids[0], np.int32), np.zeros((3, beam), np.int32)) if want_traj else None
            return ids, acc, traj

    models = _FakeModels()
    sec = np.zeros(16000, np.float32)

    def settle(s):
        if s._spec_job is not None:
            s._spec_job.result(5)

    # 7 s fed one second at a time, each interim finishing before the next second arrives: interims at 2, 4, 6 s
    s = Sess(models)
    for _ in range(7):
        s.feed(sec)
        settle(s)
    assert s.spec_runs == 3 and [(n // 16000, b) for n, b, _ in s.calls] == [(2, 1), (4, 1), (6, 1)]
    assert [d for _, _, d in s.calls] == [{}, {"draft_tokens": [2, 7, 8]}, {"draft_tokens": [4, 7, 8]}]          # each one drafted by the previous hypothesis
    out = s.stop()
    assert s.calls[-1] == (7 * 16000, 1, {"draft_tokens": [6, 7, 8]}) and out.tokens == [7, 7, 8]
    assert s.accepted_draft_tokens == 2                                        # ids 7, 8 of the draft survived; the first did not

    # a slow interim: never two in flight, the next one starts only after it finished and covers everything heard by then
    gate = threading.Event()
    s = Sess(models, gate=gate)
    for _ in range(9):
        s.feed(sec)
    assert s.spec_runs == 0 and s._spec_job is not None and s._spec_n == 2 * 16000
    gate.set()
    settle(s)
    s.feed(sec)
    settle(s)
    assert [(n // 16000) for n, _, _ in s.calls] == [2, 10]
    s.close()

    # a beam search speculates too (round 6): interims at the request's beam, drafted by the previous search's TRAJECTORY; the final decode
    # verifies the last one (reference: every recording of 12 s or more is decoded at long_beam_size, main.py:582-586)
    s = Sess(models, beam=5)
    for _ in range(7):
        s.feed(sec)
    assert s._spec_job is None and s.spec_runs == 0          # ... when switched on (settings.stream_speculate_beam_search; off by default)
    s.close()
    models_b = _FakeModels()
    models_b.settings.stream_speculate_beam_search = True
    s = Sess(models_b, beam=5)
    for _ in range(7):
        s.feed(sec)
        settle(s)
    assert [(n // 16000, b) for n, b, _ in s.calls] == [(2, 5), (4, 5), (6, 5)] and s.calls[0][2] == {}
    assert all(list(d) == ["draft_trajectory"] and d["draft_trajectory"][0].shape == (3, 5) for _, _, d in s.calls[1:])
    out = s.stop()
    assert out.tokens == [7, 7, 8] and s.calls[-1][1] == 5 and int(s.calls[-1][2]["draft_trajectory"][0][0, 0]) == 6 and s.accepted_draft_tokens == 3

    # no speculation: language detection on the final audio, interval 0
    for kw in (dict(detect=True), dict(every=0.0)):
        s = Sess(models, **kw)
        for _ in range(6):
            s.feed(sec)
        assert s._spec_job is None and s.spec_runs == 0
        out = s.stop()
        assert out.tokens == [6, 7, 8] and s.accepted_draft_tokens is None and s.calls[-1][2] == {}

    # optional work: none while requests are queued or more than half of the GPU's replicas are running device batches
    s = Sess(models)
    s._whisper.queued = 3
    for _ in range(5):
        s.feed(sec)
    assert s._spec_job is None and s.spec_skipped == 2
    s._whisper.queued, s._whisper.running = 0, 3
    for _ in range(2):
        s.feed(sec)
    assert s._spec_job is None and s.spec_skipped == 3
    s._whisper.running = 2
    for _ in range(2):
        s.feed(sec)
        settle(s)
    assert s.spec_runs == 1 and s.calls[-1][0] == 8 * 16000
    s.close()

    # the beam changes where the final call's would (long_beam_size from 12 s on): a draft of another beam size is not handed over
    models5 = _FakeModels()
    models5.settings.stream_speculate_beam_search = True
    s = Sess(models5)
    thr = models5.settings.long_beam_size_threshold // 1000
    lb = models5.settings.long_beam_size
    if lb != 1:
        for _ in range(thr + 4):
            s.feed(sec)
            settle(s)
        beams = [b for _, b, _ in s.calls]
        assert beams[0] == 1 and beams[-1] == lb and all(b == (1 if n // 16000 < thr else lb) for n, b, _ in s.calls)
        first_long = next(c for c in s.calls if c[1] == lb)
        assert first_long[2] == {}                                             # (the beam-1 hypothesis before it is no draft for a beam search)
        out = s.stop()
        assert s.calls[-1][1] == lb and list(s.calls[-1][2]) == ["draft_trajectory"] and s.accepted_draft_tokens == 3


def _pool_batcher(n_gpus, per_gpu, log, cap=8):
    from wis_hip.batching import MicroBatcher

    class Replica:
        def __init__(self, device):
            self.device = device

    def run(ctx, key, payloads):
        t0 = time.perf_counter()
        time.sleep(0.030 + 0.004 * len(payloads))          # a device batch: ~30 ms + 4 ms per utterance
        log.append((ctx.device, t0, time.perf_counter(), len(payloads), key))
        return [p[0] for p in payloads]

    # the order Whisper.__init__ builds its pool in: one replica per GPU, then the clones of each
    workers = [Replica(g) for g in range(n_gpus)] + [Replica(g) for g in range(n_gpus) for _ in range(per_gpu - 1)]
    return MicroBatcher(workers, run, lambda key: cap)


def test_burst_over_a_replica_pool_forms_device_batches_not_singletons():
    """BASELINE configs[3] / client/jmeter-asr.jmx: 64 requests arriving together at a model with 8 GPUs x 4 replicas (32 idle
    workers).  Every worker grabbing what it finds would make ~32 batches of 1-2; the batcher wakes one taker per batch worth of
    work and lets it linger (<= 0.5 ms idle GPU, <= 2 ms busy GPU) while the burst is still arriving."""
    # (thread-timing test: on a box that is busy with something else the 64 client threads may trickle in over many milliseconds - up to
    # three attempts, one clean burst is the claim)
    for attempt in range(3):
        log = []
        mb = _pool_batcher(8, 4, log)
        go = threading.Barrier(64)
        res = {}

        def client(i):
            go.wait()
            res[i] = mb.submit("k", [(i, time.perf_counter())])

        th = [threading.Thread(target=client, args=(i,)) for i in range(64)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all(res[i] == [i] for i in range(64))
        sizes = [n for _, _, _, n, _ in log]
        assert sum(sizes) == 64
        print("burst of 64 over 8 x 4 replicas: device batches", sizes, "lingered", mb.lingers)
        # no GPU ever runs more than two SMALL (< 4) batches at once
        crowded = False
        for g in range(8):
            small = [(a, b) for d, a, b, n, _ in log if d == g and n < 4]
            crowded = crowded or any(sum(1 for a2, b2 in small if a2 < b and a < b2) > 2 for a, b in small)
        # mean batch >= 6, and the work spread over the GPUs instead of piling onto the first ones
        if np.mean(sizes) >= 6 and not crowded and len({d for d, *_ in log}) >= 6:
            break
        mb.close()
    else:
        raise AssertionError(f"three bursts, none formed device batches: last {sizes}")
    # a lone request afterwards starts at once (no batching timer on the latency path)
    log.clear()
    waits = []
    for i in range(20):
        t0 = time.perf_counter()
        assert mb.submit("k", [(i, t0)]) == [i]
        waits.append(log[-1][1] - t0)
    print(f"lone request: submit -> device batch start p50 {1e3 * float(np.median(waits)):.3f} ms, max {1e3 * max(waits):.3f} ms")
    assert float(np.median(waits)) < 1e-3 or float(np.min(waits)) < 3e-4          # (median on an idle box ~0.1 ms)
    mb.close()


def test_steady_load_keeps_batches_full_on_one_gpu():
    """one GPU, 4 replicas, 24 closed-loop clients: when a batch of 8 completes its clients resubmit within a millisecond; the free
    worker waits that long instead of leaving with the first one or two"""
    for attempt in range(3):          # (thread-timing test: up to three attempts on a busy box)
        log = []
        mb = _pool_batcher(1, 4, log)
        stop = time.perf_counter() + 0.6

        def client(i):
            while time.perf_counter() < stop:
                mb.submit("k", [(i, time.perf_counter())])

        th = [threading.Thread(target=client, args=(i,)) for i in range(24)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        sizes = [n for _, _, _, n, _ in log]
        print("steady load, 24 clients on 1 x 4 replicas: mean device batch", round(float(np.mean(sizes)), 2), "batches", len(sizes), "lingered", mb.lingers)
        mb.close()
        if np.mean(sizes[4:]) >= 3.5:          # (5.2-6.8 on an idle host; the bound leaves room for a loaded CI box)
            break
    else:
        raise AssertionError(f"steady load: mean device batch {np.mean(sizes[4:]):.2f} in three attempts")


def test_device_affinity_lets_any_replica_of_that_gpu_take_the_rows_and_they_coalesce():
    """rows whose features live in one GPU's memory (streaming windows, WIS_IN_MEL_DEV) are bound to the DEVICE: whichever replica of
    that GPU is free runs them, several sessions' windows share a device batch, no other GPU ever sees them"""
    log = []
    mb = _pool_batcher(2, 3, log, cap=4)
    go = threading.Barrier(9)
    res = {}

    def client(i):
        go.wait()
        res[i] = mb.submit("dev", [(i, 0.0)], affinity=("device", 1)) if i < 6 else mb.submit("host", [(i, 0.0)])

    th = [threading.Thread(target=client, args=(i,)) for i in range(9)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(res[i] == [i] for i in range(9))
    dev = [(d, n) for d, _, _, n, key in log if key == "dev"]
    assert dev and all(d == 1 for d, _ in dev)            # bound rows never leave their GPU
    assert sum(n for _, n in dev) == 6 and len(dev) <= 3   # six rows at capacity 4: two or three batches, not six
    assert sum(n for _, _, _, n, key in log if key == "host") == 3
    mb.close()


def test_one_server_process_per_gpu_behind_one_port():
    """`python -m wis_hip.server --workers-per-node N` (the deployment that serves a node: reference entrypoint.sh:19-21 runs ONE gunicorn worker over
    `device_index=[*range(n)]`, main.py:295; one Python process answers ~380 requests/s, an MI355X decodes ~170 utterances/s of the jmeter shape):
    N server processes share the port (SO_REUSEPORT), each with its own model replicas / micro-batcher, supervised by the parent.  Fake engine
    (tools/fake_engine_app.py: a device batch of B costs 30 + 4 B ms on a GPU that runs one batch at a time), raw-socket load generator in a
    process of its own (tools/host_ceiling.py): every worker takes traffic, nothing fails, saturated workers form full device batches, the
    supervisor restarts a worker that dies and drains the node on SIGTERM."""
    import json
    import os
    import signal
    import socket
    import subprocess
    import sys
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # ---- throughput / batching through the tool (4 processes: the build container has 8 cores for servers AND generator)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_ceiling.py"), "--processes", "4", "--replicas", "1", "--clients", "96", "--client-procs", "1", "--seconds", "3"],
                         capture_output=True, text=True, timeout=180)
    line = next((ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")), None)
    assert line, out.stdout + out.stderr
    r = json.loads(line[7:])
    print(out.stdout.strip().splitlines()[-2])
    assert r["errors"] == 0 and r["supervisor_exit"] == 0
    assert len(r["batches_per_worker"]) == 4 and min(r["batches_per_worker"]) > 0          # the kernel spread the connections over every listener
    assert r["mean_device_batch"] >= 6.0, r                                                # 24 connections per one-batch-at-a-time GPU: the batches fill
    assert r["requests_per_s"] >= 150, r          # engine capacity 516 requests/s; 444 on the build container's 8 cores - a floor far below, not a benchmark (shared CI hosts)
    # ---- supervision: a worker that dies is replaced, SIGTERM drains the node
    with socket.socket() as s0:
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tools"), os.path.join(ROOT, "willow-inference-server_amd"), os.environ.get("PYTHONPATH", "")]))
    env.pop("HIP_VISIBLE_DEVICES", None)
    sup = subprocess.Popen([sys.executable, "-m", "wis_hip.server", "--host", "127.0.0.1", "--port", str(port), "--workers-per-node", "2", "--app", "fake_engine_app:create_app",
                            "--log-level", "warning", "--graceful-timeout", "3"], env=env)

    def ping():
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1) as c:
                c.sendall(b"GET /api/ping HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n")
                return b"200" in c.recv(64)
        except OSError:
            return False

    def children():
        o = subprocess.run(["ps", "-o", "pid=", "--ppid", str(sup.pid)], capture_output=True, text=True).stdout.split()
        return sorted(int(x) for x in o)

    try:
        t_end = time.time() + 60
        while time.time() < t_end and not (len(children()) == 2 and all(ping() for _ in range(6))):
            time.sleep(0.2)
        kids = children()
        assert len(kids) == 2 and ping()
        os.kill(kids[0], signal.SIGKILL)
        t_end = time.time() + 30
        while time.time() < t_end and (len(children()) != 2 or kids[0] in children()):
            time.sleep(0.2)
        assert len(children()) == 2 and kids[0] not in children()                             # replaced
        t_end = time.time() + 30
        while time.time() < t_end and not all(ping() for _ in range(8)):
            time.sleep(0.3)
        assert all(ping() for _ in range(8))                                                  # both listeners answer again
        sup.send_signal(signal.SIGTERM)
        assert sup.wait(20) == 0
        time.sleep(0.3)
        assert not ping()
    finally:
        if sup.poll() is None:
            sup.kill()
