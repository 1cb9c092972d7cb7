"""bench.py — the reference's headline benchmark on MI355X: Whisper large-v2, beam 5, the 3.84 s clip
(README.md:71-73 rows; BASELINE.json metric), measured the way WIS measures `infer_speedup`
(main.py:576,756-760): realtime multiple = audio_ms / infer_ms.

    python bench.py [--gpus N --steps K --warmup W] [--model large --beam 5 --clip 3sec --batch 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (log-mel -> encoder -> cross-K/V -> prefill -> beam-search decode ->
token ids on the host) over one device batch of `--batch` utterances whose PCM is already resident in HBM
when the timed region starts.  Weights are seeded synthetic tensors at the true large-v2 shapes (no
checkpoint exists offline); decode length follows the measurement convention of SURVEY §8d (S = 16 new
tokens for the 3.84 s clip, EOT masked until S then forced).  With N > 1 every rank owns one GPU and its
own utterance stream (weak scaling, no data-path collective); weights are generated on rank 0 and
broadcast once over RCCL/xGMI.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "willow-inference-server_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PROMPT = [50258, 50259, 50359, 50363]          # <|startoftranscript|><|en|><|transcribe|><|notimestamps|> (main.py:656-663)
FIXED_NEW = {"3sec": 16, "10sec": 40, "30sec": 96}   # SURVEY §8d decode-length convention
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(size, weights, pcm, beam, fixed_new, audio_ms):
    """PyTorch-fp32 oracle (kind "port": CTranslate2 4.1.0's int8 CPU path is not installable offline) timed on the host
    cores on the SAME workload, measured end to end (no extrapolation): log-mel + encoder of the window + the cross-K/V
    projection, the prompt prefill and all S + 1 beam-search steps of the utterance (about 15 s of CPU work for large-v2)."""
    import torch
    from oracle import audio_ref
    from oracle.whisper_ref import WhisperRef
    from wis_hip import weights as W
    cores = max(1, (os.cpu_count() or 2) // 2)     # reference CPU path: intra_threads = cpu_count // 2 (main.py:297-302)
    # thread count for the encoder-sized GEMMs: best of a few on a probe matmul of the FFN shape (more threads is not faster
    # on every host)
    probe_a, probe_w = torch.randn(1500, 1280), torch.randn(1280, 5120)
    enc_t, enc_best = cores, None
    for t in sorted({min(cores, 16), min(cores, 32), min(cores, 64), cores}):
        torch.set_num_threads(t)
        torch.mm(probe_a, probe_w)
        ta = time.perf_counter()
        for _ in range(3):
            torch.mm(probe_a, probe_w)
        dt = time.perf_counter() - ta
        if enc_best is None or dt < enc_best:
            enc_best, enc_t = dt, t
    torch.set_num_threads(enc_t)
    a = W.arch(size)
    ref = WhisperRef(weights, a["d_model"], a["n_layers"], a["n_heads"])
    # log-mel: the reference's own wis/audio.py when its tree is present (the build container), else the oracle's numpy restatement
    # of it (the GPU box has no /root/reference)
    mel_fn, mel_src = (lambda x: audio_ref.log_mel_spectrogram(audio_ref.pad_or_trim(x))), "oracle/audio_ref.py (numpy restatement of wis/audio.py)"
    if os.path.exists("/root/reference/wis/audio.py"):
        try:
            sys.path.insert(0, "/root/reference")
            from wis.audio import log_mel_spectrogram as ref_mel, pad_or_trim as ref_pad
            ref_mel(ref_pad(pcm))          # first call builds the window / filter tensors
            mel_fn, mel_src = (lambda x: ref_mel(ref_pad(x)).numpy()), "the reference's wis/audio.py (torch CPU)"
        except Exception as e:      # noqa: BLE001
            log(f"reference wis/audio.py not usable ({e}); timing the oracle's log-mel")
    t0 = time.perf_counter()
    mel = np.asarray(mel_fn(pcm), np.float32)
    t1 = time.perf_counter()
    mem = ref.encode(mel[None])[0]
    t2e = time.perf_counter()
    kw = dict(beam_size=beam, suppress_ids=W.SUPPRESS_IDS, suppress_begin=W.SUPPRESS_IDS_BEGIN, memory=mem)
    # the decode steps are hundreds of small matmuls: with every host thread on each of them torch spends its time in
    # thread hand-offs, so the thread count is the best of a few on ONE step (the encoder keeps its own best count)
    best_t, best = cores, None
    for t in sorted({min(cores, 8), min(cores, 16), min(cores, 32), cores}):
        torch.set_num_threads(t)
        ta = time.perf_counter()
        ref.generate(None, PROMPT, max_new_tokens=1, **kw)
        tb = time.perf_counter()
        if best is None or tb - ta < best:
            best, best_t = tb - ta, t
    torch.set_num_threads(best_t)
    t2 = time.perf_counter()
    ids, _ = ref.generate(None, PROMPT, fixed_new=fixed_new, **kw)          # cross-K/V + prefill + every one of the S + 1 steps
    t3 = time.perf_counter()
    enc_s, dec_s = t2e - t0, t3 - t2
    total = enc_s + dec_s
    # (round-5 review item 7: `kind` says which log-mel the host timed - "port" alone only where the reference's own wis/audio.py ran)
    kind = "port" if mel_src.startswith("the reference") else "port (log-mel: numpy oracle)"
    return {"value": round(audio_ms / 1000.0 / total, 4), "unit": "x realtime", "cores": max(enc_t, best_t), "kind": kind,
            "logmel_timed": mel_src,          # which log-mel ran on the host: the reference's own module only where /root/reference exists (not on the GPU box)
            "sample": (f"torch-fp32 oracle (KV-cached), the whole utterance measured: log-mel {1e3 * (t1 - t0):.0f} ms [{mel_src}] + encoder {1e3 * (t2e - t1):.0f} ms on {enc_t} host "
                       f"threads (best of 16/32/64/{cores} on a probe GEMM) + cross-KV, prefill and all {fixed_new + 1} beam-{beam} steps {1e3 * dec_s:.0f} ms on {best_t} threads "
                       f"(best of 8/16/32/{cores} on one step), {len(ids)} tokens = {total:.2f} s per utterance. Stand-in for the CT2 int8 CPU path (not installable offline); "
                       f"published CT2-int8 CPU figure: large beam1 3.84 s clip 3344 ms on Threadripper 5955WX (README.md:103)")}


def p50(xs):
    return float(np.median(np.asarray(xs, np.float64)))


def decode_step_bytes(a, B, beam, P, steps, w8=False):
    """Algorithmic HBM bytes of ONE decode step, averaged over the `steps` steps of a run: the decoder weight stream
    W (6 matrices per layer + the vocabulary projection, read once per step whatever the row count), the cross-attention K/V of
    every utterance (B * C) and the self-attention cache rows read so far (B * beam * t * s)."""
    d, L, V = a["d_model"], a["n_layers"], a["n_vocab"]
    wbytes = 1 if w8 else 2
    W = (L * 14 * d * d + V * d) * wbytes
    C = L * 2 * a["n_audio_ctx"] * d * 2
    s_row = L * 2 * d * 2
    t_avg = (P - 1) + (steps + 1) / 2.0
    return W + B * C + B * beam * t_avg * s_row


def concurrent_batches(lib, handles, dev, pcm, beam, B, fixed_new, audio_ms, iters=6, max_r=None):
    """Utterances per second with 1 .. len(handles) device batches of B utterances in flight on ONE GPU: every handle is a replica
    with its own stream, activations and KV caches over the SAME weight copy (wis_model_clone), driven by its own host thread -
    what `inter_threads` > 1 does in the reference's CTranslate2 model (main.py:341-355).  A decode chain is latency-bound (tens of
    small dependent launches per layer), so a second and third batch in flight run in the gaps of the first."""
    import threading
    from wis_hip import _lib, audio
    win = np.ascontiguousarray(np.tile(audio.pad_or_trim(pcm)[None], (B, 1)).astype(np.float32))
    keep = _lib.DevBuf.from_numpy(win, dev)
    prompt = np.ascontiguousarray(np.tile(np.array(PROMPT, np.int32), (B, 1)))
    rows = []
    for R in range(1, (max_r or len(handles)) + 1):
        def worker(h, n):
            opts = _lib.GenOpts(_lib.WIS_IN_PCM_DEV, beam, 0, 1.0, 1.0, 1, 1, fixed_new, 0)
            ids = np.zeros((B, 224), np.int32); lens = np.zeros(B, np.int32); scores = np.zeros(B, np.float32)
            for _ in range(n):
                _lib.check(lib.wis_generate(h, keep.ptr, B, prompt.ctypes.data_as(C.POINTER(C.c_int32)), len(PROMPT), C.byref(opts),
                                            ids.ctypes.data_as(C.POINTER(C.c_int32)), lens.ctypes.data_as(C.POINTER(C.c_int32)), scores.ctypes.data_as(C.POINTER(C.c_float))))
        for phase, n in (("warm", 2), ("timed", iters)):
            th = [threading.Thread(target=worker, args=(handles[i], n)) for i in range(R)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
        rows.append({"batches_in_flight": R, "utterances_per_s": round(R * iters * B / dt, 1), "aggregate_x_realtime": round(R * iters * B * audio_ms / 1e3 / dt, 1),
                     "ms_per_device_batch": round(1e3 * dt / iters, 2)})
    return {"workload": f"{B} x 3sec.flac per device batch, beam {beam}, replicas on one GPU sharing one weight copy, one host thread each", "rows": rows}


def rest_load(handle, a, dev, clients, iterations, fixed_new, clip_bytes, audio_ms, rest_batch=8, extra_handles=(), side_sessions=0, side_pcm=None, side_gate=None):
    """The reference's own load shape, client/jmeter-asr.jmx:53-90: `clients` threads, each looping
    POST /api/asr?task=transcribe&output=json&model=large&beam_size=5&detect_language=False with the 3.84 s clip as the
    multipart field `audio_file`.  Served by the re-hosted endpoint (wis_hip/server.py) in this process over the ASGI transport
    (no socket): container decode, micro-batching and the HIP path are all inside the measurement.
    side_sessions > 0: that many StreamingSessions (30sec.flac at beam 1 - the default beam_size, the case that speculates by default; S = 96)
    run beside the clients for the whole measured phase, each asking for an interim decode after every 2 s of audio as fast as the decodes
    allow, at most 40 x real time (the worst case for the REST traffic); side_gate =
    settings.stream_speculate_max_busy (None: the default gate; 1e9: no gate)."""
    import asyncio
    import threading
    import httpx
    from wis_hip import ctranslate2 as ct2
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch, s.fixed_new_tokens = rest_batch, fixed_new
    if side_gate is not None:
        s.stream_speculate_max_busy = side_gate
    if side_sessions:
        s.long_beam_size = 1          # (the sessions' recordings pass 12 s; the REST clients name their beam in the URL)
    models = WhisperModels(s, device_index=[dev])
    model = ct2.Whisper.from_handles([(handle, dev)] + [(h, dev) for h in extra_handles], a, max_batch=rest_batch, max_beam=5)
    models._models["large"] = model
    app = create_app(models=models, max_workers=max(64, clients))
    b = "wisBenchBoundary"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() + clip_bytes + f"\r\n--{b}--\r\n".encode()
    hdr = {"content-type": f"multipart/form-data; boundary={b}"}
    url = "/api/asr?task=transcribe&output=json&model=large&beam_size=5&detect_language=False"
    lat = []
    stop_side = threading.Event()
    side_stats = {"sessions_started": 0, "interim_decodes": 0, "skipped_by_the_load_gate": 0, "interim_ms_total": 0.0}

    def side_session():
        from wis_hip.streaming import StreamingSession
        chunk = 2 * 16000
        while not stop_side.is_set():
            sess = StreamingSession("large", 1, models=models, fixed_new_tokens=96, incremental=False)
            side_stats["sessions_started"] += 1
            try:
                for i in range(0, side_pcm.shape[0], chunk):
                    if stop_side.is_set():
                        break
                    sess.feed(side_pcm[i:i + chunk])
                    job = sess._spec_job
                    if job is not None:
                        job.result()
                    time.sleep(0.05)       # 2 s of audio every >= 50 ms: 40 x real time (the feeding itself must not be what the REST clients compete with)
            finally:
                side_stats["interim_decodes"] += sess.spec_runs
                side_stats["skipped_by_the_load_gate"] += sess.spec_skipped
                side_stats["interim_ms_total"] += sess.spec_ms
                sess.close()

    async def client(c):
        for _ in range(iterations):
            t = time.perf_counter()
            r = await c.post(url, content=body, headers=hdr)
            assert r.status_code == 200, r.text
            lat.append(1e3 * (time.perf_counter() - t))

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis", timeout=600) as c:
            await asyncio.gather(*[client(c) for _ in range(min(clients, rest_batch))])        # warm-up: one device batch
            lat.clear()
            side = [threading.Thread(target=side_session, daemon=True) for _ in range(side_sessions)]
            for t in side:
                t.start()
            if side:
                await asyncio.sleep(0.3)           # the sessions' first (undrafted) interim decodes are under way
            n0 = len(model._batcher.batches)
            t0 = time.perf_counter()
            await asyncio.gather(*[client(c) for _ in range(clients)])
            el = time.perf_counter() - t0
            sizes = [n for _, n in model._batcher.batches[n0:]]
            stop_side.set()
            for t in side:
                t.join(30)
            return el, sizes

    elapsed, sizes = asyncio.run(go())
    n = clients * iterations
    model.close()
    model._replicas = []          # the handle belongs to the caller
    if side_sessions:
        sizes = [z for z in sizes if z > 1] or sizes      # (the sessions' own decodes are device batches of one)
    extra = {}
    if side_sessions:
        extra = {"streaming_sessions_beside_the_clients": side_sessions, "load_gate (stream_speculate_max_busy)": "default 0.5" if side_gate is None else side_gate,
                 **{k: (round(v, 1) if isinstance(v, float) else v) for k, v in side_stats.items()}}
    return {**extra, "load": f"client/jmeter-asr.jmx shape: {clients} concurrent clients x {iterations} POST /api/asr (model=large, beam_size=5, 3sec.flac), in-process ASGI transport, device batches of up to {rest_batch}, "
                    f"{1 + len(extra_handles)} replica(s) on the GPU",
            "utterances_per_s": round(n / elapsed, 2), "aggregate_x_realtime": round(n * audio_ms / 1e3 / elapsed, 1),
            "p50_request_ms": round(p50(lat), 2), "max_request_ms": round(max(lat), 2), "device_batches": sizes[:32], "mean_device_batch": round(float(np.mean(sizes)), 2)}


def _lib_kind():
    from wis_hip import _lib
    return _lib.WIS_IN_PCM_HOST


def streaming_bench(handle, a, dev, clip_path):
    """BASELINE configs[4]: long-form / streaming (the reference records the whole WebRTC track and then makes ONE do_whisper call,
    main.py:963-971).  client/30sec.flac is fed to a StreamingSession as 20 ms int16 frames and the time from stop() to the result is
    reported:
      * without speculation (stream_speculate_s = 0), frames back to back, with the incremental log-mel front-end (features built in HBM
        while the audio arrives) and without it (log-mel of the whole window at stop()): the window's decode IS the latency;
      * `beam3` - the reference's own settings (request beam 5, long_beam_size 3 from 12 s on, main.py:582-586): the session decodes what it
        has heard every 2 s at the beam the final call would use at that length, each interim search drafted by the previous one's
        trajectory, and stop() verifies the last trajectory against the final window 16 steps per decoder pass (wis_generate_draft_beam);
      * `beam1` - the same with beam_size = long_beam_size = 1 (wis_generate_draft);
      real-time arrival is emulated by letting every interim decode finish before the next frame is fed (an interim decode takes < 0.2 s,
      2 s of audio take 2 s); `speculation_gpu_ms_per_session` = wall time of all interim decodes of a session (what the speculation costs);
      * 64 s of seeded noise, which crosses the 30 s chunking threshold: windows whose 22 s are complete are transcribed while the audio is
        still arriving (eager windows), stop() only has the tail left.
    Same engine handle as the headline (large-v2)."""
    from wis_hip import audio, ctranslate2 as ct2
    from wis_hip.settings import APISettings
    from wis_hip.streaming import StreamingSession
    from wis_hip.whisper import WhisperModels, do_whisper
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch = 8
    models = WhisperModels(s, device_index=[dev])
    model = ct2.Whisper.from_handles([(handle, dev)], a, max_batch=8, max_beam=5)
    models._models["large"] = model
    pcm, _ = audio.load_audio(clip_path)
    i16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    frames = [i16[i:i + 320].tobytes() for i in range(0, i16.shape[0], 320)]
    out = {"clip": "client/30sec.flac as 20 ms int16 frames, large-v2, request beam 5 -> long-audio beam 3 (reference defaults), S=96"}

    def speculating_row(request_beam, what):
        offl = []
        for _ in range(3):
            t0 = time.perf_counter()
            ref = do_whisper(pcm, "large", request_beam, models=models, fixed_new_tokens=96)
            offl.append(1e3 * (time.perf_counter() - t0))
        lat, acc, runs, cost, interims = [], [], 0, [], []
        for _ in range(3):
            sess = StreamingSession("large", request_beam, models=models, fixed_new_tokens=96, incremental=True)
            seen = 0.0
            interims = []
            for f in frames:
                sess.feed(f, 2)
                if sess._spec_job is not None and not sess._spec_job.done():
                    sess._spec_job.result()
                    interims.append(round(sess.spec_ms - seen, 1)); seen = sess.spec_ms
            runs = sess.spec_runs
            t0 = time.perf_counter()
            r = sess.stop()
            lat.append(1e3 * (time.perf_counter() - t0))
            acc.append(sess.accepted_draft_tokens)
            cost.append(sess.spec_ms)
        return {"config": what, "stop_to_result_ms": round(p50(lat), 3), "offline_do_whisper_ms": round(p50(offl), 3), "stop_over_offline": round(p50(lat) / p50(offl), 3),
                "interim_decodes_while_audio_arrived": runs, "draft_accepted_by_the_final_decode (tokens at beam 1, search steps at beam > 1)": acc,
                "same_tokens_as_offline": bool(r.tokens == ref.tokens),
                # (a forced 96-token decode on seeded weights: where the two differ, a near-tie between low-ranked beams fell the other way in the multi-row pass's summation order)
                "leading_tokens_in_common_with_offline": int(next((i for i, (x, y) in enumerate(zip(r.tokens, ref.tokens)) if x != y), min(len(r.tokens), len(ref.tokens)))),
                "speculation_gpu_ms_per_session": round(float(np.mean(cost)), 1),
                "interim_decode_ms (last session; each drafted by the previous one)": interims}

    try:
        for inc in (True, False):
            lat = []
            for _ in range(3):
                sess = StreamingSession("large", 5, models=models, fixed_new_tokens=96, incremental=inc)      # (reference defaults: a beam search at stop(), no speculation)
                for f in frames:
                    sess.feed(f, 2)
                t0 = time.perf_counter()
                r = sess.stop()
                lat.append(1e3 * (time.perf_counter() - t0))
                assert len(r.tokens) == 96
            out["stop_to_result_ms_incremental_front_end" if inc else "stop_to_result_ms_logmel_at_stop"] = round(p50(lat[1:]), 3)
        t0 = time.perf_counter()
        ref = do_whisper(pcm, "large", 5, models=models, fixed_new_tokens=96)
        out["offline_do_whisper_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        out["same_tokens_as_offline"] = bool(ref.tokens == r.tokens)
        try:
            s.stream_speculate_beam_search = True      # (off by default: see settings.py)
            out["beam3"] = speculating_row(5, "30sec.flac, large-v2, the reference's settings: request beam 5 below 12 s, long_beam_size 3 from there on (main.py:582-586), S=96, "
                                              "interim search every 2 s of audio, the final search replays the last interim trajectory (wis_generate_draft_beam)")
        except Exception as e:      # noqa: BLE001
            out["beam3"] = {"failed": repr(e)}
        try:
            # what the mechanism delivers when the draft HOLDS: the same window searched again with its own trajectory as the draft (on these seeded
            # weights the low-ranked beams of a search are chaotic - 1.2 s more audio re-orders them within a few steps, tools/traj_lab.py - so the
            # streamed row above leaves its draft early; the top hypothesis itself is stable: 94 of 96 ids)
            x = np.ascontiguousarray(audio.pad_or_trim(pcm)[None], np.float32)
            sv, pr = ct2.StorageView.from_array(x), [[50258, 50259, 50359, 50363]]
            kw = dict(beam_size=3, fixed_new_tokens=96, input_kind=_lib_kind())
            r0 = model.generate(sv, pr, return_trajectory=True, **kw)[0]
            plain, drafted, acc = [], [], None
            for _ in range(4):
                t0 = time.perf_counter(); model.generate(sv, pr, **kw); plain.append(1e3 * (time.perf_counter() - t0))
                t0 = time.perf_counter(); r1 = model.generate(sv, pr, draft_trajectory=r0.trajectory, **kw)[0]; drafted.append(1e3 * (time.perf_counter() - t0))
                acc = r1.accepted_draft_tokens
            out["beam3"]["when_the_draft_holds"] = {"what": "the final window searched at beam 3 with the trajectory of a search over the SAME window as its draft", "plain_ms": round(p50(plain[1:]), 3),
                                                    "drafted_ms": round(p50(drafted[1:]), 3), "drafted_over_plain": round(p50(drafted[1:]) / p50(plain[1:]), 3), "steps_accepted": acc,
                                                    "same_ids": bool(r1.sequences_ids[0] == r0.sequences_ids[0])}
        except Exception as e:      # noqa: BLE001
            out.setdefault("beam3", {})["when_the_draft_holds"] = {"failed": repr(e)}
        s.stream_speculate_beam_search = False
        s.beam_size, s.long_beam_size = 1, 1
        try:
            out["beam1"] = speculating_row(1, "30sec.flac, large-v2, beam 1 at every length (long_beam_size = 1), S=96, interim decode every 2 s of audio (wis_generate_draft)")
        except Exception as e:      # noqa: BLE001
            out["beam1"] = {"failed": repr(e)}
        finally:
            s.beam_size, s.long_beam_size = APISettings().beam_size, APISettings().long_beam_size
        rng = np.random.default_rng(1234)
        noise = (0.1 * rng.standard_normal(64 * 16000)).astype(np.float32)
        sess = StreamingSession("large", 5, models=models, fixed_new_tokens=48, incremental=True, speculate_every_s=0)
        t_feed = time.perf_counter()
        for i in range(0, noise.shape[0], 320):
            sess.feed(noise[i:i + 320])
        t0 = time.perf_counter()
        r = sess.stop()
        out["noise_64s"] = {"feed_wall_ms (back to back, eager windows decode meanwhile)": round(1e3 * (t0 - t_feed), 1), "stop_to_result_ms": round(1e3 * (time.perf_counter() - t0), 3),
                            "eager_windows": sess.eager_windows, "windows_from_incremental_front_end": sess.front_windows, "tokens": len(r.tokens)}
    finally:
        model.close()
        model._replicas = []      # the handle belongs to the caller
    return out


def rest_base_beam1(dev, clip_bytes, audio_ms):
    """BASELINE configs[0] shape on the GPU: Whisper base, beam 1, client/3sec.flac through the re-hosted REST endpoint, one request at a time."""
    import asyncio
    import httpx
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch, s.fixed_new_tokens = 8, FIXED_NEW["3sec"]
    models = WhisperModels(s, device_index=[dev])
    app = create_app(models=models)
    b = "wisBenchBoundary"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() + clip_bytes + f"\r\n--{b}--\r\n".encode()
    hdr = {"content-type": f"multipart/form-data; boundary={b}"}
    lat = []

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis", timeout=600) as c:
            for i in range(23):
                t = time.perf_counter()
                r = await c.post("/api/asr?model=base&beam_size=1&detect_language=False", content=body, headers=hdr)
                assert r.status_code == 200, r.text
                if i >= 3:
                    lat.append(1e3 * (time.perf_counter() - t))
    asyncio.run(go())
    models.get("base").close()
    return {"config": "base beam 1, 3sec.flac over REST /api/asr, serial requests (configs[0] shape on the GPU; container decode + HTTP handling included)",
            "p50_request_ms": round(p50(lat), 3), "x_realtime": round(audio_ms / p50(lat), 1)}


def natural_eot(lib, a, weights, dev, pcm, beam, audio_ms, make_step, timed, last_timing, batch=8):
    """The termination path every reference request takes (main.py:687-693 passes no max_length): the search ends on EOT by itself.
    Seeded weights never prefer EOT, so a second replica carries the SAME weights with an EOT ramp on the learned decoder positions
    (tests/eot_ramp.py: start 3, slope 1.2 - the torch-fp32 oracle ends this utterance after 17 decoder passes, the count of the S = 16
    convention); nothing masked, nothing forced.  Reported next to the fixed-length call with the SAME number of decoder passes on the
    same handle: what the host-side termination protocol costs (progress record in mapped host memory, two passes in the queue)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eot_ramp import with_eot_ramp
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    wr = with_eot_ramp(weights, start=3, slope=1.2)
    arena, index = W.build_arena(wr)
    h = ct2.create_handle(a, arena, index, dev, max_batch=batch, max_beam=max(beam, 1))
    del arena, wr
    rows = {}
    try:
        for B in (1, batch):
            st_nat, (_k, lens) = make_step(h, pcm, beam, B, 0, _lib.WIS_IN_PCM_DEV)
            l_nat = timed(st_nat, 20 if B == 1 else 10, 3)
            tm = last_timing(h)
            ran, needed, n_tok = int(tm["decode_steps"]), int(tm["decode_steps_needed"]), int(lens[0])
            st_fix, _k2 = make_step(h, pcm, beam, B, max(needed - 1, 1), _lib.WIS_IN_PCM_DEV)       # S + 1 passes = the passes the natural search needed
            l_fix = timed(st_fix, 20 if B == 1 else 10, 3)
            tf = last_timing(h)
            rows[f"batch_{B}"] = {"natural_eot_p50_ms": round(p50(l_nat), 3), "fixed_length_same_passes_p50_ms": round(p50(l_fix), 3),
                                  "natural_over_fixed": round(p50(l_nat) / p50(l_fix), 4), "x_realtime_natural": round(B * audio_ms / p50(l_nat), 1),
                                  "tokens_returned": n_tok, "decoder_passes_needed": needed, "decoder_passes_enqueued": ran, "overrun_passes": ran - needed,
                                  "decode_ms_natural (device clock, first beam step -> last utterance finished)": tm["decode_ms"], "decode_ms_fixed": tf["decode_ms"]}
    finally:
        lib.wis_model_destroy(h)
    return {"workload": f"large-v2 beam {beam}, 3sec.flac, EOT-ramp weights (start 3, slope 1.2), fixed_new_tokens = 0: the search ends on EOT; queue depth 2", **rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="large")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--clip", default="3sec")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--compute-type", default="float16", choices=["float16", "int8_float16"],
                    help="decoder weight storage; the headline number is float16 (int8_float16 mirrors the reference's GPU default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--natural-eot", action="store_true", help="run the natural-EOT row even with --no-extras")
    ap.add_argument("--no-natural-eot", action="store_true", help="skip the natural-EOT row (a second large-v2 replica on EOT-ramp weights)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configurations, the batch-8 line, the boundary variants and the REST load replay")
    ap.add_argument("--rest-clients", type=int, default=64)
    ap.add_argument("--no-batched", action="store_true", help="N > 1 only: skip the 8-utterances-per-device-batch leg (BASELINE configs[3]) after the headline")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    # One HIP runtime per process: torch bundles its own libamdhip64; when torch.cuda is needed (distributed branch) torch must
    # be imported BEFORE libwis_hip.so so that the library's libamdhip64.so.7 dependency resolves to the runtime torch loaded.
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    backend = os.environ.get("WIS_DIST_BACKEND", "nccl")          # "gloo": CPU dry-run of this branch (tests/test_dist_cpu.py)
    dry = use_dist and backend == "gloo"
    if use_dist:
        import torch  # noqa: F401
    from wis_hip import _lib, audio, ctranslate2 as ct2, weights as W
    from wis_hip import dist as wdist
    extras = world == 1 and not args.no_extras and not dry and (args.model, args.beam, args.clip, args.batch) == ("large", 5, "3sec", 1)
    dist = torch = None
    if use_dist:
        import torch
        import torch.distributed as dist
    if not dry:
        lib = _lib.load()
        _lib.require_gpu()
    dev = local_rank

    # launched by torch.distributed.run (RANK set) -> always take the distributed branch, even with one rank, so that the
    # RCCL init / weight broadcast / device-arena hand-off is the same code at every N
    if use_dist:
        if dry:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    a = W.arch(args.model)
    batched_B = 8 if ((world > 1 or os.environ.get("WIS_BENCH_FORCE_BATCHED")) and args.batch == 1 and not args.no_batched and not dry) else 0      # N > 1: BASELINE configs[3]'s 8 utterances per GPU measured after the headline
    max_batch = max(args.batch, 16 if extras else 1, batched_B)
    t0 = time.perf_counter()
    weights = None
    if not use_dist:
        weights = W.synthetic_weights(args.model, seed=1234)
        arena, index = W.build_arena(weights)
        handle = ct2.create_handle(a, arena, index, dev, max_batch=max_batch, max_beam=max(args.beam, 1),
                                   weight_bits=8 if args.compute_type == "int8_float16" else 16)
        del arena
    else:
        # one-time broadcast of the weight arena from rank 0 (RCCL over xGMI; no collective at request time): only rank 0
        # generates / holds the weights, every other rank computes the layout from the shapes alone
        index, total = W.synthetic_layout(args.model)
        arena = None
        if rank == 0:
            weights = W.synthetic_weights(args.model, seed=1234)
            arena, index0 = W.build_arena(weights)
            assert index0 == index
        buf = wdist.broadcast_arena(arena, total, src=0, device=None if dry else f"cuda:{dev}")
        del arena
        if dry:
            # CPU dry-run: everything up to the device hand-off ran (layout, broadcast, sharding); there is no GPU to continue on
            chk = int(buf[::4099].to(torch.int64).sum().item())
            sums = [None] * world
            dist.all_gather_object(sums, chk)
            assert len(set(sums)) == 1, "arena differs between ranks"
            lo, hi = wdist.shard_range(world * args.batch * args.steps, world, rank)
            t = torch.tensor([float(hi - lo)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if rank == 0:
                print(json.dumps({"dry_run": True, "backend": "gloo", "n_ranks": world, "arena_bytes": int(total), "arena_checksum": chk,
                                  "utterances_per_rank_max": int(t.item()), "scaling": "weak"}), flush=True)
            dist.destroy_process_group()
            return
        torch.cuda.synchronize()
        handle = ct2.create_handle(a, None, index, dev, max_batch=max_batch, max_beam=max(args.beam, 1),
                                   weight_bits=8 if args.compute_type == "int8_float16" else 16,
                                   arena_device_ptr=(buf.data_ptr(), total))
        del buf
        torch.cuda.empty_cache()
    log(f"[rank {rank}] model '{args.model}' ready on device {dev} in {time.perf_counter() - t0:.1f} s "
        f"({lib.wis_model_device_bytes(handle) / 1e9:.2f} GB resident)")

    def clip_pcm(clip):
        path = os.path.join(ROOT, "tests", "golden", "clips", clip + ".flac")
        pcm, _sr = audio.load_audio(path)
        return pcm, 1000.0 * pcm.shape[0] / 16000.0, path

    def make_step(h, pcm, beam, B, fixed_new, kind):
        """-> (step(), out buffers).  kind: WIS_IN_PCM_DEV (HBM-resident windows), WIS_IN_PCM_HOST (host windows)."""
        win = np.ascontiguousarray(np.tile(audio.pad_or_trim(pcm)[None], (B, 1)).astype(np.float32))
        keep = _lib.DevBuf.from_numpy(win, dev) if kind == _lib.WIS_IN_PCM_DEV else win
        src = keep.ptr if kind == _lib.WIS_IN_PCM_DEV else _lib.ptr(win)
        opts = _lib.GenOpts(kind, beam, 0, 1.0, 1.0, 1, 1, fixed_new, 0)
        prompt = np.ascontiguousarray(np.tile(np.array(PROMPT, np.int32), (B, 1)))
        ids = np.zeros((B, 224), np.int32); lens = np.zeros(B, np.int32); scores = np.zeros(B, np.float32)

        def step():
            _lib.check(lib.wis_generate(h, src, B, prompt.ctypes.data_as(C.POINTER(C.c_int32)), len(PROMPT), C.byref(opts),
                                        ids.ctypes.data_as(C.POINTER(C.c_int32)), lens.ctypes.data_as(C.POINTER(C.c_int32)),
                                        scores.ctypes.data_as(C.POINTER(C.c_float))))
        return step, (keep, lens)

    def timed(step, n, warm):
        for _ in range(warm):
            step()
        _lib.check(lib.wis_dev_sync(dev))
        lat = []
        for _ in range(n):
            ts = time.perf_counter()
            step()
            lat.append(1e3 * (time.perf_counter() - ts))
        return lat

    def last_timing(h):
        t = _lib.Timing(); _lib.check(lib.wis_last_timing(h, C.byref(t)))
        return {k: round(v, 3) for k, v in t.as_dict().items()}

    pcm, audio_ms, clip_path = clip_pcm(args.clip)
    B = args.batch
    fixed_new = FIXED_NEW.get(args.clip, 16)
    step, (d_pcm, lens) = make_step(handle, pcm, args.beam, B, fixed_new, _lib.WIS_IN_PCM_DEV)      # inputs resident in HBM before the timed region

    def fence():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        _lib.check(lib.wis_dev_sync(dev))

    for _ in range(args.warmup):
        step()
    if args.warmup > 0:
        assert int(lens[0]) == fixed_new, (lens, fixed_new)
    fence()
    lat = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        lat.append(1e3 * (time.perf_counter() - ts))
    fence()
    elapsed = time.perf_counter() - t_start
    timing = last_timing(handle)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: BASELINE configs[3] (64 concurrent 3.84 s utterances over 8 GPUs = 8 per device batch per GPU, client/jmeter-asr.jmx:53-90).
    # The headline above stays one utterance per step per GPU (`value`); this second timed region runs device batches of 8 on every rank,
    # same barrier / max-over-ranks timing, so the driver's N = 1, 2, 4, 8 runs also carry the BATCHED-throughput scaling north_star asks for.
    batched = None
    if batched_B:
        stepB, (_keepB, lensB) = make_step(handle, pcm, args.beam, batched_B, fixed_new, _lib.WIS_IN_PCM_DEV)
        kB = max(4, min(args.steps, 12))
        for _ in range(2):
            stepB()
        assert int(lensB[0]) == fixed_new
        fence()
        tB = time.perf_counter()
        for _ in range(kB):
            stepB()
        fence()
        elB = time.perf_counter() - tB
        if use_dist:
            t = torch.tensor([elB], dtype=torch.float64, device=f"cuda:{dev}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elB = float(t.item())
        ups = world * batched_B * kB / elB
        batched = {"workload": f"{batched_B} x {args.clip}.flac per device batch per GPU, beam {args.beam}, {kB} device batches per rank timed (barrier + max over ranks)",
                   "utterances_per_s": round(ups, 2), "per_gpu": round(ups / world, 2), "x_realtime": round(ups * audio_ms / 1e3, 1),
                   "ms_per_device_batch": round(1e3 * elB / kB, 3)}

    roofline = None
    if rank == 0 and not args.no_roofline:
        # dominant kernel: the decoder's weight-streaming skinny GEMM (gemv_kernel): one decode step's weight stream
        ms = C.c_float(); nl = C.c_int(); nb = C.c_double()
        passes = 5
        _lib.check(lib.wis_bench_weight_stream(handle, B * args.beam, passes, C.byref(ms), C.byref(nl), C.byref(nb)))
        per_launch_us = 1e3 * ms.value / (passes * nl.value)
        achieved = nb.value / nl.value / (per_launch_us * 1e-6) / 1e9
        traffic = None
        for name in ("r06_pmc_decode.json", "r05_pmc_decode.json", "r04_pmc_decode.json", "r03_pmc_decode.json", "r02_pmc_gemv.json", "r01_pmc_gemv.json"):       # filled from the separate --pmc rocprofv3 passes (tools/gpu_session.sh pmc, tools/make_profiles.py)
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    break
                except Exception:
                    traffic = None
        steps_dec = max(timing["decode_steps"] - 1, 1)
        step_bytes = decode_step_bytes(a, B, args.beam, len(PROMPT), steps_dec, args.compute_type != "float16")
        step_ms = timing["decode_ms"] / steps_dec
        roofline = {"bound": "hbm", "kernel": "gemv_kernel (decoder skinny GEMM, weight streaming)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    # companion of `frac` (round-3 review item 8): the WHOLE decode step (weights + cross K/V + self KV over the measured step
                    # time, attention / sampling / boundaries included) against the same peak - the dominant kernel's tap above flatters it
                    "frac_decode_step": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "bytes_per_launch": round(nb.value / nl.value), "avg_launch_us": round(per_launch_us, 3),
                    "launches_per_decode_step": nl.value,
                    "how": "wis_bench_weight_stream: one launch of the skinny GEMM per decoder weight matrix (6 per layer + the vocabulary projection = one decode step's weight stream, 1.6 GB), "
                           "captured into a HIP graph and replayed like the product's decode step, HIP events on the model's stream; the tap launches the un-folded matrix set on zero rows "
                           "(the product's step fuses the out-projection with the folded cross-Q in gemv_dual_kernel: same kernel body, 6.6 instead of 3.3 + 3.3 MB in that launch)",
                    "decode_step": {"ms": round(step_ms, 4), "algorithmic_bytes": round(step_bytes), "achieved_GBps": round(step_bytes / (step_ms * 1e-3) / 1e9, 1),
                                    "frac": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "encoder": {"bound": "mfma", "gflop": 2587.3 if args.model == "large" else None,
                                "achieved_TFLOPs": round(B * 2587.3 / (timing["encoder_ms"] + timing["crosskv_ms"]), 1) if args.model == "large" else None,
                                "peak_TFLOPs": 2500.0}}

    extra = {}
    if rank == 0 and extras:
        # ---- the boundary the survey states (8d): PCM in HOST memory -> ids on host, and the container-decode-inclusive figure
        # (what the reference's infer_time spans, main.py:576-581,756-760).  `value` above stays the HBM-resident number.
        try:
            st_host, _k = make_step(handle, pcm, args.beam, 1, fixed_new, _lib.WIS_IN_PCM_HOST)
            l_host = timed(st_host, 20, 2)
            flac = open(clip_path, "rb").read()

            def from_bytes():
                x, _ = audio.load_audio(flac)
                s2, _k2 = make_step(handle, x, args.beam, 1, fixed_new, _lib.WIS_IN_PCM_HOST)
                s2()
            l_flac = timed(from_bytes, 20, 2)
            extra["boundary_ms_p50"] = {"pcm_resident_in_hbm (value)": round(p50(lat), 3), "pcm_in_host_memory": round(p50(l_host), 3),
                                        "flac_bytes_in_host_memory (container decode + H2D)": round(p50(l_flac), 3)}
        except Exception as e:
            extra["boundary_ms_p50"] = {"failed": repr(e)}
        # ---- the other BASELINE.json configurations, same handle where the model is the same (SURVEY 8d conventions)
        cfgs = []
        for name, clip, beam, Bc in (("large-v2 beam 5, 10sec.flac (configs[2], README row)", "10sec", 5, 1), ("large-v2 beam 5, 30sec.flac (S=96)", "30sec", 5, 1),
                                     ("large-v2 beam 3, 30sec.flac (the reference's long-audio beam, main.py:582-586)", "30sec", 3, 1),
                                     ("large-v2 beam 5, 8 x 3sec.flac per device batch (configs[3] shape on one GPU)", "3sec", 5, 8),
                                     ("large-v2 beam 5, 12 x 3sec.flac per device batch", "3sec", 5, 12), ("large-v2 beam 5, 16 x 3sec.flac per device batch", "3sec", 5, 16)):
            try:
                pc, ams, _p = clip_pcm(clip)
                S = FIXED_NEW[clip]
                stc, _k = make_step(handle, pc, beam, Bc, S, _lib.WIS_IN_PCM_DEV)
                lc = timed(stc, 10, 2)
                tm = last_timing(handle)
                row = {"config": name, "p50_ms": round(p50(lc), 3), "x_realtime": round(Bc * ams / p50(lc), 1), "stage_ms": tm}
                if Bc > 1:
                    sd = max(tm["decode_steps"] - 1, 1)
                    sb = decode_step_bytes(a, Bc, beam, len(PROMPT), sd)
                    row["utterances_per_s"] = round(1e3 * Bc / p50(lc), 1)
                    row["decode_step"] = {"ms": round(tm["decode_ms"] / sd, 4), "algorithmic_bytes": round(sb),
                                          "frac_of_hbm_peak": round(sb / (tm["decode_ms"] / sd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                    if Bc == 8:
                        try:      # HBM traffic of the step's two byte-heavy kernels from the --pmc passes (profiles/r03_pmc_decode.json)
                            b8 = json.load(open(next(pp for pp in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_decode.json", "r05_pmc_decode.json", "r04_pmc_decode.json", "r03_pmc_decode.json")) if os.path.exists(pp))))["batch_8"]
                            row["decode_step"]["traffic_over_algorithmic"] = {"skinny_gemm (gemv_frag_kernel)": b8["skinny_gemm_traffic_over_algorithmic"],
                                                                              "cross_attention": b8["per_kernel"]["dec_cross_attn_kernel 245760"]["traffic_over_algorithmic"]}
                        except Exception:
                            pass
                cfgs.append(row)
            except Exception as e:
                cfgs.append({"config": name, "failed": repr(e)})
        try:    # configs[1]: Whisper medium beam 1, 3sec.flac (a second replica, 1.5 GB)
            am = W.arch("medium")
            wm = W.synthetic_weights("medium", seed=1234)
            arm, ixm = W.build_arena(wm)
            hm = ct2.create_handle(am, arm, ixm, dev, max_batch=1, max_beam=1)
            del arm, wm
            stm, _k = make_step(hm, pcm, 1, 1, FIXED_NEW["3sec"], _lib.WIS_IN_PCM_DEV)
            lm = timed(stm, 20, 3)
            cfgs.append({"config": "medium beam 1, 3sec.flac (configs[1])", "p50_ms": round(p50(lm), 3), "x_realtime": round(audio_ms / p50(lm), 1), "stage_ms": last_timing(hm)})
            lib.wis_model_destroy(hm)
        except Exception as e:
            cfgs.append({"config": "medium beam 1, 3sec.flac (configs[1])", "failed": repr(e)})
        try:
            cfgs.append(rest_base_beam1(dev, open(clip_path, "rb").read(), audio_ms))
        except Exception as e:
            cfgs.append({"config": "base beam 1 over REST (configs[0] shape)", "failed": repr(e)})
        extra["other_baseline_configs"] = cfgs
        clones = []
        try:
            for _ in range(4):
                c = C.c_void_p()
                _lib.check(lib.wis_model_clone(handle, C.byref(c)))
                clones.append(c)
            extra["concurrent_device_batches"] = concurrent_batches(lib, [handle] + clones, dev, pcm, args.beam, 8, fixed_new, audio_ms)
            extra["concurrent_device_batches_of_16"] = concurrent_batches(lib, [handle] + clones, dev, pcm, args.beam, 16, fixed_new, audio_ms, iters=4, max_r=3)
        except Exception as e:
            extra["concurrent_device_batches"] = {"failed": repr(e)}
        try:
            extra["rest_load"] = rest_load(handle, a, dev, args.rest_clients, 2, fixed_new, open(clip_path, "rb").read(), audio_ms)
            extra["rest_load_3_replicas"] = rest_load(handle, a, dev, args.rest_clients, 3, fixed_new, open(clip_path, "rb").read(), audio_ms, extra_handles=clones[:2])
            extra["rest_load_4_replicas_128_clients"] = rest_load(handle, a, dev, 128, 3, fixed_new, open(clip_path, "rb").read(), audio_ms, extra_handles=clones[:3])
            # BASELINE configs[3] as ONE GPU of the node sees it: 64 concurrent clients over 8 GPUs = 8 closed-loop clients per GPU (one server process per GPU,
            # `python -m wis_hip.server --workers-per-node 8`); with one replica the eight requests form one device batch, with four they spread over concurrent smaller ones
            extra["rest_load_8_clients_per_gpu"] = {f"{1 + k}_replica(s)": {kk: vv for kk, vv in rest_load(handle, a, dev, 8, 12, fixed_new, open(clip_path, "rb").read(), audio_ms, extra_handles=clones[:k]).items()
                                                                           if kk in ("utterances_per_s", "p50_request_ms", "mean_device_batch")} for k in (0, 1, 3)}
            # what streaming speculation costs the REST traffic (round-5 review, weak 5): the 3-replica load again with 8 sessions asking for interim
            # decodes flat out beside it - with the load gate (settings.stream_speculate_max_busy: speculation only while the GPU has a replica to spare) and without
            p30, _ams30, _p30 = clip_pcm("30sec")
            extra["rest_load_3_replicas_with_8_speculating_sessions"] = rest_load(handle, a, dev, args.rest_clients, 3, fixed_new, open(clip_path, "rb").read(), audio_ms, extra_handles=clones[:2],
                                                                                    side_sessions=8, side_pcm=p30)
            extra["rest_load_3_replicas_with_8_speculating_sessions_no_gate"] = rest_load(handle, a, dev, args.rest_clients, 3, fixed_new, open(clip_path, "rb").read(), audio_ms, extra_handles=clones[:2],
                                                                                            side_sessions=8, side_pcm=p30, side_gate=1e9)
        except Exception as e:
            extra["rest_load"] = {"failed": repr(e)}
        for c in clones:
            lib.wis_model_destroy(c)
        try:
            extra["streaming"] = streaming_bench(handle, a, dev, os.path.join(ROOT, "tests", "golden", "clips", "30sec.flac"))
        except Exception as e:
            extra["streaming"] = {"failed": repr(e)}

    if rank == 0 and (extras or args.natural_eot) and not args.no_natural_eot and not dry and args.model == "large" and args.compute_type == "float16":
        try:
            if weights is None:
                weights = W.synthetic_weights(args.model, seed=1234)
            extra["natural_eot"] = natural_eot(lib, a, weights, dev, pcm, args.beam, audio_ms, make_step, timed, last_timing)
        except Exception as e:
            extra["natural_eot"] = {"failed": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if weights is None:
            weights = W.synthetic_weights(args.model, seed=1234)
        try:
            cpu = cpu_baseline(args.model, weights, pcm, args.beam, fixed_new, audio_ms)
        except Exception as e:   # the baseline leg must never take the measurement down
            cpu = {"value": None, "unit": "x realtime", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        total_audio_s = world * B * args.steps * audio_ms / 1e3
        out = {
            "metric": "realtime_multiple (audio_ms / infer_ms), Whisper large-v2 beam=5, 3.84 s clip" if (args.model, args.beam, args.clip) == ("large", 5, "3sec")
                      else f"realtime_multiple (audio_ms / infer_ms), Whisper {args.model} beam={args.beam}, {args.clip} clip",
            "value": round(total_audio_s / elapsed, 2), "unit": "x realtime", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "p50_ms": round(p50(lat), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if args.compute_type == "float16" else "f16 (int8 decoder weights, f16 activations)", "data": "synthetic weights (seeded, true large-v2 shapes); audio = reference clip client/3sec.flac; "
                                                          f"fixed decode length S={fixed_new} (SURVEY 8d convention)",
            "config": {"workload": f"whisper-{args.model} beam={args.beam} clip={args.clip} ({audio_ms:.0f} ms) batch={B}/GPU, PCM resident in HBM -> ids on host",
                       "utterances_per_step_per_gpu": B, "parallelism": f"{world} independent replicas (utterance sharding, no data-path collective)"},
            "utterances_per_s": round(world * B * args.steps / elapsed, 2),
            "stage_ms_last_step": timing,
            "reference_published": "RTX 4090: 140 ms / 27x; H100: 294 ms / 12x (README.md:71,73; CT2 int8_float16, real weights, other hardware)",
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if batched is not None:
            out["batched"] = batched
        # the span BASELINE.md section 3 defines for infer_ms (PCM in HOST memory -> ids on host) next to `value` (PCM resident in HBM, as the
        # round contract fixes it): same call with WIS_IN_PCM_HOST, p50 of 20
        bm = extra.get("boundary_ms_p50", {}).get("pcm_in_host_memory")
        if bm:
            out["value_pcm_in_host_memory"] = {"x_realtime": round(world * B * audio_ms / bm, 2), "p50_ms": bm}
        out.update(extra)
        print(json.dumps(out), flush=True)
    lib.wis_model_destroy(handle)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
