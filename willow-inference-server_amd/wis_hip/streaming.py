"""Long-form / streaming ASR over the HIP path (BASELINE.json configs[4]).

The reference has no streaming inference: its WebRTC handler records the whole track and then makes one ordinary
`do_whisper` call (main.py:963-971).  Whisper's encoder is non-causal over its 30 s window, so what CAN be
done incrementally without changing results is the reference's own long-audio schedule (wis/audio.py:106-134): 22 s
windows with 4 s of context either side, stepping 14 s.  A `StreamingSession` accepts PCM as it arrives and, as soon as
the audio is known to need chunking (> 30 s buffered), transcribes every window whose 22 s are complete on a worker thread
(through the model's micro-batcher, so several sessions share device batches); `stop()` then only has the tail window left
and returns exactly what `do_whisper` returns for the complete recording.  A session with the incremental front-end is pinned
to ONE GPU for its lifetime - the least-loaded replica at creation (`Whisper.acquire_replica`), so sessions spread over the
replicas - because its features live in that GPU's HBM; its windows still go through the micro-batcher (replica affinity,
`Whisper.generate_from_device`), so concurrent sessions on a GPU coalesce into device batches.  `interim()` gives the hypothesis for what has
been heard so far (an extra decode; the final result never depends on it).  The log-mel front-end is incremental
(`_IncrementalFront`, C-ABI wis_melstream_*): each window's spectrogram tiles are computed while its audio arrives, only the
clamp / scaling is left when the window completes, and the features go to the encoder straight from HBM.

`DataChannelProtocol` is the reference's data-channel message protocol (`ping` / `start` / `stop` -> `pong` / `log` /
`infer` / `error`, main.py:906-996) driven by such a session instead of `MediaRecorderLite` + `do_whisper`; the WebRTC
transport itself (aiortc) is not available here and stays out of scope.
"""
import json
import math
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import audio, ctranslate2, weights as W
from .whisper import InvalidAudio, WhisperResult, _Tokenizer, check_language, default_models

_STEP = audio.chunk_len - audio.stride_left - audio.stride_right     # 14 s between window starts


class _IncrementalFront:
    """The log-mel front-end of a session, fed as the audio arrives: one audio.MelStream per 30 s window in
    progress - the "short" window [0 s, 30 s) that a recording of up to 30 s ends up using, and the 22 s windows of the chunk
    schedule (starts 0, 14, 28, ... s).  When a window is complete only its tail tiles and the clamp are left to compute, and the
    features are already in HBM on the replica's GPU (generate_from_device: nothing is staged through the host)."""

    def __init__(self, device):
        self.device = device
        self.short = audio.MelStream(device)
        self.windows = {}             # window start (samples) -> MelStream
        self.n = 0

    def feed(self, x, chunking):
        n0, n1 = self.n, self.n + x.shape[0]
        if self.short is not None:
            if n0 < audio.N_SAMPLES:
                self.short.feed(x[:audio.N_SAMPLES - n0])
            if n1 > audio.N_SAMPLES and chunking:          # the recording will be chunked: the short window is never used
                self.short.close()
                self.short = None
        if chunking:
            start = 0
            while start < n1:
                if start + audio.chunk_len > n0:           # window [start, start + 22 s) overlaps the new samples
                    st = self.windows.get(start)
                    if st is None:
                        st = self.windows[start] = audio.MelStream(self.device)
                        if n0 > start:                     # opened late (cannot happen when fed from the beginning)
                            raise RuntimeError("incremental front-end must see the recording from its first sample")
                    lo, hi = max(start, n0), min(start + audio.chunk_len, n1)
                    st.feed(x[lo - n0:hi - n0])
                start += _STEP
        self.n = n1

    def take(self, start):
        return self.windows.pop(start, None)

    def close(self):
        for st in list(self.windows.values()) + ([self.short] if self.short is not None else []):
            st.close()
        self.windows, self.short = {}, None


class StreamingSession:
    def __init__(self, model, beam_size=None, task="transcribe", detect_language=False, force_language=None, models=None,
                 fixed_new_tokens=0, incremental=True, speculate_every_s=None):
        self.models = models or default_models()
        s = self.models.settings
        self.model_name, self.task = model, task
        self.beam_size = s.beam_size if beam_size is None else beam_size
        self.detect_language, self.force_language = detect_language, force_language
        self.fixed_new_tokens = fixed_new_tokens
        if force_language and not check_language(force_language):
            raise ValueError(f"unsupported language {force_language!r}")
        self._whisper = self.models.get(model)
        self._chunks, self._n = [], 0   # PCM as it arrived; consolidated lazily (_audio) when a window is cut, not once per frame
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="wis-stream")
        self._windows = {}            # (start, length) -> Future[list[int]]   window token ids, computed eagerly
        self._language_job = None     # Future: language of the chunked recording (from window 0), resolved exactly once
        self._closed = False
        # speculative interim decodes of a recording that is still short enough for ONE window (<= 30 s): see _maybe_speculate
        self._spec_every = float(s.stream_speculate_s if speculate_every_s is None else speculate_every_s)
        self._spec_n, self._spec_job, self._spec_latest = 0, None, None      # samples covered when the last one was scheduled; its Future; (samples, beam, tokens, trajectory)
        self._spec_busy = float(getattr(s, "stream_speculate_max_busy", 0.5))
        self.spec_runs = 0            # interim decodes done while the audio arrived (stats / tests)
        self.spec_skipped = 0         # ... not started because the GPU had no replica to spare
        self.spec_ms = 0.0            # wall time of the interim decodes: what speculation costs the GPU per session
        self.accepted_draft_tokens = None     # tokens (beam 1) / search steps (beam > 1) of the last interim hypothesis the final decode kept (None: no draft was used)
        self.eager_windows = 0        # windows transcribed before stop() (stats / tests)
        self.front_windows = 0        # windows whose features came from the incremental front-end
        # the session's GPU: the least-loaded replica now, held (counted in its load) until close()
        self._replica = self._whisper.acquire_replica() if incremental else None
        self._front = _IncrementalFront(self._replica.device) if incremental else None

    def _audio(self):
        """Everything received so far as one array (lock held).  The chunk list collapses into that array, so the cost is paid
        once per cut window / interim / stop, not once per 20 ms frame."""
        if len(self._chunks) != 1:
            self._chunks = [np.concatenate(self._chunks) if self._chunks else np.zeros(0, np.float32)]
        return self._chunks[0]

    # ---- audio in ----------------------------------------------------------------------------------------------------
    def feed(self, samples, sample_width=None):
        """float32 PCM in [-1, 1) at 16 kHz, or raw little-endian int16 bytes (`sample_width=2`, a Willow / WebRTC frame)."""
        if self._closed:
            raise RuntimeError("session is closed")
        if isinstance(samples, (bytes, bytearray, memoryview)):
            if sample_width not in (None, 2):
                raise InvalidAudio("only 16-bit PCM frames are supported")
            samples = np.frombuffer(bytes(samples), "<i2").astype(np.float32) / 32768.0
        x = np.ascontiguousarray(samples, np.float32).reshape(-1)
        with self._lock:
            self._chunks.append(x.copy())
            self._n += x.shape[0]
            front = getattr(self, "_front", None)
            if front is not None:
                front.feed(x, self.models.settings.support_chunking)
            self._schedule_complete_windows()
            self._maybe_speculate()

    @property
    def buffered_ms(self):
        return int(self._n / audio.SAMPLE_RATE * 1000)

    # ---- scheduling --------------------------------------------------------------------------------------------------
    def _prompt(self, language):
        task_id = W.TRANSLATE if self.task == "translate" else W.TRANSCRIBE
        return [W.SOT, _Tokenizer.language_token_id(language), task_id, W.NO_TIMESTAMPS]

    def _detect(self, first_window):
        """What do_whisper does on `mel_features[0:1]` (main.py:637-643): language of the FIRST window only."""
        s = self.models.settings
        language = s.language
        if self.detect_language and not self.force_language:
            x = np.ascontiguousarray(audio.pad_or_trim(first_window)[None], np.float32)
            res = self._whisper.detect_language(ctranslate2.StorageView.from_array(x), input_kind=ctranslate2._lib.WIS_IN_PCM_HOST)
            language = res[0][0][0].strip("<|>")
        elif self.force_language:
            language = self.force_language
        if not check_language(language):
            raise ValueError(f"unsupported language {language!r}")
        return language

    def _final_beam(self, n_samples):
        s = self.models.settings
        return s.long_beam_size if int(n_samples / audio.SAMPLE_RATE * 1000) >= s.long_beam_size_threshold else self.beam_size

    def _maybe_speculate(self):
        """Called with the lock held.  A recording of up to 30 s is ONE window whose encoder needs the whole audio, so nothing of the
        FINAL answer can be computed early - but its decode, 80-95 % of the time to the answer, can be PREPARED: every `_spec_every`
        seconds of new audio the audio so far is decoded (at the beam the final call would use at this length; on whatever replica is
        free) and the hypothesis kept - the token chain at beam 1, the search's trajectory at beam > 1.  stop() hands the latest one to
        the final decode as a draft (wis_generate_draft / wis_generate_draft_beam): the final window's encoder runs, the draft is
        verified against it 16 steps per decoder pass, and step-by-step decoding only resumes where the two part - typically the last
        words.  Only for a language that needs no detection on the final audio, and only while the GPU has a replica to spare."""
        if getattr(self, "_spec_every", 0) <= 0 or self._n > 30 * audio.SAMPLE_RATE or self._n - self._spec_n < self._spec_every * audio.SAMPLE_RATE:
            return
        if self.detect_language and not self.force_language:
            return
        if self._final_beam(self._n) > 1 and not getattr(self.models.settings, "stream_speculate_beam_search", False):
            return
        if self._spec_job is not None and not self._spec_job.done():
            return
        dev = self._replica.device if self._replica is not None else None
        queued, running = self._whisper.load(dev)
        spare = self._whisper.replicas_on(dev) if dev is not None else len(self._whisper._replicas)
        if self._spec_busy < 1e6 and (queued > 0 or running > self._spec_busy * spare):      # optional work never queues behind (or in front of) real requests (>= 1e6: gate off, A/B)
            self.spec_skipped += 1
            self._spec_n = self._n
            return
        pcm, n = self._audio().copy(), self._n
        self._spec_n = n
        self._spec_job = self._pool.submit(self._speculate, pcm, n, self._final_beam(n), self._spec_latest)

    @staticmethod
    def _draft_for(beam, latest):
        """keyword arguments that hand `latest` = (samples, beam, tokens, trajectory) to a decode at `beam` as its draft ({} when it cannot be one)"""
        if not latest or latest[1] != beam:
            return {}
        if beam == 1:
            return {"draft_tokens": latest[2]} if latest[2] else {}
        return {"draft_trajectory": latest[3]} if latest[3] is not None and len(latest[3][0]) else {}

    def _speculate(self, pcm, n, beam, prev):
        language = self.force_language or self.models.settings.language
        t0 = time.perf_counter()
        ids, _, traj = self._window_decode(pcm, beam, language, draft=self._draft_for(beam, prev), want_traj=beam > 1)      # (the previous hypothesis is the draft of this one)
        with self._lock:
            if self._spec_latest is None or n > self._spec_latest[0]:
                self._spec_latest = (n, beam, list(ids), traj)
            self.spec_runs += 1
            self.spec_ms += 1e3 * (time.perf_counter() - t0)
        return ids

    def _window_tokens(self, piece, beam, language, stream=None):
        return self._window_decode(piece, beam, language, stream)[0]

    def _window_decode(self, piece, beam, language, stream=None, draft=None, want_traj=False):
        """-> (token ids, draft tokens / steps the decode kept or None, the search's trajectory or None).
        `language`: a code, or a Future resolving to one (the session-wide language job of window 0: every eager window
        waits for THAT result, so a later window can never decide the language - do_whisper always detects on window 0).
        `stream`: the window's MelStream (its samples are all fed): finish it and decode from the features in HBM.
        `draft`: {} or the keyword argument (`draft_tokens` / `draft_trajectory`) of an earlier hypothesis for this window."""
        if hasattr(language, "result"):
            language = language.result()
        kw = dict(draft or {})
        if want_traj:
            kw["return_trajectory"] = True
        if stream is not None:
            try:
                stream.finish(to_host=False)
                r = self._whisper.generate_from_device(stream.device, stream.device_ptr, self._prompt(language), beam_size=beam,
                                                       fixed_new_tokens=self.fixed_new_tokens, replica=self._replica, **kw)
                self.front_windows += 1
            finally:
                stream.close()
        else:
            x = np.ascontiguousarray(audio.pad_or_trim(piece)[None], np.float32)
            r = self._whisper.generate(ctranslate2.StorageView.from_array(x), [self._prompt(language)], beam_size=beam,
                                       return_scores=False, fixed_new_tokens=self.fixed_new_tokens, input_kind=ctranslate2._lib.WIS_IN_PCM_HOST, **kw)[0]
        return r.sequences_ids[0], getattr(r, "accepted_draft_tokens", None), getattr(r, "trajectory", None)

    def _schedule_complete_windows(self):
        """Called with the lock held.  Once more than 30 s are buffered the final call WILL chunk (main.py:588), with the
        long-audio beam (>= 12 s, main.py:582-586); every window that already has its full 22 s has its final content."""
        s = self.models.settings
        n = self._n
        if not s.support_chunking or n <= 30 * audio.SAMPLE_RATE:
            return
        start = 0
        while start + audio.chunk_len <= n:
            key = (start, audio.chunk_len)
            if key not in self._windows:
                piece = self._audio()[start:start + audio.chunk_len].copy()
                if self._language_job is None:         # start == 0 here: the chunked call detects on exactly this window
                    self._language_job = self._pool.submit(self._detect, piece)
                front = getattr(self, "_front", None)
                args = (piece, s.long_beam_size, self._language_job) + ((front.take(start),) if front is not None else ())
                self._windows[key] = self._pool.submit(self._window_tokens, *args)
                self.eager_windows += 1
            start += _STEP

    # ---- results -----------------------------------------------------------------------------------------------------
    def _transcribe(self, final):
        t0 = time.perf_counter()
        with self._lock:
            pcm = self._audio().copy()
        if pcm.shape[0] == 0:
            raise InvalidAudio("empty audio")
        s = self.models.settings
        duration_ms = int(pcm.shape[0] / audio.SAMPLE_RATE * 1000)
        beam = s.long_beam_size if duration_ms >= s.long_beam_size_threshold else self.beam_size
        tokenizer = self.models.tokenizer_for(self.model_name)
        front = getattr(self, "_front", None)
        if duration_ms > 30 * 1000 and s.support_chunking:
            with self._lock:
                if self._language_job is None:     # chunking disabled while feeding, or a burst longer than 30 s fed at once
                    self._language_job = self._pool.submit(self._detect, pcm[:audio.chunk_len].copy())
            language = self._language_job.result()
            seqs = []
            for piece, stride in audio.chunk_iter(pcm):
                start = len(seqs) * _STEP
                fut = self._windows.get((start, piece.shape[0]))
                if fut is not None:
                    ids = fut.result()
                else:      # a tail window: at stop() its stream holds everything but the last tiles
                    st = front.take(start) if (final and front is not None and front.n == pcm.shape[0]) else None
                    ids = self._window_tokens(piece, beam, language, st) if st is not None else self._window_tokens(piece, beam, language)
                seqs.append((ids, stride))
            tokens = [int(t) for t in audio.find_longest_common_sequence(seqs, tokenizer)]
        else:
            language = self._detect(pcm)           # short recording: one window = the whole audio; nothing is cached, a later
            st = None                              # (longer) call detects again on its own first window
            if final and front is not None and front.short is not None and front.n == pcm.shape[0]:
                st, front.short = front.short, None
            draft = {}
            if final:      # the latest COMPLETED interim hypothesis (one still running is not waited for: its decode is what stop() is here to avoid)
                with self._lock:
                    draft = self._draft_for(beam, getattr(self, "_spec_latest", None))
            tokens, accepted, _ = self._window_decode(pcm, beam, language, st, draft=draft)
            if final and draft:
                self.accepted_draft_tokens = accepted
        text = tokenizer.decode(tokens).strip()
        ms = (time.perf_counter() - t0) * 1000
        out = WhisperResult((language, text, ms, None, math.floor(duration_ms / ms) if ms > 0 else 0, duration_ms))
        out.tokens = tokens
        return out

    def interim(self):
        """Hypothesis for the audio received so far (does not end the session)."""
        return self._transcribe(final=False)

    def stop(self):
        """End of the recording: the same 6-tuple `do_whisper` returns for the complete audio (infer_time counts only the work
        left at stop time - the windows transcribed while the audio was arriving are already done)."""
        job = getattr(self, "_spec_job", None)
        if job is not None and not job.done():
            job.cancel()          # not started yet: never will be (one that is running finishes beside the final decode; its result is not waited for)
        try:
            return self._transcribe(final=True)
        finally:
            self.close()

    def close(self):
        self._closed = True
        self._pool.shutdown(wait=False, cancel_futures=True)
        front = getattr(self, "_front", None)
        if front is not None:
            front.close()
        r, self._replica = getattr(self, "_replica", None), None
        if r is not None:
            self._whisper.release_replica(r)


class DataChannelProtocol:
    """The reference's WebRTC data-channel protocol (main.py:906-996) as a transport-free state machine: feed it the JSON text
    messages and the decoded audio frames, send back the JSON strings it returns."""

    def __init__(self, models=None, model=None, beam_size=None, task="transcribe", detect_language=None, session_factory=StreamingSession):
        self.models = models or default_models()
        s = self.models.settings
        self.model = model or s.whisper_model_default
        self.beam_size = s.beam_size if beam_size is None else beam_size
        self.detect_language = s.detect_language if detect_language is None else detect_language
        self.task = task
        self._factory = session_factory
        self._frames = None           # list of PCM frames while recording

    @staticmethod
    def _msg(type_, message=None, obj=None):
        return json.dumps({"type": type_, "message": message, "obj": obj})

    def on_audio(self, frame, sample_width=2):
        if self._frames is not None:
            self._frames.append((frame, sample_width))

    def on_message(self, message):
        if not isinstance(message, str):
            return []
        try:
            m = json.loads(message)
            mtype, mmsg, mobj = m["type"], m.get("message"), m.get("obj")
        except Exception:
            return [self._msg("error", "could not parse message")]
        if mtype == "ping":
            return [self._msg("pong", mmsg)]
        if mtype == "start":
            self._frames = []
            return [self._msg("log", "ASR Recording - start talking and press stop when done")]
        if mtype == "stop":
            if self._frames is None:
                return [self._msg("error", "Recording not yet started")]
            obj = mobj or {}
            model = obj.get("model") or self.model
            beam_size = obj.get("beam_size") or self.beam_size
            detect_language = obj.get("detect_language") or self.detect_language
            out = [self._msg("log", f"Doing ASR with model {model} beam size {beam_size} detect language {detect_language} - please wait")]
            frames, self._frames = self._frames, None
            # model / beam are only known at "stop" (per-message overrides, main.py:940-943), so the session starts here and
            # the recorded frames are replayed into it
            try:
                sess = self._factory(model, beam_size, self.task, detect_language, None, models=self.models)
                for frame, width in frames:
                    sess.feed(frame, width)
                language, text, infer_time, translation, infer_speedup, audio_duration = sess.stop()
            except (InvalidAudio, ValueError) as e:
                return out + [self._msg("error", str(e))]
            out.append(self._msg("infer", obj=dict(text=text)))
            if translation:
                out.append(self._msg("log", f"ASR Translation from {language}:  {translation}"))
            out += [self._msg("log", f"ASR Infer time: {infer_time} ms"), self._msg("log", f"ASR Audio Duration: {audio_duration} ms"),
                    self._msg("log", f"ASR Speedup: {infer_speedup}x faster than realtime")]
            return out
        return [self._msg("error", f'unknown message type "{mtype}"')]
