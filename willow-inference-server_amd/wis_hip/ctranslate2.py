"""Call-compatible stand-in for the slice of `ctranslate2` the reference uses:

    ctranslate2.models.Whisper(path, device=, compute_type=, inter_threads=, device_index=[..] | intra_threads=)
    ctranslate2.StorageView.from_array(ndarray f32 [B, 80, 3000])
    model.generate(features, [prompt] * B, beam_size=int, return_scores=False) -> results[i].sequences_ids[0]
    model.detect_language(features) -> [[("<|xx|>", prob), ...], ...]
    ctranslate2.get_supported_compute_types(device)

(reference main.py:341-444, 454, 535-537, 638-639, 685-693, 707, 713).  The arithmetic runs in the
hand-written HIP kernels of libwis_hip.so; there is no CPU fallback and no dependency on CTranslate2.

`model_path` is either a CTranslate2 Whisper model directory (model.bin [+ config.json]) or the
string "synthetic:<size>[:seed]" for seeded synthetic weights at the true shapes (no checkpoint exists
offline).
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib, weights as W
from .batching import MicroBatcher
from .languages import LANGUAGE_CODES


def get_supported_compute_types(device="cuda", device_index=0):
    buf = C.create_string_buffer(64)
    _lib.check(_lib.load().wis_supported_compute_types(device_index, buf, 64))
    return set(buf.value.decode().split(","))


class StorageView:
    """Zero-copy wrapper of a host ndarray (ctranslate2.StorageView.from_array, main.py:638,685)."""

    def __init__(self, array):
        self.array = array

    @classmethod
    def from_array(cls, array):
        a = np.asarray(array)
        if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
            raise ValueError("StorageView.from_array expects a C-contiguous float32 array")
        return cls(a)

    @property
    def shape(self):
        return list(self.array.shape)


class WhisperGenerationResult:
    def __init__(self, sequences_ids, scores, no_speech_prob=0.0):
        self.sequences_ids = sequences_ids
        self.sequences = [[str(t) for t in s] for s in sequences_ids]
        self.scores = scores
        self.no_speech_prob = no_speech_prob

    def __repr__(self):
        return f"WhisperGenerationResult(sequences_ids={self.sequences_ids}, scores={self.scores})"


class _Replica:
    def __init__(self, handle, device):
        self.handle, self.device = handle, device
        self.lock = threading.Lock()
        self.inflight = 0
        self.stage = None           # device block for batches of device-resident features (lazy)


def make_config(a, max_batch, max_beam, suppress_ids=None, suppress_begin=None, lang_ids=None, n_vocab=None, weight_bits=16):
    sup = np.asarray(W.SUPPRESS_IDS if suppress_ids is None else suppress_ids, np.int32)
    beg = np.asarray(W.SUPPRESS_IDS_BEGIN if suppress_begin is None else suppress_begin, np.int32)
    lang = np.asarray(W.LANG_IDS if lang_ids is None else lang_ids, np.int32)
    cfg = _lib.Config()
    cfg.d_model, cfg.n_heads = a["d_model"], a["n_heads"]
    cfg.n_enc_layers = cfg.n_dec_layers = a["n_layers"]
    cfg.n_vocab = n_vocab or a["n_vocab"]
    cfg.n_audio_ctx, cfg.n_text_ctx, cfg.n_mels = a["n_audio_ctx"], a["n_text_ctx"], a["n_mels"]
    cfg.max_batch, cfg.max_beam = max_batch, max_beam
    cfg.eot, cfg.sot, cfg.no_timestamps, cfg.no_speech = W.EOT, W.SOT, W.NO_TIMESTAMPS, W.NO_SPEECH
    cfg.suppress_ids = sup.ctypes.data_as(C.POINTER(C.c_int32)); cfg.n_suppress = len(sup)
    cfg.suppress_ids_begin = beg.ctypes.data_as(C.POINTER(C.c_int32)); cfg.n_suppress_begin = len(beg)
    cfg.lang_ids = lang.ctypes.data_as(C.POINTER(C.c_int32)); cfg.n_lang = len(lang)
    cfg.decoder_weight_bits = int(weight_bits)
    cfg._keep = (sup, beg, lang)
    return cfg


def make_tensor_index(index):
    arr = (_lib.Tensor * len(index))()
    keep = []
    for i, e in enumerate(index):
        nm = e["name"].encode()
        keep.append(nm)
        arr[i].name = nm
        arr[i].dtype = _lib.WIS_DT_F16 if e["dtype"] == "f16" else _lib.WIS_DT_F32
        arr[i].rank = len(e["shape"])
        for j, s in enumerate(e["shape"]):
            arr[i].shape[j] = s
        arr[i].offset = e["offset"]
    arr._keep = keep
    return arr


def create_handle(a, arena, index, device, max_batch=8, max_beam=5, arena_device_ptr=None, n_vocab=None, **cfgkw):
    """arena: uint8 ndarray (host) or None with arena_device_ptr = (ptr, nbytes) already on `device`."""
    cfg = make_config(a, max_batch, max_beam, n_vocab=n_vocab, **cfgkw)
    tens = make_tensor_index(index)
    h = C.c_void_p()
    if arena_device_ptr is not None:
        p, nbytes = arena_device_ptr
        rc = _lib.load().wis_model_create(C.byref(cfg), C.c_void_p(p), nbytes, 1, tens, len(index), device, C.byref(h))
    else:
        rc = _lib.load().wis_model_create(C.byref(cfg), _lib.ptr(arena), arena.nbytes, 0, tens, len(index), device, C.byref(h))
    _lib.check(rc)
    return h


def clone_handle(h):
    """Another replica on the same GPU sharing `h`'s weights (wis_model_clone): own stream, activations and KV caches."""
    c = C.c_void_p()
    _lib.check(_lib.load().wis_model_clone(h, C.byref(c)))
    return c


def create_replicas(a, arena, index, devices, max_batch=8, max_beam=5, **cfgkw):
    """One replica per entry of `devices` from ONE host upload: the arena goes to the first device over PCIe, every other
    replica receives it device-to-device (wis_dev_copy_peer: xGMI between peers) in a doubling tree - 0 -> 1, then {0 -> 2,
    1 -> 3}, ... - and is created from the device-resident copy (wis_model_create(arena_on_device=1)).  This is the
    single-process form of the one-time weight broadcast (the one-process-per-GPU form is wis_hip.dist.broadcast_arena over
    RCCL); nothing crosses GPUs at request time.  Returns the handles in `devices` order."""
    if len(devices) == 1:
        return [create_handle(a, arena, index, devices[0], max_batch, max_beam, **cfgkw)]
    lib = _lib.load()
    bufs = [None] * len(devices)
    handles = []
    try:
        bufs[0] = _lib.DevBuf.from_numpy(arena, devices[0])
        have = 1
        while have < len(devices):          # doubling: every device that holds the arena feeds one that does not
            for src in range(have):
                dst = have + src
                if dst >= len(devices):
                    break
                bufs[dst] = _lib.DevBuf(arena.nbytes, devices[dst])
                _lib.check(lib.wis_dev_copy_peer(devices[dst], bufs[dst].ptr, devices[src], bufs[src].ptr, arena.nbytes))
            have *= 2
        # every copy is done (hipMemcpyPeer is synchronous): each device's arena copy is released as soon as ITS replica has
        # re-packed it, so a device never holds more than its own arena copy + its model
        for i, d in enumerate(devices):
            handles.append(create_handle(a, None, index, d, max_batch, max_beam, arena_device_ptr=(bufs[i].ptr.value, arena.nbytes), **cfgkw))
            bufs[i].free()
            bufs[i] = None
    except BaseException:
        for h in handles:                   # a failed fan-out leaves nothing behind on any GPU
            lib.wis_model_destroy(h)
        raise
    finally:
        for b in bufs:
            if b is not None:
                b.free()
    return handles


def _last_trajectory(r, b, beam):
    """(tok [n][beam], org [n][beam]) int32: the live sets the last search on replica r left after each step (wis_last_trajectory)."""
    tok = np.zeros((256, beam), np.int32)
    org = np.zeros((256, beam), np.int32)
    n = C.c_int32(0)
    i32p = C.POINTER(C.c_int32)
    _lib.check(_lib.load().wis_last_trajectory(r.handle, b, tok.ctypes.data_as(i32p), org.ctypes.data_as(i32p), 256, C.byref(n)))
    return tok[:n.value].copy(), org[:n.value].copy()


def _generate_chunk(r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind, device_ptr=None, draft=None, want_traj=False):
    """mel: host ndarray [B, ...] - or, with device_ptr, just the batch size B of features already resident on r.device.
    draft (one utterance): token ids of an earlier hypothesis (beam 1: wis_generate_draft) or the trajectory (tok [n][beam], org [n][beam]) of
    an earlier beam search (wis_generate_draft_beam) - verified in multi-row passes before ordinary steps go on.
    want_traj: every result carries `.trajectory`, what a later call takes as its draft."""
    B = int(mel) if device_ptr is not None else mel.shape[0]
    o = _lib.GenOpts(kind, beam, max_new, lp, patience, int(bool(suppress_blank)), int(bool(suppress_default)), int(fixed_new), 0)
    pr = np.ascontiguousarray(np.asarray(prompts, np.int32).reshape(B, P))
    ids = np.zeros((B, max_new), np.int32)
    lens = np.zeros(B, np.int32)
    scores = np.zeros(B, np.float32)
    src = C.c_void_p(device_ptr) if device_ptr is not None else _lib.ptr(mel)
    i32p, lib = C.POINTER(C.c_int32), _lib.load()
    acc = None
    if draft is not None and len(draft) == 2 and isinstance(draft[0], np.ndarray):      # (tok, org): a beam search's trajectory
        dt, do = (np.ascontiguousarray(np.asarray(a, np.int32)) for a in draft)
        if dt.ndim != 2 or dt.shape != do.shape or dt.shape[1] != beam:
            raise ValueError(f"draft trajectory must be two [steps][{beam}] arrays, got {dt.shape} / {do.shape}")
        acc = C.c_int32(0)
        _lib.check(lib.wis_generate_draft_beam(r.handle, src, pr.ctypes.data_as(i32p), P, C.byref(o), dt.ctypes.data_as(i32p), do.ctypes.data_as(i32p), int(dt.shape[0]),
                                               ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), scores.ctypes.data_as(C.POINTER(C.c_float)), C.byref(acc)))
    elif draft is not None and len(draft):
        d = np.ascontiguousarray(np.asarray(draft, np.int32))
        acc = C.c_int32(0)
        _lib.check(lib.wis_generate_draft(r.handle, src, pr.ctypes.data_as(i32p), P, C.byref(o), d.ctypes.data_as(i32p), int(d.shape[0]),
                                          ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), scores.ctypes.data_as(C.POINTER(C.c_float)), C.byref(acc)))
    else:
        _lib.check(lib.wis_generate(r.handle, src, B, pr.ctypes.data_as(i32p), P, C.byref(o), ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                    scores.ctypes.data_as(C.POINTER(C.c_float))))
    out = [WhisperGenerationResult([ids[b, :lens[b]].tolist()], [float(scores[b])]) for b in range(B)]
    if acc is not None:
        out[0].accepted_draft_tokens = int(acc.value)      # tokens (beam 1) or search steps (beam > 1) of the draft the final decode kept
    if want_traj:
        for b in range(B):
            out[b].trajectory = _last_trajectory(r, b, beam)
    return out



MAX_DECODER_ROWS = 96      # csrc/kernels.hpp MAX_ROWS: decoder rows per device pass
MAX_BEAM = 8               # csrc/kernels.hpp MAX_R: rows per utterance (beam size)
MAX_REPLICAS_PER_DEVICE = 4   # default ceiling for inter_threads -> replicas per GPU (measured on MI355X, bench.py "concurrent_device_batches": 118 / 152 / 165 / 173 / 160 utterances/s with 1..5 batches of 8 in flight)
MAX_PROMPT = 16            # wis_generate: prompt tokens per utterance
MAX_HYPOTHESES = 24        # csrc/kernels.hpp MAX_HYP: finished hypotheses an utterance's search can hold


def _check_patience(beam_size, patience):
    """CTranslate2 searches until round(beam_size * patience) hypotheses have finished; the engine stores MAX_HYPOTHESES - beam_size + 1.
    A patience beyond that is a request error (it used to be clamped silently: a shorter search than CTranslate2's, other ids)."""
    p = float(patience) if float(patience) > 0 else 1.0
    if int(round(int(beam_size) * p)) > MAX_HYPOTHESES - int(beam_size) + 1:
        raise ValueError(f"patience {patience} at beam_size {beam_size} is beyond the engine's hypothesis storage (patience <= {(MAX_HYPOTHESES - int(beam_size) + 1) / int(beam_size):.3g} at this beam)")


def _capacity(max_batch, key):
    """Utterances per device batch: the decoder handles <= MAX_DECODER_ROWS rows per pass - utterances x beam while decoding,
    utterances x P in the merged prefill + first step (wis_generate enforces B * P <= MAX_ROWS and B * beam <= MAX_ROWS)."""
    P, beam = key[0], key[1]
    return max(1, min(max_batch, MAX_DECODER_ROWS // max(beam, 1), MAX_DECODER_ROWS // max(P, 1)))


_MEL_BYTES = 80 * 3000 * 4


def _run_batch(replica, key, rows):
    P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind = key[:9]
    draft = key[9] if len(key) > 9 else None          # (a drafted utterance has a key of its own: it never shares a device batch)
    want_traj = bool(key[10]) if len(key) > 10 else False
    dr = (draft[1] if draft and len(rows) == 1 else None)
    prompts = [p for _, p in rows]
    if kind == _lib.WIS_IN_MEL_DEV:
        # features that already live in this replica's HBM (streaming sessions: audio.MelStream.finish): one utterance goes in
        # by pointer, several are gathered device-to-device into the replica's staging block - nothing visits the host
        with replica.lock:
            if len(rows) == 1:
                ptr = int(rows[0][0])
            else:
                if replica.stage is None or replica.stage.nbytes < len(rows) * _MEL_BYTES:
                    replica.stage = _lib.DevBuf(max(len(rows), 8) * _MEL_BYTES, replica.device)
                lib = _lib.load()
                for i, (src, _) in enumerate(rows):
                    _lib.check(lib.wis_dev_copy_peer(replica.device, C.c_void_p(replica.stage.ptr.value + i * _MEL_BYTES), replica.device, C.c_void_p(int(src)), _MEL_BYTES))
                ptr = replica.stage.ptr.value
            return _generate_chunk(replica, len(rows), prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind, device_ptr=ptr,
                                   draft=dr, want_traj=want_traj)
    mel = np.ascontiguousarray(np.stack([m for m, _ in rows]))
    with replica.lock:
        return _generate_chunk(replica, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind,
                               draft=dr, want_traj=want_traj)


class Whisper:
    """One replica per entry of `device_index` (reference: `device_index=[*range(cuda_num_devices)]`, main.py:295)."""

    is_multilingual = True

    def __init__(self, model_path, device="cuda", device_index=0, compute_type="default", inter_threads=1, intra_threads=0,
                 max_batch=8, max_beam=5, weights=None, arch=None, replicas_per_device=None, **_ignored):
        if device not in ("cuda", "auto", "hip", "gpu"):
            raise ValueError(f"wis_hip runs on MI355X GPUs only (device={device!r}); there is no CPU path")
        _lib.require_gpu()
        # "float16" (default / "auto") or "int8_float16" (the reference's GPU default, main.py:242): per-row int8 DECODER
        # weights with f16 activations; everything else stays f16
        if compute_type in ("int8_float16", "int8"):
            self.compute_type = "int8_float16"
        elif compute_type in ("default", "auto", "float16", "float32"):
            self.compute_type = "float16"
        else:
            raise ValueError(f"unsupported compute_type {compute_type!r} (float16, int8_float16)")
        cfg = {}
        if weights is None:
            weights, arch, cfg = self._load(model_path)
        self.arch, self.decode_config = arch, cfg
        devs = list(device_index) if isinstance(device_index, (list, tuple)) else [int(device_index)]
        if not 1 <= int(max_batch) <= MAX_DECODER_ROWS:
            raise ValueError(f"max_batch={max_batch}: a device batch holds 1..{MAX_DECODER_ROWS} utterances")
        if not 1 <= int(max_beam) <= MAX_BEAM:
            raise ValueError(f"max_beam={max_beam}: the engine decodes with beam sizes 1..{MAX_BEAM}")
        arena, index = W.build_arena(weights)
        kw = dict(suppress_ids=cfg.get("suppress_ids"), suppress_begin=cfg.get("suppress_ids_begin"), lang_ids=cfg.get("lang_ids"),
                  weight_bits=8 if self.compute_type == "int8_float16" else 16)
        self._replicas = [_Replica(h, d) for h, d in zip(create_replicas(arch, arena, index, devs, max_batch, max_beam, **kw), devs)]
        # CTranslate2's `inter_threads` = batches a model runs in parallel (reference main.py:341-355 passes ctranslate2_threads).  Here:
        # replicas PER GPU that share one weight copy (wis_model_clone) and run their device batches concurrently on their own
        # streams - a decode chain is latency-bound and leaves most of the chip idle, a second batch in flight fills it.  The count is
        # bounded by `replicas_per_device` (each replica owns its activations and KV caches: ~0.7 GB per utterance slot for large-v2).
        per_dev = MAX_REPLICAS_PER_DEVICE if replicas_per_device is None else int(replicas_per_device)
        per_dev = max(1, min(int(inter_threads) if inter_threads else 1, per_dev))
        for r in list(self._replicas):
            for _ in range(per_dev - 1):
                self._replicas.append(_Replica(clone_handle(r.handle), r.device))
        self.max_batch, self.max_beam = max_batch, max_beam
        self._pick = threading.Lock()
        # concurrent generate() calls coalesce into device batches, one worker per GPU replica (wis_hip/batching.py)
        self._batcher = MicroBatcher(self._replicas, _run_batch, lambda key: _capacity(max_batch, key))

    @classmethod
    def from_handles(cls, handles, arch, max_batch=8, max_beam=5, decode_config=None):
        """Wrap replicas that already exist: [(wis_model handle, device), ...] - e.g. models created from a weight arena an
        RCCL broadcast delivered straight into device memory (`create_handle(arena_device_ptr=...)`, wis_hip/dist.py)."""
        self = cls.__new__(cls)
        self.compute_type = "float16"
        self.arch, self.decode_config = arch, decode_config or {}
        self._replicas = [_Replica(h, d) for h, d in handles]
        self.max_batch, self.max_beam = max_batch, max_beam
        self._pick = threading.Lock()
        self._batcher = MicroBatcher(self._replicas, _run_batch, lambda key: _capacity(max_batch, key))
        return self

    @staticmethod
    def _load(model_path):
        """'synthetic:<size>[:seed]' (no checkpoint offline), a CTranslate2 model directory (what WIS ships) or a
        Hugging Face safetensors checkpoint directory."""
        if isinstance(model_path, str) and model_path.startswith("synthetic:"):
            parts = model_path.split(":")
            size = parts[1]
            seed = int(parts[2]) if len(parts) > 2 else 1234
            return W.synthetic_weights(size, seed=seed), W.arch(size), {}
        if not os.path.isdir(model_path):
            raise FileNotFoundError(f"{model_path}: not a model directory (CTranslate2 or Hugging Face; or use 'synthetic:<size>')")
        return W.load_model_dir(model_path)

    def close(self):
        b = self.__dict__.pop("_batcher", None)
        if b is not None:
            b.close()

    def __del__(self):
        try:
            self.close()
            for r in getattr(self, "_replicas", []):
                if r.handle:
                    _lib.load().wis_model_destroy(r.handle)
                    r.handle = None
        except Exception:
            pass

    def _acquire(self):
        with self._pick:
            r = min(self._replicas, key=lambda x: x.inflight)
            r.inflight += 1
        return r

    def _release(self, r):
        with self._pick:
            r.inflight -= 1

    @staticmethod
    def _features(features, input_kind=_lib.WIS_IN_MEL_HOST):
        a = features.array if isinstance(features, StorageView) else np.asarray(features)
        if a.dtype != np.float32:
            a = a.astype(np.float32)
        a = np.ascontiguousarray(a)
        if input_kind == _lib.WIS_IN_PCM_HOST:
            if a.ndim != 2 or a.shape[1] != _lib.N_SAMPLES:
                raise ValueError(f"PCM input must be [batch, {_lib.N_SAMPLES}] float32 (pad_or_trim'ed 30 s windows), got {a.shape}")
        elif input_kind == _lib.WIS_IN_MEL_HOST:
            if a.ndim != 3 or a.shape[1:] != (80, 3000):
                raise ValueError(f"features must be [batch, 80, 3000] float32, got {a.shape}")
        else:
            raise ValueError("the Python face takes host arrays (WIS_IN_MEL_HOST or WIS_IN_PCM_HOST)")
        return a

    def generate(self, features, prompts, *, asynchronous=False, beam_size=5, patience=1, num_hypotheses=1, length_penalty=1,
                 repetition_penalty=1, no_repeat_ngram_size=0, max_length=448, return_scores=False, return_no_speech_prob=False,
                 max_initial_timestamp_index=50, suppress_blank=True, suppress_tokens=(-1,), sampling_topk=1,
                 sampling_temperature=1, fixed_new_tokens=0, input_kind=_lib.WIS_IN_MEL_HOST, draft_tokens=None, draft_trajectory=None,
                 return_trajectory=False):
        """`draft_tokens` (one utterance, beam_size 1): the ids of an earlier hypothesis for this audio - wis_generate_draft verifies them in
        multi-row passes and decodes on behind the accepted prefix; the result is the greedy decode of THESE features either way.
        `draft_trajectory` (one utterance, beam_size > 1): `(tok, org)` as an earlier result's `.trajectory` gives it (`return_trajectory=True`) -
        the beam search is replayed along it 16 steps per decoder pass (wis_generate_draft_beam); the result is the beam search of THESE features."""
        if num_hypotheses != 1 or repetition_penalty != 1 or no_repeat_ngram_size != 0 or sampling_topk != 1:
            raise NotImplementedError("only the decoding options WIS uses are implemented (defaults of CTranslate2 4.1.0)")
        mel = self._features(features, input_kind)
        B = mel.shape[0]
        if len(prompts) != B:
            raise ValueError("one prompt per batch item")
        P = len(prompts[0])
        if any(len(p) != P for p in prompts):
            raise ValueError("all prompts must have the same length")
        # the engine's capacity limits are request errors (ValueError -> HTTP 400 in wis_hip.server), not internal ones
        if not 1 <= int(beam_size) <= self.max_beam:
            raise ValueError(f"beam_size {beam_size} outside 1..{self.max_beam} (setting max_beam; engine ceiling {MAX_BEAM})")
        if not 1 <= P <= MAX_PROMPT:
            raise ValueError(f"prompt length {P} outside 1..{MAX_PROMPT}")
        _check_patience(beam_size, patience)
        max_new = min(max_length // 2, max_length - P)
        key = (P, int(beam_size), max_new, float(length_penalty), float(patience), bool(suppress_blank), list(suppress_tokens) == [-1],
               int(fixed_new_tokens), int(input_kind))
        key = key + self._draft_key(B, beam_size, draft_tokens, draft_trajectory, return_trajectory)
        rows = [(np.ascontiguousarray(mel[b]), [int(t) for t in prompts[b]]) for b in range(B)]
        return self._batcher.submit(key, rows)

    @staticmethod
    def _draft_key(B, beam_size, draft_tokens, draft_trajectory, return_trajectory):
        """The batcher-key tail of a drafted / trajectory-returning utterance: a draft makes the key unique (it never shares a device batch)."""
        d = None
        if B == 1 and int(beam_size) == 1 and draft_tokens is not None and len(draft_tokens):
            d = (object(), tuple(int(t) for t in draft_tokens))
        elif B == 1 and int(beam_size) > 1 and draft_trajectory is not None and len(draft_trajectory[0]):
            tok, org = (np.ascontiguousarray(np.asarray(a, np.int32)) for a in draft_trajectory)
            if tok.ndim == 2 and tok.shape[1] == int(beam_size) and tok.shape == org.shape:      # (a draft of another beam size cannot be followed: plain call)
                d = (object(), (tok, org))
        if d is None and not return_trajectory:
            return ()
        return (d, bool(return_trajectory))

    def load(self, device=None):
        """(utterances queued, device batches running[, on `device`]) in this model's micro-batcher"""
        return self._batcher.load(device)

    def replicas_on(self, device):
        return sum(1 for r in self._replicas if r.device == device)

    def acquire_replica(self):
        """The least-loaded replica, counted as in use until release_replica (a streaming session pins its log-mel front-end
        and its windows to ONE GPU for its lifetime; sessions spread over the replicas like requests do)."""
        return self._acquire()

    def release_replica(self, r):
        self._release(r)

    def replica_on(self, device):
        """A replica that lives on `device` (streaming sessions keep their features on one GPU)."""
        for r in self._replicas:
            if r.device == device:
                return r
        raise ValueError(f"no replica on device {device}")

    def generate_from_device(self, device, mel_device_ptr, prompt, *, beam_size=5, max_length=448, length_penalty=1, patience=1,
                             suppress_blank=True, fixed_new_tokens=0, replica=None, draft_tokens=None, draft_trajectory=None, return_trajectory=False):
        """One utterance whose log-mel features ALREADY live in HBM on `device` (f32 [80][3000] at `mel_device_ptr`, e.g. an
        audio.MelStream after finish()): WIS_IN_MEL_DEV - nothing is staged through the host.  Goes through the micro-batcher bound
        to that DEVICE: any replica of the GPU can read the features, so concurrent windows of several streaming sessions on one GPU
        coalesce into one device batch like REST requests do, whatever replica each session holds (round 4 bound a window to the
        session's own replica: sessions pinned to different replicas of a GPU never shared a batch)."""
        # `replica`: the replica the caller holds (a streaming session pins itself to the least-loaded one, acquire_replica: the load
        # accounting that spreads sessions over GPUs); it only has to live on the features' device
        r = replica if replica is not None else self.replica_on(device)
        if r.device != device:
            raise ValueError(f"replica lives on device {r.device}, the features on {device}")
        P = len(prompt)
        max_new = min(max_length // 2, max_length - P)
        if not 1 <= int(beam_size) <= self.max_beam:
            raise ValueError(f"beam_size {beam_size} outside 1..{self.max_beam}")
        key = (P, int(beam_size), max_new, float(length_penalty), float(patience), bool(suppress_blank), True, int(fixed_new_tokens),
               int(_lib.WIS_IN_MEL_DEV))
        _check_patience(beam_size, patience)
        key = key + self._draft_key(1, beam_size, draft_tokens, draft_trajectory, return_trajectory)
        return self._batcher.submit(key, [(int(mel_device_ptr), [int(t) for t in prompt])], affinity=("device", device))[0]

    def _generate_chunk(self, r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind):
        """One `wis_generate` call on replica r (the caller serialises access to r)."""
        return _generate_chunk(r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind)

    def detect_language(self, features, input_kind=_lib.WIS_IN_MEL_HOST):
        mel = self._features(features, input_kind)
        B = mel.shape[0]
        n_lang = len(W.LANG_IDS)
        out = []
        r = self._acquire()
        try:
            with r.lock:
                for s in range(0, B, self.max_batch):
                    m = mel[s:s + self.max_batch]
                    probs = np.zeros((m.shape[0], n_lang), np.float32)
                    m = np.ascontiguousarray(m)
                    _lib.check(_lib.load().wis_detect_language(r.handle, _lib.ptr(m), input_kind, m.shape[0],
                                                               probs.ctypes.data_as(C.POINTER(C.c_float))))
                    for row in probs:
                        order = np.argsort(-row, kind="stable")
                        out.append([(f"<|{LANGUAGE_CODES[i]}|>", float(row[i])) for i in order])
        finally:
            self._release(r)
        return out

    def handoff_state(self, raise_flag_on=None):
        """[(retries, spin_disabled)] per replica (wis_debug_handoff): calls that were repeated in the ticket form because the
        cross-attention's granule hand-off timed out - a request never fails for it.  raise_flag_on = replica index: raise the
        give-up flag there by hand (tests of the repeat path)."""
        out = []
        for i, r in enumerate(self._replicas):
            n, off = C.c_int(0), C.c_int(0)
            with r.lock:
                _lib.check(_lib.load().wis_debug_handoff(r.handle, int(raise_flag_on == i), C.byref(n), C.byref(off)))
            out.append((n.value, bool(off.value)))
        return out

    def last_timing(self, replica=0):
        t = _lib.Timing()
        _lib.check(_lib.load().wis_last_timing(self._replicas[replica].handle, C.byref(t)))
        return t.as_dict()


class models:  # namespace shim: `ctranslate2.models.Whisper`
    Whisper = Whisper
