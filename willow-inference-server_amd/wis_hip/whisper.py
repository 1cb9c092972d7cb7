"""The ASR orchestrator: what the reference's `do_whisper` does (main.py:554-770), on the wis_hip engine.

Same call signature, same per-request selection surface (model in {tiny, base, small, medium, large}, beam_size,
detect_language, force_language, translate; main.py:564-573) and the same 6-tuple result
    (language, text, infer_time_ms, translation, infer_speedup, audio_duration_ms)          (main.py:763-770)
including the reference's behaviours: >= long_beam_size_threshold ms switches to long_beam_size (main.py:582-586),
> 30 s is chunked into 22 s windows with 4 s context, decoded `concurrent_gpu_chunks` at a time and stitched with
find_longest_common_sequence (main.py:588-611, 677-711).

No tokenizer files exist offline (the reference loads HF WhisperProcessor from the model dir, main.py:329-334):
prompt ids are the fixed multilingual ids; `text` is produced by an optional tokenizer
(`tokenizers` JSON next to the model) and otherwise is the space-joined token ids.  The returned tuple also
carries `.tokens`.
"""
import math
import os
import threading
import time

import numpy as np

from . import audio, ctranslate2, weights as W
from .languages import LANGUAGE_CODES, LANGUAGES
from .settings import get_api_settings

MODEL_SIZES = ("tiny", "base", "small", "medium", "large")
SPECIAL_IDS = list(range(W.EOT, W.N_VOCAB))     # <|endoftext|> ... timestamps: everything >= 50257 is special


class WhisperResult(tuple):
    """6-tuple like the reference's return value, plus the raw token ids."""
    tokens = None
    translation_tokens = None


def special_ids_from_tokenizer_json(path):
    """`tokenizer.all_special_ids` as HF computes it for a fast tokenizer: the ids of the `added_tokens` entries flagged
    `special: true` in tokenizer.json (for the Whisper checkpoints: <|endoftext|>, <|startoftranscript|>, the language and
    task tokens, <|nospeech|>, <|notimestamps|> - NOT the <|0.00|>... timestamp tokens, which are added but not special).
    The reference strips exactly this list before stitching windows (wis/audio.py:141-146)."""
    import json
    with open(path, "r", encoding="utf-8") as f:
        tj = json.load(f)
    return sorted({int(t["id"]) for t in tj.get("added_tokens", []) if t.get("special")})


class _Tokenizer:
    # id-only form (no tokenizer files, synthetic weights): every id from <|endoftext|> up is treated as special
    all_special_ids = SPECIAL_IDS

    def __init__(self, path=None):
        self._tok = None
        if path and os.path.exists(os.path.join(path, "tokenizer.json")):
            from tokenizers import Tokenizer
            self._tok = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))
            ids = special_ids_from_tokenizer_json(os.path.join(path, "tokenizer.json"))
            if ids:                     # the checkpoint's own list, as `WhisperProcessor.tokenizer.all_special_ids` (wis/audio.py:141)
                self.all_special_ids = ids

    @property
    def has_vocabulary(self):
        return self._tok is not None

    def decode(self, ids):
        """`WhisperProcessor.decode(tokens)` (main.py:714): the ids generate returns carry no special tokens besides what the
        model emitted itself; they are kept (skip_special_tokens defaults to False there too)."""
        ids = [int(t) for t in ids]
        if self._tok is not None:
            return self._tok.decode(ids, skip_special_tokens=False)
        return " ".join(str(t) for t in ids)

    @staticmethod
    def language_token_id(code):
        return W.LANG_IDS[LANGUAGE_CODES.index(code)]


class WhisperModels:
    """Lazy per-size registry (reference `LazyModels`, main.py:319-448): a model is built on first use; preload_* /
    warm-up mirror load_models / warm_models (main.py:451-511)."""

    def __init__(self, settings=None, device_index=None):
        self.settings = settings or get_api_settings()
        self._models, self._lock = {}, threading.Lock()
        n = ctranslate2._lib.device_count()
        self.device_index = list(range(n)) if device_index is None else list(device_index)
        self.tokenizer = _Tokenizer(None)      # id-only tokenizer (synthetic weights); real checkpoints get their own below
        self.tokenizers = {}

    def path_for(self, size):
        return self.settings.whisper_model_path.format(size=size)

    def get(self, size):
        if size not in MODEL_SIZES:
            raise ValueError(f"unknown model {size!r}")
        with self._lock:
            if size not in self._models:
                path = self.path_for(size)
                synthetic = path.startswith("synthetic:")
                if not synthetic:
                    if not os.path.isdir(path):
                        raise FileNotFoundError(
                            f"Whisper model directory {path!r} not found (setting whisper_model_path; the reference layout is "
                            "models/tovera-wis-whisper-<size>).  Seeded synthetic weights are only served when asked for explicitly: "
                            "whisper_model_path=synthetic:{size}")
                    tok = _Tokenizer(path)
                    if not tok.has_vocabulary and not self.settings.allow_token_id_text:
                        raise FileNotFoundError(f"{path}: no tokenizer.json - text output needs the checkpoint's tokenizer (the reference "
                                                "loads WhisperProcessor from the model dir, main.py:329-334); set allow_token_id_text=1 to serve "
                                                "token ids as text")
                    self.tokenizers[size] = tok
                max_beam = min(max(int(self.settings.max_beam), int(self.settings.beam_size), int(self.settings.long_beam_size)), ctranslate2.MAX_BEAM)
                self._models[size] = ctranslate2.models.Whisper(path, device="cuda", compute_type=self.settings.compute_type,
                                                                inter_threads=self.settings.ctranslate2_threads,
                                                                device_index=self.device_index, max_batch=self.settings.max_batch,
                                                                replicas_per_device=self.settings.replicas_per_gpu, max_beam=max_beam)
                import logging
                logging.getLogger("wis_hip").info("whisper %s loaded: beam_size 1..%d served (max_beam), device batches of up to %d utterances, %d replica(s)",
                                                  size, max_beam, self.settings.max_batch, len(self._models[size]._replicas))
            return self._models[size]

    def tokenizer_for(self, size):
        return self.tokenizers.get(size, self.tokenizer)

    def preload(self):
        s = self.settings
        for size in MODEL_SIZES:
            if s.preload_all_models or getattr(s, f"preload_whisper_model_{size}"):
                self.get(size)

    def warm(self, clip):
        for _ in range(3):
            for size in list(self._models):
                do_whisper(clip, size, self.settings.beam_size, "transcribe", False, "en", models=self)


_default_models = None


def default_models():
    global _default_models
    if _default_models is None:
        _default_models = WhisperModels()
    return _default_models


class InvalidAudio(ValueError):
    """The container could not be decoded (the REST layer answers HTTP 400 "Invalid audio", main.py:1311-1314)."""


def check_language(language):
    return language in LANGUAGES


def chunkit(lst, num):
    for i in range(0, len(lst), num):
        yield lst[i:i + num]


def do_whisper(audio_file, model, beam_size=None, task="transcribe", detect_language=False, force_language=None, translate=False,
               models=None, fixed_new_tokens=None):
    models = models or default_models()
    s = models.settings
    if fixed_new_tokens is None:
        fixed_new_tokens = s.fixed_new_tokens
    beam_size = s.beam_size if beam_size is None else beam_size
    whisper_model = models.get(model)
    first_time_start = time.perf_counter()

    # STEP 1 — load audio and extract features
    if isinstance(audio_file, np.ndarray):
        pcm, sr = audio_file.astype(np.float32), 16000
    else:
        try:
            pcm, sr = audio.load_audio(audio_file)
        except Exception as e:
            raise InvalidAudio(str(e)) from e
    if pcm.shape[0] == 0:
        raise InvalidAudio("empty audio")
    audio_duration = int(pcm.shape[0] / sr * 1000)
    if audio_duration >= s.long_beam_size_threshold:
        beam_size = s.long_beam_size
    use_chunking = audio_duration > 30 * 1000 and s.support_chunking
    strides = []
    if use_chunking:
        windows = []
        for chunk, stride in audio.chunk_iter(pcm):
            windows.append(audio.pad_or_trim(chunk))
            strides.append(stride)
        windows = np.stack(windows)
    else:
        windows = audio.pad_or_trim(pcm)[None]
    if s.fuse_logmel:
        # the 30 s PCM windows go to the replica as they are: log-mel runs on THAT replica's GPU inside generate and the
        # features never leave HBM (WIS_IN_PCM_HOST) - same kernels, same results as the two-step form below
        features, kind = np.ascontiguousarray(windows, np.float32), ctranslate2._lib.WIS_IN_PCM_HOST
    else:
        # the reference's two-step form (main.py:606-614, 685): features to the host, then StorageView.from_array
        # (on one of the model's own GPUs - never on a device the server was not configured to use)
        dev = whisper_model._replicas[0].device if getattr(whisper_model, "_replicas", None) else None
        features, kind = audio.log_mel_spectrogram(windows, device=dev).numpy(), ctranslate2._lib.WIS_IN_MEL_HOST
    total_chunk_count = features.shape[0]
    tokenizer = models.tokenizer_for(model)

    # STEP 2 — language
    language = s.language
    if detect_language and not force_language:
        results = whisper_model.detect_language(ctranslate2.StorageView.from_array(np.ascontiguousarray(features[0:1])), input_kind=kind)
        lang_token, _probability = results[0][0]
        language = lang_token.strip("<|>")
    elif force_language:
        language = force_language
    if not check_language(language):
        raise ValueError(f"unsupported language {language!r}")
    task_id = W.TRANSLATE if task == "translate" else W.TRANSCRIBE
    prompt = [W.SOT, _Tokenizer.language_token_id(language), task_id, W.NO_TIMESTAMPS]

    # STEP 3 — run the model, `concurrent_gpu_chunks` windows per generate call
    results = []
    for batch in chunkit(features, s.concurrent_gpu_chunks):
        feats = ctranslate2.StorageView.from_array(np.ascontiguousarray(batch))
        results.extend(whisper_model.generate(feats, [prompt] * len(batch), beam_size=beam_size, return_scores=False,
                                              fixed_new_tokens=fixed_new_tokens, input_kind=kind))
    assert len(results) == total_chunk_count, "Result length doesn't match expected total_chunk_count"
    if use_chunking:
        tokens = audio.find_longest_common_sequence([(results[i].sequences_ids[0], strides[i]) for i in range(total_chunk_count)],
                                                    tokenizer)
        tokens = [int(t) for t in tokens]
    else:
        tokens = results[0].sequences_ids[0]
    text = tokenizer.decode(tokens).strip()

    translation = None
    if translate and total_chunk_count <= s.concurrent_gpu_chunks:       # main.py:729-748 (its `len(int)` bug aside: short audio only)
        tprompt = [W.SOT, _Tokenizer.language_token_id(language), W.TRANSLATE, W.NO_TIMESTAMPS]
        feats = ctranslate2.StorageView.from_array(np.ascontiguousarray(features))
        tres = whisper_model.generate(feats, [tprompt] * total_chunk_count, beam_size=beam_size, fixed_new_tokens=fixed_new_tokens, input_kind=kind)
        translation = tokenizer.decode(tres[0].sequences_ids[0]).strip()
        out_translation_tokens = tres[0].sequences_ids[0]
    else:
        out_translation_tokens = None

    infer_time_milliseconds = (time.perf_counter() - first_time_start) * 1000
    infer_speedup = math.floor(audio_duration / infer_time_milliseconds)
    out = WhisperResult((language, text, infer_time_milliseconds, translation, infer_speedup, audio_duration))
    out.tokens = tokens
    out.translation_tokens = out_translation_tokens
    return out
