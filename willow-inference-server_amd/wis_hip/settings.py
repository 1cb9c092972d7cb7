"""Same field names and defaults as the reference's `settings.py:6-82` (pydantic-settings `APISettings`, env-overridable,
case-insensitive, no prefix).  `pydantic_settings` is not installable offline, so this is a plain dataclass that reads the
environment the same way; an optional `custom_settings.py` on sys.path shadows it, as in reference main.py:68-77.
Extra fields (not in the reference) are marked."""
import os
from dataclasses import dataclass, field, fields
from functools import lru_cache
from typing import List


@dataclass
class APISettings:
    name: str = "Willow Inference Server"
    description: str = "High Performance Language Inference API"
    version: str = "1.0"
    beam_size: int = 1
    long_beam_size: int = 3
    long_beam_size_threshold: int = 12000
    ctranslate2_threads: int = 10
    language: str = "en"
    detect_language: bool = False
    preload_all_models: bool = False
    preload_whisper_model_tiny: bool = True
    preload_whisper_model_base: bool = True
    preload_whisper_model_small: bool = True
    preload_whisper_model_medium: bool = True
    preload_whisper_model_large: bool = True
    sv_memory_threshold: int = 5798205849
    support_chunking: bool = True
    chunking_memory_threshold: int = 3798205849
    concurrent_gpu_chunks: int = 2
    support_sv: bool = False
    sv_threshold: float = 0.75
    whisper_model_default: str = "medium"
    cors_allowed_origins: List[str] = field(default_factory=list)
    aiortc_debug: bool = False
    # --- not in the reference: where the weights come from.  "{size}" is substituted.  The default is the reference's own
    # model directory layout (models/tovera-wis-whisper-{size}, utils.sh:99-108, main.py:341-444): a CTranslate2 or Hugging
    # Face checkpoint directory with its tokenizer files; a missing directory is an ERROR at first use.  Seeded synthetic
    # weights ("synthetic:{size}": benchmarks and tests, no checkpoint exists offline) must be asked for explicitly.
    whisper_model_path: str = "models/tovera-wis-whisper-{size}"
    # text output needs the checkpoint's tokenizer (tokenizer.json); without one a real model refuses to load unless this is
    # set, in which case `text` is the space-joined token ids (synthetic weights always behave that way: ids are the result)
    allow_token_id_text: bool = False
    # log-mel on the replica's GPU inside generate (PCM in, mel never leaves HBM) instead of the reference's
    # mel -> host -> StorageView round trip; results are identical (same kernels)
    fuse_logmel: bool = True
    max_batch: int = 8
    # replicas per GPU sharing one weight copy, each with its own stream / activations / KV caches: `ctranslate2_threads` (the
    # reference's inter_threads) device batches run concurrently per GPU, up to this many
    replicas_per_gpu: int = 4
    # largest beam_size a request may ask for = the engine's ceiling, 8 (csrc/kernels.hpp MAX_R): the reference hands any `beam_size`
    # query parameter to CTranslate2 as it is (main.py:1180-1209, 685-693), so clients that send 6..8 keep working; only a beam the
    # engine cannot decode is answered with HTTP 400 (not a 500).  The setting sizes the self-attention KV caches of EVERY replica:
    # max_batch * max_beam slots x 448 positions x d x 2 (K, V) x layers x 2 bytes = 73 MB per slot for large-v2, i.e. 4.7 GB per
    # replica at 8 x 8 (2.9 GB at 8 x 5) - 18.8 GB per large-v2 model and GPU with four replicas, of 288 GB.  Lower it to save memory
    # (the effective ceiling is logged when a model is loaded).
    max_beam: int = 8
    # streaming sessions of up to 30 s (BASELINE configs[4]): while the audio arrives, decode what has been heard every this many seconds
    # of new audio (at the beam the final call would use at that length - `beam_size`, `long_beam_size` from 12 s on - and only when the
    # language is known without detection); stop() then verifies the last such hypothesis against the FINAL window in multi-row passes
    # instead of decoding token by token: beam 1 the token chain (wis_generate_draft), a beam search its trajectory (wis_generate_draft_beam).
    # 0 = off.  The answer is the decode of the final window either way.
    stream_speculate_s: float = 2.0
    # speculation is OPTIONAL work: an interim decode is only started while the session's GPU has a replica to spare - fewer device batches
    # running than this fraction of its replicas and nothing queued (advisor, round 5: every session re-decoding every 2 s beside real
    # requests pushed those back in the FIFO batcher)
    stream_speculate_max_busy: float = 0.5
    # ... and whether sessions whose final call is a BEAM SEARCH speculate at all.  Off by default: a draft trajectory only helps while the final
    # search re-traces it, and how long it does is a property of the checkpoint - on the seeded weights of this build the low-ranked beams are
    # chaotic (1.2 s more audio re-orders them within 3-14 steps, tools/traj_lab.py; profiles/r06_traj_lab.txt), so 14 interim searches per
    # session (1.8 s of GPU time) buy nothing; tiny / base follow a draft to its end.  Measure on the real checkpoint, then switch on.
    stream_speculate_beam_search: bool = False
    # measurement convention for seeded synthetic weights, which never emit EOT: decode exactly this many tokens
    # (EOT masked until then, then forced).  0 = off: the product default, natural termination
    fixed_new_tokens: int = 0
    # "float16" or "int8_float16" (the reference picks int8_float16 on GPUs, main.py:242: here it quantises the decoder weights)
    compute_type: str = "float16"

    def __post_init__(self):
        env = {k.lower(): v for k, v in os.environ.items()}
        for f in fields(self):
            if f.name in env:
                raw = env[f.name]
                cur = getattr(self, f.name)
                if isinstance(cur, bool):
                    setattr(self, f.name, raw.strip().lower() in ("1", "true", "yes", "on"))
                elif isinstance(cur, int):
                    setattr(self, f.name, int(raw))
                elif isinstance(cur, float):
                    setattr(self, f.name, float(raw))
                elif isinstance(cur, list):
                    setattr(self, f.name, [s for s in raw.strip("[]").replace('"', "").split(",") if s])
                else:
                    setattr(self, f.name, raw)


@lru_cache()
def get_api_settings() -> APISettings:
    try:
        from custom_settings import get_api_settings as custom   # reference main.py:68-77
        return custom()
    except ImportError:
        return APISettings()
