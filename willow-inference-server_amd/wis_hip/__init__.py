"""wis_hip — MI355X-native Whisper ASR hot path for Willow Inference Server.

Host-side mirror of the reference's interface for this path:
  wis_hip.audio        <->  wis/audio.py            (log_mel_spectrogram, pad_or_trim, chunk_iter, find_longest_common_sequence)
  wis_hip.ctranslate2  <->  ctranslate2             (models.Whisper, StorageView, get_supported_compute_types)
  wis_hip.whisper      <->  main.py:do_whisper      (orchestrator + per-request model/beam selection)
  wis_hip.settings     <->  settings.py             (same field names / defaults)
All arithmetic runs in libwis_hip.so (hand-written gfx950 HIP kernels) through ctypes.
"""
__all__ = ["audio", "ctranslate2", "weights", "languages"]
