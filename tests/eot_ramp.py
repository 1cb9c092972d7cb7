"""Test helper (not product code): seeded synthetic weights whose decoder emits EOT by itself.

Seeded random Whisper weights never rank EOT high, so every earlier generate-vs-oracle test ended on the
max-length step or under the fixed-length measurement convention.  The reference path always ends on EOT
(main.py:687-693 passes no max_length).  `with_eot_ramp` adds, to the learned decoder position table, a
multiple of the EOT token's (tied) embedding that grows with the text position: the residual stream carries
it to the final LayerNorm, so the EOT logit rises by roughly `slope` per step from position `start` on and
the search meets EOT candidates in mid-flight - first beyond the top k, then inside it.  Engine and oracle
read the same f16-rounded table."""
import numpy as np

EOT = 50257


def with_eot_ramp(w, start=6, slope=0.02, eot=EOT, ctx=448):
    """-> copy of the weight dict `w` with decoder positions p >= start shifted by slope * (p - start + 1) * e_eot / |e_eot|."""
    out = dict(w)
    pos = np.asarray(w["decoder/position_encodings/encodings"], np.float32).copy()
    e = np.asarray(w["decoder/embeddings/weight"], np.float32)[eot]
    e = e / np.linalg.norm(e)
    ramp = np.clip(np.arange(pos.shape[0], dtype=np.float32) - (start - 1), 0, None) * np.float32(slope)
    pos += ramp[:, None] * e[None, :]
    out["decoder/position_encodings/encodings"] = pos.astype(np.float16)
    return out
