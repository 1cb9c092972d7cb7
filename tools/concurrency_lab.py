"""Tuning aid: utterances per second with 1..R device batches in flight on one GPU (bench.py's concurrent_batches alone, so that
launch-side switches can be compared without the whole bench): python tools/concurrency_lab.py [B] [R] [iters]
Environment switches read by the library at load: WIS_GEMM_PERSIST, WIS_CA_SPIN, WIS_FRAG_KSPLIT, ..."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "willow-inference-server_amd"))
import bench  # noqa: E402
from wis_hip import _lib, audio, ctranslate2 as ct2, weights as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lib = _lib.load()
a = W.arch("large")
arena, index = W.build_arena(W.synthetic_weights("large"))
h = ct2.create_handle(a, arena, index, 0, max_batch=B, max_beam=5)
del arena
handles = [h]
for _ in range(R - 1):
    c = C.c_void_p()
    _lib.check(lib.wis_model_clone(h, C.byref(c)))
    handles.append(c)
pcm, _sr = audio.load_audio(os.path.join(ROOT, "tests", "golden", "clips", "3sec.flac"))
audio_ms = 1000.0 * pcm.shape[0] / 16000.0
out = bench.concurrent_batches(lib, handles, 0, pcm, 5, B, bench.FIXED_NEW["3sec"], audio_ms, iters=iters)
sw = {k: v for k, v in os.environ.items() if k.startswith("WIS_") and k not in ("WIS_TAG", "WIS_LIB_PATH")}
print(json.dumps({"switches": sw, "B": B, "utterances_per_s": [r["utterances_per_s"] for r in out["rows"]], "ms_per_device_batch": [r["ms_per_device_batch"] for r in out["rows"]]}))
