"""ASGI app factory with the ENGINE replaced by a sleep (tools/host_ceiling.py, tests/test_server_cpu.py): everything of the re-hosted server but
wis_generate is real - ASGI, multipart parsing, FLAC container decode, pad_or_trim, the micro-batcher, result objects, JSON.

    python -m wis_hip.server --workers-per-node 8 --app fake_engine_app:create_app        (PYTHONPATH: tools/ and willow-inference-server_amd/)

One "GPU" per process with WIS_FAKE_REPLICAS (4) replicas; a device batch of B utterances costs WIS_FAKE_MS (30) + WIS_FAKE_MS_PER_UTT (4) x B ms -
large-v2 beam 5 on 3.84 s utterances on an MI355X (62 ms for 8).  Every process appends its device-batch sizes, one per line, to
$WIS_FAKE_STATS_DIR/worker_<index>.txt."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "willow-inference-server_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def create_app():
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    from wis_hip.server import create_app as real_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels
    base, per = float(os.environ.get("WIS_FAKE_MS", "30")) * 1e-3, float(os.environ.get("WIS_FAKE_MS_PER_UTT", "4")) * 1e-3
    replicas, batch = int(os.environ.get("WIS_FAKE_REPLICAS", "4")), int(os.environ.get("WIS_FAKE_BATCH", "8"))
    sizes = []
    stats_dir = os.environ.get("WIS_FAKE_STATS_DIR")
    log = None
    if stats_dir:      # one line per device batch, written as it happens (a SIGTERM'd uvicorn re-raises the signal after draining: no atexit)
        os.makedirs(stats_dir, exist_ok=True)
        log = open(os.path.join(stats_dir, f"worker_{os.environ.get('WIS_WORKER_INDEX', '0')}.txt"), "a", buffering=1)

    def fake_chunk(r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind, device_ptr=None, **_kw):
        B = int(mel) if device_ptr is not None else mel.shape[0]
        sizes.append(B)
        if log is not None:
            log.write(f"{B}\n")
        time.sleep(base + per * B)
        return [ct2.WhisperGenerationResult([[400 + i for i in range(16)]], [-0.5]) for _ in range(B)]

    ct2._generate_chunk = fake_chunk
    _lib.device_count = lambda: 1
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch = batch
    models = WhisperModels(s, device_index=[0])
    model = ct2.Whisper.from_handles([(None, 0)] * replicas, W.arch("large"), max_batch=batch, max_beam=5)
    models._models["large"] = model
    return real_app(models=models, max_workers=64)
