"""Host-side ceiling of the re-hosted /api/asr (no GPU needed): how many requests per second can ONE Python process take in, decode,
batch and answer when the engine costs the host nothing?

    python tools/host_ceiling.py [--gpus 8] [--replicas 4] [--clients 256] [--seconds 4]

The engine call (`wis_hip.ctranslate2._generate_chunk`, i.e. wis_generate behind ctypes with the GIL released) is replaced by a sleep
of 30 ms + 4 ms per utterance of the device batch - what a large-v2 beam-5 batch of 3.84 s utterances costs on an MI355X (62 ms for
8) - so everything else is real: the ASGI app, multipart parsing, the FLAC container decode (csrc/audio_io.c through ctypes),
pad_or_trim to a 30 s window, the micro-batcher (one worker thread per replica), stacking the PCM windows of a device batch, result
objects and JSON.  Clients run in the same process over httpx's ASGI transport (their cost is charged to the server: a lower
bound).  The question it answers (round-4 review item 5d): can one process feed 8 GPUs x 160 utterances/s = 1280 requests/s?
"""
import argparse
import asyncio
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "willow-inference-server_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--replicas", type=int, default=4)
    ap.add_argument("--clients", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import httpx
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels

    sizes = []

    def fake_chunk(r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind, device_ptr=None):
        B = int(mel) if device_ptr is not None else mel.shape[0]
        sizes.append(B)
        time.sleep(0.030 + 0.004 * B)
        return [ct2.WhisperGenerationResult([[400 + i for i in range(16)]], [-0.5]) for _ in range(B)]

    ct2._generate_chunk = fake_chunk
    _lib.device_count = lambda: a.gpus
    s = APISettings()
    s.whisper_model_path = "synthetic:{size}"
    s.max_batch = a.batch
    models = WhisperModels(s, device_index=list(range(a.gpus)))
    handles = [(None, d) for d in range(a.gpus)] + [(None, d) for d in range(a.gpus) for _ in range(a.replicas - 1)]
    model = ct2.Whisper.from_handles(handles, W.arch("large"), max_batch=a.batch, max_beam=5)
    models._models["large"] = model
    app = create_app(models=models, max_workers=max(64, a.clients))
    clip = open(os.path.join(ROOT, "tests", "golden", "clips", "3sec.flac"), "rb").read()
    b = "wisBenchBoundary"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() + clip + f"\r\n--{b}--\r\n".encode()
    hdr = {"content-type": f"multipart/form-data; boundary={b}"}
    url = "/api/asr?task=transcribe&output=json&model=large&beam_size=5&detect_language=False"
    lat = []

    async def client(c, stop):
        while time.perf_counter() < stop:
            t = time.perf_counter()
            r = await c.post(url, content=body, headers=hdr)
            assert r.status_code == 200, r.text
            lat.append(time.perf_counter() - t)

    async def go():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://wis", timeout=600) as c:
            await asyncio.gather(*[client(c, time.perf_counter() + 0.5) for _ in range(min(a.clients, 32))])      # warm-up
            lat.clear(); sizes.clear()
            t0 = time.perf_counter()
            await asyncio.gather(*[client(c, t0 + a.seconds) for _ in range(a.clients)])
            return time.perf_counter() - t0

    el = asyncio.run(go())
    n = len(lat)
    cap = a.gpus * a.replicas * a.batch / (0.030 + 0.004 * a.batch)
    print(f"fake engine: {a.gpus} GPUs x {a.replicas} replicas, device batches of <= {a.batch} at 30 + 4 B ms -> engine capacity {cap:.0f} utterances/s")
    print(f"{a.clients} in-process clients, {el:.2f} s: {n / el:.0f} requests/s answered, p50 {1e3 * float(np.median(lat)):.1f} ms, mean device batch {np.mean(sizes):.2f} "
          f"({len(sizes)} batches), host cores {os.cpu_count()}")
    model._replicas = []
    model.close()


if __name__ == "__main__":
    main()
