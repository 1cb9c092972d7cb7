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


def _request_bytes(host, port):
    clip = open(os.path.join(ROOT, "tests", "golden", "clips", "3sec.flac"), "rb").read()
    b = "wisBenchBoundary"
    body = (f"--{b}\r\nContent-Disposition: form-data; name=\"audio_file\"; filename=\"3sec.flac\"\r\nContent-Type: audio/flac\r\n\r\n").encode() + clip + f"\r\n--{b}--\r\n".encode()
    url = "/api/asr?task=transcribe&output=json&model=large&beam_size=5&detect_language=False"
    head = (f"POST {url} HTTP/1.1\r\nHost: {host}:{port}\r\nContent-Type: multipart/form-data; boundary={b}\r\nContent-Length: {len(body)}\r\nConnection: keep-alive\r\n\r\n").encode()
    return head + body


def client_worker(host, port, conns, seconds):
    """One load-generator process: `conns` keep-alive connections (the jmeter threads of client/jmeter-asr.jmx:53-90), each looping POST /api/asr
    over a raw socket (pre-built request bytes, minimal response parsing: the generator must not be what is measured)."""
    import json
    import re
    req = _request_bytes(host, port)
    lat, errors = [], [0]

    async def one(stop):
        r, w = await asyncio.open_connection(host, port)
        try:
            while time.perf_counter() < stop:
                t = time.perf_counter()
                w.write(req)
                await w.drain()
                head = await r.readuntil(b"\r\n\r\n")
                n = int(re.search(rb"content-length:\s*(\d+)", head.lower()).group(1))
                await r.readexactly(n)
                if not head.startswith(b"HTTP/1.1 200"):
                    errors[0] += 1
                lat.append(time.perf_counter() - t)
        finally:
            w.close()

    async def go():
        stop = time.perf_counter() + seconds
        await asyncio.gather(*[one(stop) for _ in range(conns)])

    t0 = time.perf_counter()
    asyncio.run(go())
    print(json.dumps({"n": len(lat), "elapsed": time.perf_counter() - t0, "p50_ms": 1e3 * float(np.median(lat)) if lat else None, "errors": errors[0]}), flush=True)


def multi_process(a):
    """`--processes N`: the deployment of a node - `python -m wis_hip.server --workers-per-node N` (one server process per GPU behind ONE port,
    SO_REUSEPORT) with the fake engine - under the jmeter shape from `--client-procs` separate load-generator processes."""
    import glob
    import json
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as s0:
        s0.bind(("127.0.0.1", 0))
        port = s0.getsockname()[1]
    stats = tempfile.mkdtemp(prefix="wis_fake_stats_")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tools"), os.path.join(ROOT, "willow-inference-server_amd"), os.environ.get("PYTHONPATH", "")]),
               WIS_FAKE_STATS_DIR=stats, WIS_FAKE_REPLICAS=str(a.replicas), WIS_FAKE_BATCH=str(a.batch))
    env.pop("HIP_VISIBLE_DEVICES", None)
    sup = subprocess.Popen([sys.executable, "-m", "wis_hip.server", "--host", "127.0.0.1", "--port", str(port), "--workers-per-node", str(a.processes),
                            "--app", "fake_engine_app:create_app", "--log-level", "warning", "--graceful-timeout", "5"], env=env)
    try:
        deadline = time.time() + 60
        ready = 0
        while time.time() < deadline and ready < 3 * a.processes:      # every listener answers (connections are hashed over them: ask often)
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=1) as c:
                    c.sendall(b"GET /api/ping HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n")
                    if b"200" in c.recv(64):
                        ready += 1
            except OSError:
                time.sleep(0.2)
        time.sleep(1.0)
        per = max(1, a.clients // a.client_procs)
        cmd = [sys.executable, os.path.abspath(__file__), "--client-worker", "--port", str(port), "--clients", str(per)]
        warm = [subprocess.Popen(cmd + ["--seconds", "1"], stdout=subprocess.PIPE, env=env) for _ in range(a.client_procs)]
        for w in warm:
            w.communicate()
        t0 = time.perf_counter()
        gens = [subprocess.Popen(cmd + ["--seconds", str(a.seconds)], stdout=subprocess.PIPE, env=env) for _ in range(a.client_procs)]
        res = [json.loads(g.communicate()[0].decode().strip().splitlines()[-1]) for g in gens]
        el = time.perf_counter() - t0
    finally:
        sup.send_signal(signal.SIGTERM)
        try:
            sup.wait(20)
        except subprocess.TimeoutExpired:
            sup.kill()
    n = sum(r["n"] for r in res)
    sizes, per_worker = [], []
    for f in sorted(glob.glob(os.path.join(stats, "worker_*.txt"))):
        d = [int(x) for x in open(f).read().split()]
        sizes += d
        per_worker.append(len(d))
    span = max(r["elapsed"] for r in res)
    cap = a.processes * a.replicas * a.batch / (0.030 + 0.004 * a.batch)
    print(f"fake engine: {a.processes} server processes (one GPU each) x {a.replicas} replicas behind ONE port (SO_REUSEPORT), device batches of <= {a.batch} at 30 + 4 B ms -> engine capacity {cap:.0f} utterances/s")
    print(f"{per * a.client_procs} keep-alive connections from {a.client_procs} load-generator processes, {span:.2f} s: {n / span:.0f} requests/s answered, p50 "
          f"{np.median([r['p50_ms'] for r in res if r['p50_ms']]):.1f} ms, {sum(r['errors'] for r in res)} errors, mean device batch {np.mean(sizes) if sizes else float('nan'):.2f} "
          f"({len(sizes)} batches incl. warm-up; per worker {per_worker}), host cores {os.cpu_count()} shared by servers and generators (wall {el:.1f} s)")
    print("RESULT " + json.dumps({"processes": a.processes, "replicas": a.replicas, "connections": per * a.client_procs, "requests_per_s": round(n / span, 1), "errors": sum(r["errors"] for r in res),
                                  "mean_device_batch": round(float(np.mean(sizes)), 2) if sizes else None, "batches_per_worker": per_worker, "supervisor_exit": sup.returncode, "host_cores": os.cpu_count()}))
    return n / span, (float(np.mean(sizes)) if sizes else 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--replicas", type=int, default=4)
    ap.add_argument("--clients", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--processes", type=int, default=0, help="N > 0: `python -m wis_hip.server --workers-per-node N` (one process per GPU behind one port) instead of one in-process app")
    ap.add_argument("--client-procs", type=int, default=2, help="load-generator processes (--processes mode)")
    ap.add_argument("--client-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--port", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.client_worker:
        return client_worker("127.0.0.1", a.port, a.clients, a.seconds)
    if a.processes > 0:
        return multi_process(a)
    import httpx
    from wis_hip import _lib, ctranslate2 as ct2, weights as W
    from wis_hip.server import create_app
    from wis_hip.settings import APISettings
    from wis_hip.whisper import WhisperModels

    sizes = []

    def fake_chunk(r, mel, prompts, P, beam, max_new, lp, patience, suppress_blank, suppress_default, fixed_new, kind, device_ptr=None, **_kw):
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
