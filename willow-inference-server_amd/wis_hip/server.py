"""Thin re-host of the reference's REST ASR endpoints over the HIP path:

    GET  /api/ping      (main.py:1129-1137)
    POST /api/asr       multipart upload `audio_file`             (main.py:1168-1234)
    POST /api/willow    raw body + x-audio-* headers              (main.py:1237-1377)

Same query parameters, same JSON fields, same 400 behaviour (`{"error": "Invalid force_language"}`, `{"error": "Invalid
audio"}`).  Differences, all deliberate:
  * inference runs OFF the event loop (a thread pool calls `do_whisper`; ctypes releases the GIL), so concurrent requests
    reach the model together and the micro-batcher (wis_hip/batching.py) turns them into device batches - the reference blocks
    its single worker inside `do_whisper` (main.py:1205, 1338);
  * audio containers: WAV, FLAC and raw PCM are decoded natively (csrc/audio_io.c); other codecs need PyAV, which the
    reference uses (`audio_to_wav`, main.py:108-120) and this image does not have -> HTTP 400 "Invalid audio";
  * speaker verification (`voice_auth`) is a different model family and out of scope -> HTTP 400.
WebRTC (`/api/rtc/asr`), TTS, nginx auth and the static sites are not re-hosted (out of scope).

    python -m wis_hip.server --host 0.0.0.0 --port 19000        (or: uvicorn --factory wis_hip.server:create_app ...)
"""
import asyncio
import io
import logging
import os
import wave
from concurrent.futures import ThreadPoolExecutor
from contextlib import asynccontextmanager
from email.parser import BytesParser
from email.policy import HTTP

from fastapi import FastAPI, Request
from fastapi.middleware.cors import CORSMiddleware
from fastapi.responses import JSONResponse

from .settings import get_api_settings
from .whisper import InvalidAudio, WhisperModels, check_language, do_whisper

logger = logging.getLogger("infer")


def write_stream_wav(data, rate, bits, ch):
    """Raw PCM from a Willow device -> in-memory WAV (main.py:98-105)."""
    f = io.BytesIO()
    w = wave.open(f, "wb")
    w.setparams((ch, bits // 8, rate, 0, "NONE", "NONE"))
    w.writeframesraw(bytes(data))
    w.close()
    f.seek(0)
    return f


def _multipart_boundary(content_type):
    for piece in (content_type or "").split(";")[1:]:
        k, _, v = piece.strip().partition("=")
        if k.strip().lower() == "boundary":
            return v.strip().strip('"').encode("latin-1")
    return None


def parse_multipart(body, content_type, field="audio_file"):
    """The one multipart field /api/asr needs (python-multipart is not installed, so FastAPI's UploadFile is unavailable).
    Parts are located with bytes.find on the boundary and only the part HEADERS go through the e-mail header parser: the payload
    (the uploaded audio, tens of kilobytes to megabytes) is sliced, never scanned line by line - the generic MIME parser spent
    2.6 ms of GIL-held time on the 67 KB reference clip, the largest per-request cost of the host path (tools/host_ceiling.py)."""
    if "multipart/form-data" not in (content_type or ""):
        raise ValueError("expected multipart/form-data")
    boundary = _multipart_boundary(content_type)
    if not boundary:
        raise ValueError("multipart boundary missing")
    body = bytes(body) if not isinstance(body, bytes) else body
    delim = b"--" + boundary
    pos = body.find(delim)
    while pos >= 0:
        start = pos + len(delim)
        if body[start:start + 2] == b"--":           # closing delimiter
            break
        eol = body.find(b"\r\n", start)
        if eol < 0:
            break
        head_end = body.find(b"\r\n\r\n", eol)
        if head_end < 0:
            break
        nxt = body.find(b"\r\n" + delim, head_end + 4)
        if nxt < 0:
            raise ValueError("multipart body is not terminated")
        headers = BytesParser(policy=HTTP).parsebytes(body[eol + 2:head_end] + b"\r\n\r\n", headersonly=True)
        if headers.get_param("name", header="content-disposition") == field:
            payload = body[head_end + 4:nxt]
            cte = (headers.get("content-transfer-encoding") or "").strip().lower()
            if cte == "base64":
                import base64
                payload = base64.b64decode(payload)
            elif cte == "quoted-printable":
                import quopri
                payload = quopri.decodestring(payload)
            return payload
        pos = nxt + 2
    raise ValueError(f"multipart field {field!r} missing")


class BadRequest(ValueError):
    """A request parameter the engine cannot serve -> HTTP 400 with the message."""


def _as_bool(v, default):
    if v is None:
        return default
    return str(v).strip().lower() in ("1", "true", "yes", "on")


def create_app(models=None, settings=None, max_workers=None):
    s = settings or (models.settings if models is not None else get_api_settings())
    state = {"models": models}
    # enough threads that a full device batch per GPU replica can be waiting in the micro-batcher at once
    pool = ThreadPoolExecutor(max_workers=max_workers or 8 * max(1, s.max_batch), thread_name_prefix="wis-req")

    def get_models():
        if state["models"] is None:
            state["models"] = WhisperModels(s)
        return state["models"]

    @asynccontextmanager
    async def lifespan(_app):       # main.py:1097-1101: load, then warm on the reference clip when one is configured
        if os.environ.get("WIS_PRELOAD", "0") == "1":
            m = get_models()
            m.preload()
            clip = os.environ.get("WIS_WARM_CLIP")
            if clip and os.path.exists(clip):
                m.warm(clip)
        yield
        pool.shutdown(wait=False, cancel_futures=True)

    app = FastAPI(title=s.name, description=s.description, version=s.version, openapi_url="/api/openapi.json", docs_url="/api/docs",
                  redoc_url="/api/redoc", lifespan=lifespan)
    if s.cors_allowed_origins:
        app.add_middleware(CORSMiddleware, allow_origins=s.cors_allowed_origins, allow_credentials=True, allow_methods=["*"], allow_headers=["*"])

    def params(request):
        q = request.query_params
        model = q.get("model", s.whisper_model_default)
        beam = q.get("beam_size")
        try:
            beam = int(beam) if beam not in (None, "") else s.beam_size
        except ValueError:
            raise BadRequest(f"Invalid beam_size {beam!r}")
        # the engine's beam ceiling is a property of the request, not a server fault (the reference accepts any int, main.py:1180)
        if not 1 <= beam <= s.max_beam:
            raise BadRequest(f"beam_size {beam} outside 1..{s.max_beam}")
        return dict(model=model, beam_size=beam,
                    detect_language=_as_bool(q.get("detect_language"), s.detect_language), force_language=q.get("force_language") or None,
                    translate=_as_bool(q.get("translate"), False))

    async def run_whisper(audio_file, p):
        loop = asyncio.get_running_loop()
        return await loop.run_in_executor(pool, lambda: do_whisper(audio_file, p["model"], p["beam_size"], "transcribe", p["detect_language"],
                                                                   p["force_language"], p["translate"], models=get_models()))

    def bad(msg):
        return JSONResponse(content={"error": msg}, status_code=400)

    @app.get("/api/ping")
    async def ping():
        return JSONResponse(content={"message": "pong"})

    @app.post("/api/asr")
    async def asr(request: Request):
        try:
            p = params(request)
        except BadRequest as e:
            return bad(str(e))
        if p["force_language"] and not check_language(p["force_language"]):
            return bad("Invalid force_language")
        try:
            data = parse_multipart(await request.body(), request.headers.get("content-type"))
            res = await run_whisper(io.BytesIO(data), p)
        except InvalidAudio as e:
            logger.debug("ASR: %s - returning HTTP 400", e)
            return bad("Invalid audio")
        except ValueError as e:        # malformed upload / unknown model (the reference lets these surface as 500s)
            return bad(str(e))
        language, text, infer_time, translation, infer_speedup, audio_duration = res
        out = {"infer_time": infer_time, "infer_speedup": infer_speedup, "audio_duration": audio_duration, "language": language, "text": text}
        if translation:
            out["translation"] = translation
        return JSONResponse(content=out)

    @app.post("/api/willow")
    async def willow(request: Request):
        try:
            p = params(request)
        except BadRequest as e:
            return bad(str(e))
        q = request.query_params
        save_audio, stats, voice_auth = _as_bool(q.get("save_audio"), False), _as_bool(q.get("stats"), False), _as_bool(q.get("voice_auth"), False)
        if p["force_language"] and not check_language(p["force_language"]):
            return bad("Invalid force_language")
        if voice_auth:
            return bad("voice_auth (speaker verification) is not part of this build")
        h = request.headers
        sample_rate, bits, channel = h.get("x-audio-sample-rate", "").lower(), h.get("x-audio-bits", "").lower(), h.get("x-audio-channel", "").lower()
        codec = h.get("x-audio-codec", "").lower()
        chunks = []
        async for chunk in request.stream():
            chunks.append(chunk)
        body = b"".join(chunks)
        try:
            if codec == "pcm":
                audio_file = write_stream_wav(body, int(sample_rate), int(bits), int(channel))
            elif codec in ("wav", "flac"):
                audio_file = io.BytesIO(body)
            else:
                raise InvalidAudio(f"codec {codec!r} needs PyAV")
            if save_audio:
                path = os.environ.get("WIS_SAVE_AUDIO_PATH", "nginx/static/audio/willow.wav")
                os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                with open(path, "wb") as f:
                    f.write(audio_file.getbuffer())
            res = await run_whisper(audio_file, p)
        except InvalidAudio as e:
            logger.debug("WILLOW: %s - returning HTTP 400", e)
            return bad("Invalid audio")
        except ValueError as e:        # bad x-audio-* header values / unknown model
            return bad("Invalid audio" if "invalid literal" in str(e) else str(e))
        language, text, infer_time, translation, infer_speedup, audio_duration = res
        if stats:
            out = {"infer_time": infer_time, "infer_speedup": infer_speedup, "audio_duration": audio_duration, "language": language, "text": text}
        else:
            out = {"language": language, "text": text}
        if translation:
            out["translation"] = translation
        return JSONResponse(content=out)

    app.state.wis = state
    app.state.pool = pool
    return app


def _listen_socket(host, port):
    """A listening socket of its own on the SHARED (host, port): SO_REUSEPORT lets every worker process bind the same address and the kernel
    spreads incoming connections over the listeners (by a hash of the connection's addresses) - one port for the node, no proxy process in
    the data path."""
    import socket
    fam = socket.AF_INET6 if ":" in host else socket.AF_INET
    sock = socket.socket(fam, socket.SOCK_STREAM)
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)
    sock.bind((host, port))
    sock.listen(2048)
    return sock


def _load_factory(spec):
    """'package.module:callable' -> the callable (default: this module's create_app; tests / tools/host_ceiling.py serve a fake-engine app)"""
    import importlib
    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name or "create_app")


def serve_worker(args):
    """One server process of a node (spawned by `supervise`): sees ONE GPU (the parent set HIP_VISIBLE_DEVICES before this interpreter
    started, so device 0 here is the node's device `args.device`), listens on the shared port, drains on SIGTERM (uvicorn: stops accepting,
    finishes the requests in flight)."""
    import uvicorn
    app = _load_factory(args.app)()
    config = uvicorn.Config(app, log_level=args.log_level, timeout_graceful_shutdown=args.graceful_timeout, timeout_keep_alive=3600,
                            forwarded_allow_ips=os.environ.get("FORWARDED_ALLOW_IPS", "127.0.0.1"))
    logger.info("worker %d: GPU %s, pid %d, %s:%d", args.worker_index, os.environ.get("HIP_VISIBLE_DEVICES", "all"), os.getpid(), args.host, args.port)
    uvicorn.Server(config).run(sockets=[_listen_socket(args.host, args.port)])


def supervise(args, devices):
    """`--workers-per-node N`: N server processes behind ONE port - the deployment that serves a node (utterances shard over
    the GPUs with no exchange; reference main.py:295 `device_index=[*range(n)]` inside one gunicorn worker, entrypoint.sh:19-21).  One Python
    process answers 370-390 requests/s (tools/host_ceiling.py: ASGI + multipart + FLAC decode + batching under the GIL), an MI355X decodes
    ~170 utterances/s of the jmeter shape, so ONE process cannot feed eight GPUs: each GPU gets a process of its own (its model replicas, its
    micro-batcher, its thread pool), pinned with HIP_VISIBLE_DEVICES, all listening on the same address with SO_REUSEPORT.  This parent only
    supervises: it restarts a worker that dies (three times, then gives the node up), forwards SIGTERM / SIGINT and waits `--graceful-timeout`
    seconds for the workers to drain before it kills what is left (the reference's gunicorn flags: --graceful-timeout 10)."""
    import signal
    import subprocess
    import sys
    import time
    n = args.workers_per_node
    base = [sys.executable, "-m", "wis_hip.server", "--host", args.host, "--port", str(args.port), "--log-level", args.log_level, "--app", args.app,
            "--graceful-timeout", str(args.graceful_timeout)]
    procs, restarts, stopping = {}, {}, []

    def spawn(i):
        env = dict(os.environ)
        if devices:
            env["HIP_VISIBLE_DEVICES"] = str(devices[i % len(devices)])
        env["WIS_WORKER_INDEX"] = str(i)
        procs[i] = subprocess.Popen(base + ["--worker-index", str(i)], env=env)

    def stop(signum, _frame):
        stopping.append(signum)

    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    for i in range(n):
        spawn(i)
    logger.info("supervisor pid %d: %d worker processes on %s:%d (GPUs %s)", os.getpid(), n, args.host, args.port, devices or "unpinned")
    rc = 0
    while not stopping:
        time.sleep(0.2)
        for i, p in list(procs.items()):
            if p.poll() is not None and not stopping:
                restarts[i] = restarts.get(i, 0) + 1
                logger.warning("worker %d (pid %d) exited with %s", i, p.pid, p.returncode)
                if restarts[i] > 3:
                    logger.error("worker %d keeps dying: shutting the node down", i)
                    stopping.append(signal.SIGTERM)
                    rc = 1
                    break
                spawn(i)
    for p in procs.values():
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
    deadline = time.time() + args.graceful_timeout + 2
    for p in procs.values():
        try:
            p.wait(max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main(argv=None):
    """`python -m wis_hip.server [--host 0.0.0.0] [--port 19000] [--workers-per-node N]` - the reference serves on 19000 behind nginx
    (entrypoint.sh:19-21, nginx.conf:99-103).  N > 1: one server process per GPU behind the one port (supervise)."""
    import argparse
    import uvicorn
    ap = argparse.ArgumentParser(description="Willow Inference Server ASR endpoints over the MI355X HIP path")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=19000)
    ap.add_argument("--log-level", default=os.environ.get("LOG_LEVEL", "info"))
    ap.add_argument("--workers-per-node", type=int, default=1, help="server processes behind the one port: one per GPU (0 = one per visible GPU)")
    ap.add_argument("--devices", default="", help="comma-separated GPU ids the workers are pinned to, round-robin (default: 0 .. workers-1)")
    ap.add_argument("--graceful-timeout", type=int, default=10, help="seconds a worker may take to finish the requests in flight on SIGTERM")
    ap.add_argument("--app", default="wis_hip.server:create_app", help="module:factory of the ASGI app the workers serve")
    ap.add_argument("--worker-index", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO))
    if args.worker_index is not None:
        return serve_worker(args)
    n = args.workers_per_node
    if n == 0:
        from . import _lib
        n = args.workers_per_node = max(1, _lib.device_count())
    if n > 1:
        devices = [int(d) for d in args.devices.split(",") if d.strip()] or list(range(n))
        raise SystemExit(supervise(args, devices))
    uvicorn.run(_load_factory(args.app)(), host=args.host, port=args.port, log_level=args.log_level)


if __name__ == "__main__":
    main()
