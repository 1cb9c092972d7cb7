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


def main(argv=None):
    """`python -m wis_hip.server [--host 0.0.0.0] [--port 19000]` - the reference serves on 19000 behind nginx
    (entrypoint.sh:19-21, nginx.conf:99-103)."""
    import argparse
    import uvicorn
    ap = argparse.ArgumentParser(description="Willow Inference Server ASR endpoints over the MI355X HIP path")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=19000)
    ap.add_argument("--log-level", default=os.environ.get("LOG_LEVEL", "info"))
    args = ap.parse_args(argv)
    uvicorn.run(create_app(), host=args.host, port=args.port, log_level=args.log_level)


if __name__ == "__main__":
    main()
