"""OpenAI-compatible server mode on the MI355X session pool: several ``speech-to-speech serve`` processes
or hosts share one GPU pool through the reference's own client handler (``--stt openai``, STT/openai_compatible_handler.py:44-160,
docs/openai-compatible-stt.md) with zero in-process changes.

    POST /v1/audio/transcriptions   multipart form: file = mono PCM16 WAV @16 kHz (what the reference client posts), model,
                                    language, response_format in {json, text}
                                    -> {"text": ..., "language": ...}   |   text/plain
    POST /v1/audio/speech           JSON {"input": text, "response_format": "pcm" | "wav"} -> 16 kHz int16 mono audio, streamed
                                    chunk by chunk as the pool produces it
    GET  /health                    pool statistics

Requests from all client processes are batched by the same SessionPool the in-process handlers use.  The multipart body is
parsed with the standard library (python-multipart is not a dependency).  `python -m speech_to_speech_amd.server` serves with
uvicorn; `create_app` is what the tests drive."""

import io
import wave
from email.parser import BytesParser
from email.policy import HTTP
from typing import Any

import numpy as np


def parse_multipart(body: bytes, content_type: str) -> dict[str, Any]:
    """-> {field name: str | bytes (file parts)} of a multipart/form-data body."""
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
    out: dict[str, Any] = {}
    if not msg.is_multipart():
        raise ValueError("expected multipart/form-data")
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        if name is None:
            continue
        payload = part.get_payload(decode=True)
        out[name] = payload if part.get_filename() is not None else (payload or b"").decode("utf-8", "replace")
    return out


def wav_to_float32(data: bytes, target_sr: int = 16000) -> np.ndarray:
    """PCM16 WAV bytes -> mono float32 in [-1, 1] at 16 kHz (other rates are resampled with scipy's polyphase filter, the
    reference's own dependency for that)."""
    with wave.open(io.BytesIO(data), "rb") as w:
        sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError(f"expected 16-bit PCM, got {8 * sw}-bit samples")
    x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    if sr != target_sr:
        from math import gcd

        from scipy.signal import resample_poly

        g = gcd(sr, target_sr)
        x = resample_poly(x, target_sr // g, sr // g).astype(np.float32)
    return x


def float_pcm16_wav(pcm16: np.ndarray, sr: int = 16000) -> bytes:
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.asarray(pcm16, dtype="<i2").tobytes())
    return buf.getvalue()


def create_app(pool: Any, stt_model: str = "whisper", tokenizer: Any = None, stt_options: dict | None = None, tts_prompt: Any = None,
               tts_max_new_tokens: int = 1536, default_language: str | None = "en"):
    """pool: a SessionPool holding the STT engine ``stt_model`` (and optionally "tts"); tokenizer: ``decode(ids) -> str``;
    tts_prompt: ``text -> TTSPrompt`` (the handler's prompt builder)."""
    from fastapi import FastAPI, HTTPException, Request
    from fastapi.responses import JSONResponse, PlainTextResponse, Response, StreamingResponse
    from starlette.concurrency import run_in_threadpool

    from .text import IdTextCodec

    tok = tokenizer or IdTextCodec()
    app = FastAPI(title="MI355X speech server")

    @app.get("/health")
    def health():
        return {"status": "ok", "pool": dict(pool.stats)}

    @app.post("/v1/audio/transcriptions")
    async def transcriptions(request: Request):
        try:
            form = parse_multipart(await request.body(), request.headers.get("content-type", ""))
            audio = wav_to_float32(form["file"])
        except KeyError:
            raise HTTPException(status_code=400, detail="missing 'file' part")
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=400, detail=f"bad request: {e}")
        language = (form.get("language") or form.get("languages[]") or default_language or "auto")
        opts = dict(stt_options or {})
        if stt_model == "whisper":
            # a language the vocabulary does not know would raise inside the BATCH and fail every co-batched session: refuse it here
            vocab = opts.get("vocab")
            known = getattr(vocab, "lang_to_id", None)
            if language != "auto" and known and language not in known:
                raise HTTPException(status_code=400, detail=f"unsupported language {language!r}")
            opts.setdefault("language", language)
        try:
            res = await run_in_threadpool(lambda: pool.transcribe(stt_model, audio, **opts))
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=f"transcription failed: {type(e).__name__}")
        text = tok.decode(res["ids"]).strip()
        lang = res.get("language") or (language if language != "auto" else None)
        if form.get("response_format", "json") == "text":
            return PlainTextResponse(text)
        return JSONResponse({"text": text, "language": lang})

    @app.post("/v1/audio/speech")
    async def speech(request: Request):
        if tts_prompt is None or "tts" not in getattr(pool, "_engines", {}):
            raise HTTPException(status_code=404, detail="this server has no TTS engine")
        body = await request.json()
        text = (body.get("input") or "").strip()
        if not text:
            raise HTTPException(status_code=400, detail="missing 'input'")
        fmt = body.get("response_format", "pcm")
        if fmt not in ("pcm", "wav"):
            raise HTTPException(status_code=400, detail="response_format must be 'pcm' or 'wav'")
        # the frame budget of one request must fit the engine, or the whole lane batch it joins would fail ("frame budget exceeded")
        cap = int(getattr(pool.engine("tts"), "max_frames", tts_max_new_tokens))
        try:
            budget = int(body.get("max_new_tokens", tts_max_new_tokens))
        except (TypeError, ValueError):
            raise HTTPException(status_code=400, detail="max_new_tokens must be an integer")
        budget = max(1, min(budget, cap, int(tts_max_new_tokens)))
        try:
            prompt = tts_prompt(text)
        except ValueError as e:
            raise HTTPException(status_code=400, detail=f"input too long: {e}")
        chunks, cancel = pool.synthesize(prompt, budget)
        if fmt == "wav":
            try:
                pcm = await run_in_threadpool(lambda: np.concatenate([np.asarray(c, dtype=np.int16) for c in chunks] or [np.zeros(0, np.int16)]))
            except BaseException:
                cancel.set()
                raise
            return Response(float_pcm16_wav(pcm), media_type="audio/wav")

        def gen():
            started = getattr(chunks, "started", None)      # (a pool that returns a plain iterator has no pacing signal)
            try:
                for c in chunks:
                    if started is not None:     # no silence trimming on this path: the first chunk is the first audio
                        started.set()
                        started = None
                    yield np.asarray(c, dtype="<i2").tobytes()
            finally:
                cancel.set()         # client gone (generator closed) or stream complete: the lane stops generating for this request

        return StreamingResponse(gen(), media_type="audio/pcm")

    return app


def main(argv=None) -> None:  # pragma: no cover - thin CLI
    import argparse

    import uvicorn

    from .handlers.whisper import build_whisper_engine
    from .session_pool import SessionPool
    from .text import WhisperVocab

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--stt-model", default="synthetic:test-tiny")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-batch", type=int, default=64)
    a = ap.parse_args(argv)
    pool = SessionPool.for_device(a.device, max_batch=a.max_batch)
    user = object()
    pool.attach(user)          # the server is a user of the device's pool like any handler: handlers coming and going do not take its engines away
    try:
        pool.ensure_engine("whisper", lambda: build_whisper_engine(a.stt_model, a.device, "bfloat16", a.max_batch))
        vocab = WhisperVocab.synthetic(pool.engine("whisper").dims.vocab)
        uvicorn.run(create_app(pool, "whisper", stt_options={"vocab": vocab, "max_new_tokens": 128}), host=a.host, port=a.port)
    finally:
        pool.detach(user)


if __name__ == "__main__":  # pragma: no cover
    main()
