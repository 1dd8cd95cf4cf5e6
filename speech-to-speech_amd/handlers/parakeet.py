"""``--stt mi355x-parakeet``: drop-in for the reference's ParakeetTDTSTTHandler (STT/parakeet_tdt_handler.py:236-370,
621-634) on the MI355X session pool; live partials follow the re-transcribe-the-window pattern of
STT/smart_progressive_streaming.py:69-157 (window capped at 15 s)."""

from __future__ import annotations

import logging
from typing import Any, Iterator

import torch

from ..boundary import BaseSTTHandler, PartialTranscription, Transcription
from ..checkpoints import PARAKEET_DIMS, load_safetensors_dir, parse_model_name
from ..session_pool import SessionPool
from ..text import load_tokenizer

logger = logging.getLogger(__name__)
PROGRESSIVE_WINDOW_S = 15.0


def build_parakeet_engine(model_name: str, device: str, compute_type: str, max_batch: int, max_seconds: float):
    from ..parakeet import ParakeetEngine
    from ..synth_weights import parakeet_state_dict

    kind, key, seed = parse_model_name(model_name)
    dtype = getattr(torch, compute_type)
    if kind == "synthetic":
        dims = PARAKEET_DIMS[key]()
        return ParakeetEngine(dims, parakeet_state_dict(dims, seed=seed or 12, gain=1.5), dtype=dtype, max_batch=max_batch, max_seconds=max_seconds, device=device)
    from ..synth_weights import ParakeetDims

    return ParakeetEngine(ParakeetDims.tdt_0_6b(), load_safetensors_dir(key), dtype=dtype, max_batch=max_batch, max_seconds=max_seconds, device=device)


class MI355XParakeetSTTHandler(BaseSTTHandler):
    def setup(self, model_name: str = "synthetic:test-tiny", device: str = "cuda:0", compute_type: str = "bfloat16", language: str | None = "en",
              max_batch: int = 64, batch_window_ms: float = 4.0, max_seconds: float = 30.0, enable_live_transcription: bool = False,
              live_transcription_update_interval: float = 0.5, gen_kwargs: dict | None = None, pool: SessionPool | None = None, **_ignored: Any) -> None:
        self.language = language or "en"
        self.enable_live_transcription = enable_live_transcription
        kind, key, _ = parse_model_name(model_name) if pool is None or "parakeet" not in getattr(pool, "_engines", {}) else ("pool", None, None)
        self.tokenizer = load_tokenizer(key if kind == "dir" else None)
        self.pool = pool or SessionPool.for_device(device, max_batch=max_batch, batch_window_ms=batch_window_ms)
        self.pool.ensure_engine("parakeet", lambda: build_parakeet_engine(model_name, device, compute_type, max_batch, max_seconds))

    def process(self, vad_audio: Any) -> Iterator[Any]:
        audio = vad_audio.audio
        if vad_audio.mode == "progressive":
            window = audio[-int(PROGRESSIVE_WINDOW_S * 16000):]
            res = self.pool.try_transcribe("parakeet", window, max_queue=self.pool.max_batch)
            if res is None:
                return     # busy: skip this partial, exactly like the reference's 10 ms try-lock
            yield PartialTranscription(text=self.tokenizer.decode(res["ids"]).strip(), turn_id=vad_audio.turn_id, turn_revision=vad_audio.turn_revision)
            return
        try:
            res = self.pool.transcribe("parakeet", audio)
            text = self.tokenizer.decode(res["ids"]).strip()
        except Exception as e:  # noqa: BLE001
            logger.error("Parakeet inference failed: %s", e)
            text = ""
        yield Transcription(text=text, language_code=self.language, turn_id=vad_audio.turn_id, turn_revision=vad_audio.turn_revision,
                            speech_stopped_at_s=vad_audio.created_at_s)
