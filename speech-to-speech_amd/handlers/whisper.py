"""``--stt mi355x-whisper``: drop-in for the reference's WhisperSTTHandler (STT/whisper_stt_handler.py:47-282) on the
MI355X session pool.  Same ``process()`` contract: progressive input -> at most one PartialTranscription, final input ->
exactly one Transcription carrying turn ids, ``speech_stopped_at_s = vad_audio.created_at_s`` and a language code
(``<code>-auto`` when the handler was started with language="auto")."""

from __future__ import annotations

import logging
from typing import Any, Iterator

import torch

from ..boundary import BaseSTTHandler, PartialTranscription, Transcription
from ..checkpoints import WHISPER_DIMS, load_safetensors_dir, parse_model_name
from ..session_pool import SessionPool, release_pool_on_failure
from ..text import WhisperVocab, load_tokenizer

logger = logging.getLogger(__name__)
SUPPORTED_LANGUAGES = ["en", "fr", "es", "zh", "ja", "ko", "hi", "de", "pt", "pl", "it", "nl"]
DEFAULT_LANGUAGE = "en"


def build_whisper_engine(model_name: str, device: str, torch_dtype: str, max_batch: int):
    from ..synth_weights import whisper_state_dict
    from ..whisper import WhisperEngine

    kind, key, seed = parse_model_name(model_name)
    dtype = getattr(torch, torch_dtype)
    if kind == "synthetic":
        dims = WHISPER_DIMS[key]()
        if dims.d_model >= 1024:      # big synthetic models are generated on the GPU directly
            return WhisperEngine.from_random(dims, dtype, max_batch=max_batch, device=device, seed=seed)
        return WhisperEngine(dims, whisper_state_dict(dims, seed=seed, gain=2.0, pos_std=2.0), dtype=dtype, max_batch=max_batch, device=device)
    import json
    import os

    with open(os.path.join(key, "config.json")) as f:
        cfg = json.load(f)
    from ..synth_weights import WhisperDims

    dims = WhisperDims(n_mels=cfg["num_mel_bins"], d_model=cfg["d_model"], n_heads=cfg["encoder_attention_heads"], enc_layers=cfg["encoder_layers"],
                       dec_layers=cfg["decoder_layers"], ffn=cfg["encoder_ffn_dim"], vocab=cfg["vocab_size"], max_target=cfg["max_target_positions"])
    return WhisperEngine(dims, load_safetensors_dir(key), dtype=dtype, max_batch=max_batch, device=device)


class MI355XWhisperSTTHandler(BaseSTTHandler):
    @release_pool_on_failure
    def setup(self, model_name: str = "synthetic:test-tiny", device: str = "cuda:0", torch_dtype: str = "bfloat16", language: str | None = "en",
              max_batch: int = 64, batch_window_ms: float = 4.0, gen_kwargs: dict | None = None, vocab: WhisperVocab | None = None,
              pool: SessionPool | None = None, batch_cap: int = 0, **_ignored: Any) -> None:
        self.gen_kwargs = dict(gen_kwargs or {})
        self.start_language = language
        self.last_language = language if language != "auto" else None
        kind, key, _ = parse_model_name(model_name) if pool is None or "whisper" not in getattr(pool, "_engines", {}) else ("pool", None, None)
        self.tokenizer = load_tokenizer(key if kind == "dir" else None)
        self._owns_pool = pool is None
        self.pool = pool or SessionPool.for_device(device, max_batch=max_batch, batch_window_ms=batch_window_ms)
        self.pool.attach(self)                 # every user is counted, also one that was handed its pool (session_pool.attach)
        if int(batch_cap) > 0:
            self.pool.configure(stt_max_batch=int(batch_cap))
        self.pool.ensure_engine("whisper", lambda: build_whisper_engine(model_name, device, torch_dtype, max_batch))
        # special-token ids, suppress_tokens and begin_suppress_tokens come from the checkpoint's generation_config.json - what
        # the reference's model.generate() applies (generation_whisper.py:1774-1812); synthetic checkpoints get a vocabulary
        # laid out for their size.  Every id is checked against the model's vocabulary before anything is launched.
        dims = getattr(self.pool.engine("whisper"), "dims", None)
        if vocab is None:
            vocab = WhisperVocab.from_checkpoint(key) if kind == "dir" else WhisperVocab.synthetic(dims.vocab if dims is not None else 51866)
        self.vocab = vocab.validate(dims.vocab) if dims is not None else vocab
        self.task = self.gen_kwargs.get("task", "transcribe")
        if language not in (None, "auto") and self.vocab.multilingual and language not in self.vocab.lang_to_id:
            raise ValueError(f"language {language!r} is not supported by this checkpoint ({sorted(self.vocab.lang_to_id)})")
        (self.pool.warm_once("whisper", self.warmup) if hasattr(self.pool, "warm_once") else self.warmup())      # one warm-up per GPU engine, not per pipeline unit

    def warmup(self) -> None:
        import numpy as np

        logger.info("Warming up %s", self.__class__.__name__)
        lang = self.last_language if self.last_language in self.vocab.lang_to_id else (DEFAULT_LANGUAGE if DEFAULT_LANGUAGE in self.vocab.lang_to_id else "auto")
        self.pool.transcribe("whisper", np.zeros(16000, np.float32), language=lang, max_new_tokens=2, vocab=self.vocab, task=self.task)

    def process(self, vad_audio: Any) -> Iterator[Any]:
        forced = self.gen_kwargs.get("language") or (self.start_language if self.start_language not in (None, "auto") else None)
        opts = dict(language=forced or "auto", max_new_tokens=int(self.gen_kwargs.get("max_new_tokens", 128)), vocab=self.vocab, task=getattr(self, "task", "transcribe"))
        if vad_audio.mode == "progressive":
            # the reference transcribes EVERY progressive item (STT/whisper_stt_handler.py:166-282 has no skip path); the pool serves live
            # partials behind the queued finals (live=True) but never drops them (rounds 2 - 5 shed them by pool backlog)
            try:
                res = self.pool.transcribe("whisper", vad_audio.audio, live=True, **opts)
            except Exception as e:  # noqa: BLE001
                logger.debug("Whisper progressive transcription failed: %s", e)
                return
        else:
            try:
                res = self.pool.transcribe("whisper", vad_audio.audio, **opts)
            except Exception as e:  # noqa: BLE001  (reference policy: inference failure degrades to an empty transcript)
                logger.error("Whisper inference failed: %s", e)
                res = {"ids": [], "language": None}
        language_code = forced or res.get("language") or self.last_language or DEFAULT_LANGUAGE
        if language_code in SUPPORTED_LANGUAGES:
            self.last_language = language_code
        text = self.tokenizer.decode(res["ids"]).strip()
        if self.start_language == "auto":
            language_code += "-auto"
        if vad_audio.mode == "progressive":
            yield PartialTranscription(text=text, turn_id=vad_audio.turn_id, turn_revision=vad_audio.turn_revision)
            return
        yield Transcription(text=text, language_code=language_code, turn_id=vad_audio.turn_id, turn_revision=vad_audio.turn_revision,
                            speech_stopped_at_s=vad_audio.created_at_s)

    def on_session_end(self) -> None:
        """Per-session state back to its start values: the language carried from turn to turn (`last_language`, the fallback of an
        auto-detect session) must not leak into the next session of this pipeline unit; BaseSTTHandler clears its completed-final set."""
        super().on_session_end()
        start = getattr(self, "start_language", None)
        self.last_language = start if start != "auto" else None
        logger.debug("MI355X Whisper session state reset")

    def cleanup(self) -> None:
        """The model belongs to the device's SessionPool (weights once per GPU): the handler lets go of it, the pool closes its engines
        when the last handler that obtained it through `for_device` is gone."""
        logger.info("Cleaning up %s", self.__class__.__name__)
        pool, self.pool = getattr(self, "pool", None), None
        if pool is not None:
            pool.detach(self)
