"""Text <-> token glue on the host.  Real tokenizer / vocabulary files live in checkpoints that are not available
offline, so every consumer takes an object with ``decode(ids) -> str`` / ``encode(text) -> ids`` and the synthetic
checkpoints use the trivial codecs below."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

import numpy as np


class IdTextCodec:
    """Detokeniser for synthetic checkpoints: token id i -> 't<i>' words (deterministic, invertible)."""

    def decode(self, ids: Sequence[int]) -> str:
        return " ".join(f"t{int(i)}" for i in ids)

    def encode(self, text: str) -> list[int]:
        return [int(w[1:]) for w in text.split() if w.startswith("t") and w[1:].isdigit()]


def load_tokenizer(path: str | None):
    """tokenizers.Tokenizer from ``<path>/tokenizer.json`` when present, else the id codec."""
    import os

    if path and os.path.exists(os.path.join(path, "tokenizer.json")):
        from tokenizers import Tokenizer

        tok = Tokenizer.from_file(os.path.join(path, "tokenizer.json"))

        class _Wrap:
            def decode(self, ids):
                return tok.decode(list(ids), skip_special_tokens=True)

            def encode(self, text):
                return tok.encode(text).ids

        return _Wrap()
    return IdTextCodec()


@dataclass
class WhisperVocab:
    """Special-token ids of a Whisper vocabulary (multilingual large-v3 defaults)."""

    sot: int = 50258
    eot: int = 50257
    transcribe: int = 50360
    notimestamps: int = 50364
    lang_to_id: dict = field(default_factory=lambda: {"en": 50259, "zh": 50260, "de": 50261, "es": 50262, "ru": 50263, "ko": 50264,
                                                      "fr": 50265, "ja": 50266, "pt": 50267, "tr": 50268, "pl": 50269, "ca": 50270,
                                                      "nl": 50271, "ar": 50272, "sv": 50273, "it": 50274, "id": 50275, "hi": 50276})
    suppress: tuple = ()
    begin_suppress: tuple = (220, 50257)

    def prompt(self, language: str | None) -> list[int]:
        lang = -1 if language in (None, "auto") else self.lang_to_id[language]
        return [self.sot, lang, self.transcribe, self.notimestamps]

    def id_to_lang(self, token_id: int) -> str | None:
        for k, v in self.lang_to_id.items():
            if v == token_id:
                return k
        return None


class ByteTextFrontend:
    """Synthetic text side of the TTS prompt: UTF-8 bytes -> rows of a [258, E] table (256 bytes + bos + eos).
    (The real Qwen3-TTS chat template / text embedding + projection are in the unavailable qwen_tts package.)"""

    def __init__(self, table: np.ndarray):
        assert table.shape[0] == 258
        self.table = table.astype(np.float32)

    def embed(self, text: str, max_tokens: int) -> np.ndarray:
        ids = [256] + list(text.encode("utf-8"))[: max(0, max_tokens - 2)] + [257]
        return self.table[np.asarray(ids)]
