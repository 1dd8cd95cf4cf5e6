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
    """Special-token ids and logits-processor lists of one Whisper checkpoint.

    The reference's ``model.generate`` takes all of this from the checkpoint's ``generation_config.json``
    (transformers generation_whisper.py:1455-1608 init tokens, :1774-1812 SuppressTokens / SuppressTokensAtBegin), so the
    handler builds it with ``from_checkpoint(dir)``; synthetic checkpoints use ``synthetic(vocab_size)``.  The bare
    constructor holds the multilingual large-v3 ids WITHOUT a suppress list (tests / explicit overrides)."""

    sot: int = 50258
    eot: int = 50257
    transcribe: int = 50360
    notimestamps: int = 50364
    lang_to_id: dict = field(default_factory=lambda: {"en": 50259, "zh": 50260, "de": 50261, "es": 50262, "ru": 50263, "ko": 50264,
                                                      "fr": 50265, "ja": 50266, "pt": 50267, "tr": 50268, "pl": 50269, "ca": 50270,
                                                      "nl": 50271, "ar": 50272, "sv": 50273, "it": 50274, "id": 50275, "hi": 50276})
    suppress: tuple = ()
    begin_suppress: tuple = (220, 50257)
    translate: int | None = None

    @property
    def multilingual(self) -> bool:
        return len(self.lang_to_id) > 0

    def prompt(self, language: str | None, task: str = "transcribe") -> list[int]:
        """Forced decoder prefix: <sot><lang><task><notimestamps>; English-only checkpoints (no language tokens): <sot><notimestamps>
        (generation_whisper.py:1455-1608).  language None / "auto" leaves a -1 placeholder that the engine fills by detection."""
        if not self.multilingual:
            return [self.sot, self.notimestamps]
        if language in (None, "auto"):
            lang = -1
        elif language in self.lang_to_id:
            lang = self.lang_to_id[language]
        else:
            raise ValueError(f"language {language!r} is not in this checkpoint's lang_to_id ({sorted(self.lang_to_id)})")
        task_id = self.translate if (task == "translate" and self.translate is not None) else self.transcribe
        return [self.sot, lang, task_id, self.notimestamps]

    def id_to_lang(self, token_id: int) -> str | None:
        for k, v in self.lang_to_id.items():
            if v == token_id:
                return k
        return None

    def validate(self, vocab_size: int) -> "WhisperVocab":
        """Every id must index the model's vocabulary (an out-of-range suppress / language id would write past the masks)."""
        ids = [self.sot, self.eot, self.notimestamps, *self.lang_to_id.values(), *self.suppress, *self.begin_suppress]
        if self.lang_to_id:
            ids.append(self.transcribe)      # English-only checkpoints never emit a task token (prompt() skips it): -1 / absent is fine
        bad = [i for i in ids if not 0 <= int(i) < vocab_size]
        if bad:
            raise ValueError(f"Whisper vocabulary ids {sorted(set(bad))[:8]} are outside the model's {vocab_size}-token vocabulary; "
                             "build the WhisperVocab from the checkpoint (from_checkpoint) or WhisperVocab.synthetic(vocab_size)")
        return self

    @classmethod
    def synthetic(cls, vocab_size: int) -> "WhisperVocab":
        """Vocabulary of a ``synthetic:<dims>`` checkpoint.  Full-size vocabularies keep the real layouts (51 865 / 51 866:
        multilingual, 51 864: English-only tiny.en / base.en ...); small test vocabularies put the special tokens in the top
        20 ids: sot = V-20, languages V-19 .. V-16 (en, fr, de, es), transcribe V-15, notimestamps V-14, eot V-1."""
        V = int(vocab_size)
        if V >= 51865:
            return cls()
        if V == 51864:
            return cls(sot=50257, eot=50256, transcribe=50358, notimestamps=50362, lang_to_id={}, begin_suppress=(220, 50256))
        if V < 32:
            raise ValueError(f"a Whisper vocabulary needs at least 32 tokens, got {V}")
        return cls(sot=V - 20, eot=V - 1, transcribe=V - 15, notimestamps=V - 14,
                   lang_to_id={"en": V - 19, "fr": V - 18, "de": V - 17, "es": V - 16},
                   suppress=tuple(range(V - 20, V - 13)), begin_suppress=(min(220, V - 21), V - 1))

    @classmethod
    def from_checkpoint(cls, path: str) -> "WhisperVocab":
        """From ``<path>/generation_config.json`` (the source transformers' generate() itself reads): suppress_tokens,
        begin_suppress_tokens, lang_to_id, task_to_id, decoder_start_token_id, eos_token_id, no_timestamps_token_id;
        fields a config lacks fall back to ``config.json`` and then to the tokenizer's special tokens."""
        import json
        import os

        def read(name):
            f = os.path.join(path, name)
            if os.path.exists(f):
                with open(f) as fh:
                    return json.load(fh)
            return {}

        gen, cfg = read("generation_config.json"), read("config.json")
        tok_ids: dict[str, int] = {}
        tj = read("tokenizer.json")
        for t in tj.get("added_tokens", []):
            tok_ids[t["content"]] = int(t["id"])

        def pick(*cands, default=None):
            for c in cands:
                if c is not None:
                    return c
            return default

        eot = pick(gen.get("eos_token_id"), cfg.get("eos_token_id"), tok_ids.get("<|endoftext|>"))
        sot = pick(gen.get("decoder_start_token_id"), cfg.get("decoder_start_token_id"), tok_ids.get("<|startoftranscript|>"))
        if isinstance(eot, list):
            eot = eot[0]
        task = gen.get("task_to_id") or {}
        transcribe = pick(task.get("transcribe"), tok_ids.get("<|transcribe|>"))
        translate = pick(task.get("translate"), tok_ids.get("<|translate|>"))
        nots = pick(gen.get("no_timestamps_token_id"), tok_ids.get("<|notimestamps|>"))
        langs = gen.get("lang_to_id")
        if langs is None:
            langs = {k: v for k, v in tok_ids.items() if len(k) in (6, 7) and k.startswith("<|") and k.endswith("|>") and k[2:-2].isalpha()
                     and k[2:-2] not in ("nospeech", "nocaptions")} if gen.get("is_multilingual", cfg.get("vocab_size", 0) >= 51865) else {}
        lang_to_id = {k.strip("<|>"): int(v) for k, v in langs.items()}
        missing = [n for n, v in (("eos_token_id", eot), ("decoder_start_token_id", sot), ("no_timestamps_token_id", nots)) if v is None]
        if lang_to_id and transcribe is None:
            missing.append("task_to_id.transcribe")
        if missing:
            raise ValueError(f"{path}: generation_config.json / config.json / tokenizer.json do not define {missing}")
        return cls(sot=int(sot), eot=int(eot), transcribe=int(transcribe) if transcribe is not None else -1, notimestamps=int(nots),
                   lang_to_id=lang_to_id, suppress=tuple(int(i) for i in (gen.get("suppress_tokens") or cfg.get("suppress_tokens") or ())),
                   begin_suppress=tuple(int(i) for i in (gen.get("begin_suppress_tokens") or cfg.get("begin_suppress_tokens") or (220, int(eot)))),
                   translate=int(translate) if translate is not None else None)


class PromptTooLong(ValueError):
    """The text does not fit the prefill budget: the handler splits it at sentence ends and synthesises the pieces one after the other
    (every other ValueError of a front-end - unknown speaker, bad reference - is a configuration error and is NOT retried on pieces)."""


class ByteTextFrontend:
    """Synthetic text side of the TTS prompt: UTF-8 bytes -> rows of a [258, E] table (256 bytes + bos + eos).
    (The real Qwen3-TTS chat template / text embedding + projection are in the unavailable qwen_tts package.)"""

    def __init__(self, table: np.ndarray):
        assert table.shape[0] == 258
        self.table = table.astype(np.float32)

    def embed(self, text: str, max_tokens: int, instruct: str | None = None, speaker_row: np.ndarray | None = None, ref_text: str | None = None,
              ref_rows: np.ndarray | None = None, streaming: bool = False):
        """[speaker row] [bos instruct eos] [bos (ref_text) text eos] [reference code rows] - the synthetic stand-in of the voice modes'
        prompt (voice design: instruct; voice clone: speaker row, reference text and, unless x-vector only, reference code rows)."""
        raw = list(((ref_text.strip() + " ") if ref_text else "").encode("utf-8")) + list(text.encode("utf-8"))
        ins = [256] + list(instruct.encode("utf-8")) + [257] if instruct else []
        extra = len(ins) + (speaker_row is not None) + (0 if ref_rows is None else len(ref_rows))
        if len(raw) > max(0, max_tokens - 2 - extra):
            # never truncate silently (the tail of the sentence would simply not be spoken): the handler splits long inputs
            raise PromptTooLong(f"TTS input of {len(raw)} bytes exceeds the prompt budget of {max_tokens - 2 - extra}; split the text into shorter sentences")
        ids = ins + [256] + raw + [257]
        rows = self.table[np.asarray(ids)]
        E = rows.shape[1]
        head = np.asarray(speaker_row, np.float32).reshape(1, E) if speaker_row is not None else np.zeros((0, E), np.float32)
        tail = np.asarray(ref_rows, np.float32).reshape(-1, E) if ref_rows is not None else np.zeros((0, E), np.float32)
        if streaming:
            # streaming-text mode (reference non_streaming_mode = False): the prefill ends with the FIRST text token, the rest of the text and the
            # eos row are handed to the frame loop one row per frame (-> (prefill rows, trailing rows); sibling modeling_qwen3_omni_moe.py:3744-3801)
            k = len(ins) + 1 + min(1, len(raw))
            return np.concatenate([head, rows[:k], tail]).astype(np.float32), rows[k:].astype(np.float32)
        return np.concatenate([head, rows, tail]) if len(head) or len(tail) else rows
