"""``--tts mi355x-qwen3``: drop-in for the reference's Qwen3TTSHandler (TTS/qwen3_tts_handler.py) on the MI355X session
pool.  Kept from the reference contract (tests/test_qwen3_tts_handler_backend.py:505-735, 915-964):
  * ``EndOfResponse`` -> ``AUDIO_RESPONSE_DONE``; never touches ``should_listen``;
  * output = int16 numpy blocks of ``blocksize`` (512) samples @ 16 kHz, leading silence trimmed with 40 ms pre-roll,
    tail zero-padded;
  * the cancel generation captured at the start of the utterance is checked between chunks;
  * the frame budget estimator (words / chars / CJK chars / punctuation -> seconds -> 12.5 Hz * 1.35 margin,
    >= 360, multiple of the chunk size, <= cap) and the coalescing of already-queued TTSInputs of the same response;
  * the three generation modes and their dispatch (:841-858): voice clone when a reference is set (`ref_spk` [+ `ref_rvq`, `ref_text`],
    `xvec_only`), else custom voice (speaker [+ instruct]) or voice design (instruct) by the checkpoint's model type, else the
    reference's "Base model requires a voice-clone reference" error; the per-session voice override (:506-553) and
    `on_session_end()` / `cleanup()` (:1005-1034)."""

from __future__ import annotations

import logging
import math
import os
import re
import unicodedata
from time import perf_counter
from typing import Any, Iterator, Optional

import numpy as np
import torch

from ..boundary import (AUDIO_RESPONSE_DONE, PIPELINE_END, SESSION_END, AssistantOutputEvent, AssistantTextPart, BaseHandler, EndOfResponse, TTSInput,
                        is_control_message)
from ..checkpoints import TTS_DIMS, load_safetensors_dir, parse_model_name
from ..session_pool import SessionPool, release_pool_on_failure
from ..text import ByteTextFrontend, PromptTooLong
from ..tts import AudioBlocker, TTSPrompt

logger = logging.getLogger(__name__)

DEFAULT_MAX_NEW_TOKENS = 1536
MIN_UTTERANCE_TOKENS = 360
TOKENS_PER_SECOND = 12.5
WORDS_PER_SECOND, CHARS_PER_SECOND, CJK_CHARS_PER_SECOND = 2.6, 14.0, 5.5
SAFETY_MARGIN, BASE_PROMPT_SECONDS, PUNCTUATION_PAUSE_SECONDS = 1.35, 1.0, 0.5
# language codes a session may carry -> the names a Qwen3-TTS checkpoint's layout uses (same table as the reference's
# QWEN3_LANGUAGE_ALIASES, TTS/qwen3_tts_handler.py:68-89; unknown values pass through lower-cased)
LANGUAGE_ALIASES = {"zh": "chinese", "zh-cn": "chinese", "zh-hans": "chinese", "zh-hant": "chinese", "cmn": "chinese", "en": "english", "en-us": "english",
                    "en-gb": "english", "ja": "japanese", "jp": "japanese", "ko": "korean", "kr": "korean", "de": "german", "fr": "french", "ru": "russian",
                    "pt": "portuguese", "pt-br": "portuguese", "pt-pt": "portuguese", "es": "spanish", "it": "italian"}


def normalize_language(language: Optional[str]) -> str:
    """`_normalize_language` of the reference (TTS/qwen3_tts_handler.py:394-400): None / empty -> "auto"; '_' -> '-', lower case, aliases."""
    if language is None:
        return "auto"
    n = str(language).strip().replace("_", "-").lower()
    return LANGUAGE_ALIASES.get(n, n) if n else "auto"


def resolve_voice(layout: Any, speaker: Optional[str], language: Optional[str]) -> tuple[Optional[str], Optional[str]]:
    """(speaker key, language key) of a checkpoint layout for a session's settings: the language goes through `normalize_language` and is
    matched case-insensitively ("auto" / a language the layout does not list -> no language row, like the reference's 'auto'); the speaker is
    matched case-insensitively, an unknown name is an error listing the supported ones, and no speaker on a layout that HAS speakers
    resolves to its first one (reference `_resolve_speaker` :596-610, `_validate...` :525-531)."""
    lang_ids = dict(getattr(layout, "language_ids", None) or {})
    spk_ids = dict(getattr(layout, "speaker_ids", None) or {})
    lang = normalize_language(language)
    lang_key = None
    if lang != "auto" and lang_ids:
        by_lower = {k.lower(): k for k in lang_ids}
        lang_key = by_lower.get(lang)
        if lang_key is None:
            logger.warning("Qwen3-TTS: language %r is not in the checkpoint's language table %s; using auto", language, sorted(lang_ids))
    spk_key = None
    if spk_ids:
        if speaker:
            by_lower = {k.lower(): k for k in spk_ids}
            spk_key = by_lower.get(str(speaker).strip().lower())
            if spk_key is None:
                raise ValueError(f"speaker {speaker!r} is not supported by this checkpoint; supported: {', '.join(sorted(spk_ids, key=str.lower)) or 'unknown'}")
        else:
            spk_key = next(iter(spk_ids))
    return spk_key, lang_key


def infer_model_type(model_name: Any) -> str:
    """`_infer_model_type_from_name` of the reference (:402-414): 'voicedesign' / 'customvoice' / 'base' in the model name, else 'base'."""
    name = str(model_name or "").lower()
    for needle, kind in (("voicedesign", "voice_design"), ("customvoice", "custom_voice"), ("base", "base")):
        if needle in name:
            return kind
    return "base"


def load_speaker_embedding(path: Any, width: int) -> np.ndarray:
    """A precomputed speaker embedding (`qwen3_tts_ref_spk`): `.npy` (float vector) or raw little-endian float32 - `width` values, the
    talker's row width (the x-vector AFTER the speaker encoder's projection).  The reference hands the path to qwentts.cpp, whose .spk
    container is not documented offline: a file of any other size is refused, never reinterpreted."""
    p = os.fspath(path)
    v = np.load(p).astype(np.float32).reshape(-1) if p.endswith(".npy") else np.fromfile(p, dtype="<f4")
    if v.shape[0] != width:
        raise ValueError(f"{p}: {v.shape[0]} float32 values, expected the talker width {width} (a raw / .npy speaker embedding)")
    return v


def load_reference_codes(path: Any, n_groups: int) -> np.ndarray:
    """Precomputed codec frames of the reference audio (`qwen3_tts_ref_rvq`): `.npy` int array [frames, groups] (or [groups, frames]) or
    raw little-endian int32 laid out frame-major."""
    p = os.fspath(path)
    c = np.load(p) if p.endswith(".npy") else np.fromfile(p, dtype="<i4")
    c = np.asarray(c, dtype=np.int64)
    if c.ndim == 1:
        if c.shape[0] % n_groups:
            raise ValueError(f"{p}: {c.shape[0]} ids are not a whole number of {n_groups}-group frames")
        c = c.reshape(-1, n_groups)
    elif c.ndim == 2 and c.shape[1] != n_groups and c.shape[0] == n_groups:
        c = c.T
    if c.ndim != 2 or c.shape[1] != n_groups:
        raise ValueError(f"{p}: expected [frames, {n_groups}] codec ids, got {c.shape}")
    return c


CJK_PATTERN = re.compile(r"[぀-ヿ㐀-䶿一-鿿가-힯豈-﫿\U00020000-\U0002fa1f]")


def estimate_max_new_tokens(text: Optional[str], chunk_size: int = 8, cap: int = DEFAULT_MAX_NEW_TOKENS) -> int:
    """Codec-frame budget for one utterance (reference _estimate_max_new_tokens, TTS/qwen3_tts_handler.py:615-658)."""
    text = (text or "").strip()
    chunk_size, cap = max(1, int(chunk_size)), max(1, int(cap))
    if not text:
        return min(cap, MIN_UTTERANCE_TOKENS)
    words = len(re.findall(r"\w+", text, flags=re.UNICODE))
    chars = len(re.sub(r"\s+", "", text))
    cjk = len(CJK_PATTERN.findall(text))
    seconds = max(words / WORDS_PER_SECOND if words else 0.0, chars / CHARS_PER_SECOND if chars else 0.0,
                  cjk / CJK_CHARS_PER_SECOND if cjk else 0.0)
    seconds += sum(unicodedata.category(ch).startswith("P") for ch in text) * PUNCTUATION_PAUSE_SECONDS + BASE_PROMPT_SECONDS
    tokens = math.ceil(seconds * TOKENS_PER_SECOND * SAFETY_MARGIN)
    aligned = max(chunk_size, math.ceil(tokens / chunk_size) * chunk_size)
    return min(cap, max(MIN_UTTERANCE_TOKENS, aligned))


def build_tts_engine(model_name: str, device: str, dtype: str, max_batch: int, max_new_tokens: int, ctx_frames: int, chunk_size: int):
    from ..synth_weights import _normal, qwen3_tts_state_dict
    from ..tts import TTSEngine

    kind, key, seed = parse_model_name(model_name)
    if kind == "dir":
        # a checkpoint directory: config.json -> dims, safetensors -> the converter's names (name_map.json may extend the rules),
        # prompt_layout.json -> prompt assembly; see qwen3_tts_checkpoint.py for what is verified and what is not
        from ..qwen3_tts_checkpoint import TableTextFrontend, load_qwen3_tts_checkpoint
        from ..text import load_tokenizer

        dims, sd, layout, text = load_qwen3_tts_checkpoint(key)
        pad = text["text.pad_embed"] if text and "text.pad_embed" in text else torch.zeros(dims.t_hidden)
        eng = TTSEngine(dims, sd, dtype=getattr(torch, dtype), max_batch=max_batch, max_prompt=128, max_frames=max_new_tokens,
                        max_codec_frames=ctx_frames + chunk_size, pad_embed=pad, device=device)
        if not text or "text.embedding.weight" not in text:
            raise KeyError(f"{key}: the checkpoint has no text embedding table ('text.embedding.weight' after name mapping)")
        proj = None
        if "text.proj.0.weight" in text:
            proj = [(text["text.proj.0.weight"].numpy(), text["text.proj.0.bias"].numpy()), (text["text.proj.1.weight"].numpy(), text["text.proj.1.bias"].numpy())]
        tok = load_tokenizer(key)
        from ..text import IdTextCodec

        eng.text_frontend = TableTextFrontend(text["text.embedding.weight"].numpy(), layout,
                                              tok if hasattr(tok, "encode") and not isinstance(tok, IdTextCodec) else None, proj,
                                              codec_table=sd["talker.model.codec_embedding.weight"].float().numpy(),
                                              group_tables=[sd[f"code_predictor.model.codec_embedding.{g}.weight"].float().numpy() for g in range(dims.n_groups - 1)])
        return eng
    dims = TTS_DIMS[key]()
    sd = qwen3_tts_state_dict(dims, seed=seed or 5)
    pad = _normal(seed or 5, "tts_pad_embed", (dims.t_hidden,), 0.5)
    eng = TTSEngine(dims, sd, dtype=getattr(torch, dtype), max_batch=max_batch, max_prompt=128, max_frames=max_new_tokens,
                    max_codec_frames=ctx_frames + chunk_size, pad_embed=pad, device=device)
    eng.text_frontend = ByteTextFrontend(_normal(seed or 5, "text_table", (258, dims.t_hidden), 1.0).numpy())
    return eng


class MI355XQwen3TTSHandler(BaseHandler):
    @release_pool_on_failure
    def setup(self, should_listen: Any = None, model_name: str = "synthetic:test-tiny", device: str = "cuda:0", dtype: str = "bfloat16",
              speaker: str | None = None, language: str | None = "auto", instruct: str | None = None, streaming_chunk_size: int = 8,
              max_new_tokens: int = DEFAULT_MAX_NEW_TOKENS, blocksize: int = 512, max_batch: int = 64, batch_window_ms: float = 4.0,
              codec_context_frames: int = 25, lanes: int = 3, cancel_scope: Any = None, speculative_turns: Any = None, gen_kwargs: dict | None = None,
              pool: SessionPool | None = None, lead_s: float = 0.0, admit_wait_ms: float = 0.0, ref_audio: Any = None, ref_spk: Any = None,
              ref_rvq: Any = None, ref_text: str | None = None, xvec_only: bool = False, non_streaming_mode: bool | None = None,
              model_type: str | None = None, **_ignored: Any) -> None:
        self.should_listen = should_listen          # owned by the reference runtime; never touched here
        self.model_name = model_name
        self.ref_audio, self.ref_spk, self.ref_rvq, self.ref_text, self.xvec_only = ref_audio, ref_spk, ref_rvq, ref_text, bool(xvec_only)
        # non_streaming_mode (reference arguments :103-108): True (the reference's default) pre-fills the full target text before decode; False =
        # streaming-text mode: the prefill ends with the first text token and the rest of the text reaches the talker one row per frame
        # (engine: TTSPrompt.trailing -> s2s_tts_set_trailing; the split and the per-step rule are the architectural sibling's,
        # modeling_qwen3_omni_moe.py:3744-3801,3169-3174 - the pinned backend's own layout is not available offline)
        self.non_streaming_mode = True if non_streaming_mode is None else bool(non_streaming_mode)
        self._model_type_override = model_type
        self.cancel_scope, self.speculative_turns = cancel_scope, speculative_turns
        self.streaming_chunk_size, self.max_new_tokens, self.blocksize = int(streaming_chunk_size), int(max_new_tokens), int(blocksize)
        self.speaker, self.language, self.instruct = speaker, language, instruct
        # what on_session_end() restores (reference :206-209, :1005-1009)
        self._initial_speaker, self._initial_ref_audio, self._initial_ref_spk, self._initial_ref_rvq = speaker, ref_audio, ref_spk, ref_rvq
        self._voice_cache: dict = {}
        self._owns_pool = pool is None
        self.pool = pool or SessionPool.for_device(device, max_batch=max_batch, batch_window_ms=batch_window_ms,
                                                  tts_chunk_size=self.streaming_chunk_size, tts_ctx_frames=codec_context_frames, tts_lanes=lanes)
        self.pool.attach(self)                 # every user is counted, also one that was handed its pool (session_pool.attach)
        if float(lead_s) > 0.0:             # paced frame loop (0 = eager, the pool's default)
            self.pool.configure(tts_lead_s=float(lead_s), tts_admit_wait_s=float(admit_wait_ms) * 1e-3)
        self.pool.ensure_engine("tts", lambda: build_tts_engine(model_name, device, dtype, max_batch, self.max_new_tokens, codec_context_frames,
                                                               self.streaming_chunk_size))
        eng = self.pool.engine("tts")
        if hasattr(eng, "configure_sampling"):
            eng.configure_sampling(gen_kwargs)      # before the lanes clone the engine
        # the session's voice is resolved ONCE, here (a bad speaker name fails setup, not every turn): language aliases ('en' -> 'english'),
        # case-insensitive speaker, first supported speaker when none is given
        layout = getattr(getattr(eng, "text_frontend", None), "layout", None)
        self._voice = resolve_voice(layout, speaker, language) if layout is not None else (None, None)
        if self._has_voice_clone_reference():
            self._voice_reference()      # a reference that cannot be loaded fails setup, not every turn
        self.pool.warm_tts_lanes()       # load + warm up synchronously, like the reference's setup()

    # ---- generation modes (reference :402-414, :441-446, :506-553, :574-610, :841-858) -----------------------------------------------
    def _layout(self) -> Any:
        pool = getattr(self, "pool", None)
        return getattr(getattr(pool.engine("tts"), "text_frontend", None), "layout", None) if pool is not None else None

    def _model_type(self) -> str:
        """The checkpoint's kind: an explicit `model_type` argument, else prompt_layout.json's `model_type`, else the reference's
        inference from the model name (:402-414) when the name says anything."""
        override = getattr(self, "_model_type_override", None)
        if override:
            return override
        lay = self._layout()
        if lay is not None and getattr(lay, "model_type", None):
            return lay.model_type
        name = str(getattr(self, "model_name", None) or "").lower()
        if any(k in name for k in ("voicedesign", "customvoice", "base")):
            return infer_model_type(name)
        # no kind anywhere (synthetic models, exported test checkpoints): the reference's DEFAULT checkpoint is the CustomVoice one
        # (arguments_classes/qwen3_tts_arguments.py) - plain text -> speech with the layout's first speaker, no reference needed
        return "custom_voice"

    def _supported_speakers(self) -> Optional[list[str]]:
        lay = self._layout()
        ids = getattr(lay, "speaker_ids", None) if lay is not None else None
        return list(ids) if ids else None

    def _has_voice_clone_reference(self) -> bool:
        return bool(getattr(self, "ref_audio", None) or getattr(self, "ref_spk", None))

    def _clear_cached_voice_reference(self) -> None:
        self.ref_spk = None
        self.ref_rvq = None

    def _voice_reference(self) -> tuple[np.ndarray, Optional[np.ndarray]]:
        """(speaker row, reference code rows | None) of the current voice-clone reference, loaded once per (ref_spk, ref_rvq, xvec_only).
        A reference given only as AUDIO needs the speaker encoder and the codec ENCODER of the tokenizer - neither is on the path this
        backend builds (SURVEY 8 a13 is the decoder): it is refused with the way out, not approximated."""
        spk, rvq = getattr(self, "ref_spk", None), getattr(self, "ref_rvq", None)
        if not spk:
            raise ValueError("mi355x-qwen3 voice cloning needs a precomputed speaker embedding (qwen3_tts_ref_spk, and qwen3_tts_ref_rvq for "
                             f"in-context cloning): extracting them from qwen3_tts_ref_audio={getattr(self, 'ref_audio', None)!r} takes the speaker "
                             "encoder / codec encoder, which this backend does not implement")
        xvec_only = bool(getattr(self, "xvec_only", False))
        key = (os.fspath(spk), os.fspath(rvq) if rvq else None, xvec_only)
        cache = self.__dict__.setdefault("_voice_cache", {})
        if key not in cache:
            eng = self.pool.engine("tts")
            fe = eng.text_frontend
            row = load_speaker_embedding(spk, int(eng.dims.t_hidden))
            rows = None
            if rvq and not xvec_only:
                if not hasattr(fe, "code_rows"):
                    raise ValueError("in-context voice cloning (qwen3_tts_ref_rvq) needs a checkpoint front-end; use xvec_only with a synthetic model")
                rows = fe.code_rows(load_reference_codes(rvq, int(eng.dims.n_groups)))
            cache.clear()
            cache[key] = (row, rows)
        return cache[key]

    def _apply_session_voice_override(self, model_type: str, runtime_config: Any = None, response: Any = None) -> None:
        """`response.audio.output.voice`, else `runtime_config.session.audio.output.voice` (reference :506-553).  Custom voice: a supported
        speaker (case-insensitive) replaces the session's speaker and drops any clone reference; an unsupported name is ignored with a
        warning.  Other kinds: the voice must be a reference file - here a `.spk` / `.npy` speaker embedding, since raw audio cannot be
        encoded by this backend - else it is ignored with a warning."""
        def _voice(obj: Any, *path: str) -> Optional[str]:
            for name in path:
                obj = getattr(obj, name, None)
                if obj is None:
                    return None
            return str(obj) if obj else None

        session_voice = _voice(response, "audio", "output", "voice") or _voice(runtime_config, "session", "audio", "output", "voice")
        if not session_voice:
            return
        if model_type == "custom_voice":
            supported = self._supported_speakers()
            if supported is not None:
                match = {s.lower(): s for s in supported}.get(session_voice.lower())
                if match is None:
                    logger.warning("Ignoring Qwen3-TTS session voice override %r because it is not a supported CustomVoice speaker. "
                                   "Supported speakers: %s", session_voice, ", ".join(sorted(supported, key=str.lower)) or "unknown")
                    return
                session_voice = match
            if session_voice != getattr(self, "speaker", None):
                self._voice = None       # resolved again on the next prompt
            self.speaker = session_voice
            self.ref_audio = None
            self._clear_cached_voice_reference()
            return
        if os.path.isfile(os.path.expanduser(session_voice)) and session_voice.lower().endswith((".spk", ".npy")):
            self.ref_spk, self.ref_rvq, self.ref_audio = os.path.expanduser(session_voice), None, None
            return
        logger.warning("Ignoring Qwen3-TTS session voice override because it is not a speaker-embedding file path: %r", session_voice)

    def on_session_end(self) -> None:
        """Reference :1005-1009: the voice a session selected (speaker / clone reference) does not carry over to the next session of this
        pipeline unit."""
        self.speaker = getattr(self, "_initial_speaker", None)
        self.ref_audio = getattr(self, "_initial_ref_audio", None)
        self.ref_spk = getattr(self, "_initial_ref_spk", None)
        self.ref_rvq = getattr(self, "_initial_ref_rvq", None)
        self._voice = None
        logger.debug("MI355X Qwen3-TTS session state reset")

    def cleanup(self) -> None:
        """Reference :1011-1034 deletes its model and empties the allocator cache; here the engines belong to the device's SessionPool: the
        handler lets go, the pool closes them (weights, KV arenas, graphs) when its last `for_device` handler is gone."""
        try:
            self.__dict__.setdefault("_voice_cache", {}).clear()
            pool, self.pool = getattr(self, "pool", None), None
            if pool is not None and pool.detach(self) and torch.cuda.is_available():
                torch.cuda.empty_cache()
            logger.info("MI355X Qwen3-TTS handler cleaned up")
        except Exception as e:  # noqa: BLE001
            logger.warning("Cleanup error: %s", e)

    def _estimate_max_new_tokens(self, text: Optional[str]) -> int:
        return estimate_max_new_tokens(text, self.streaming_chunk_size, self.max_new_tokens)

    def _coalesce_pending_tts_input(self, current: Any) -> tuple[str, Optional[str]]:
        """Merge already-queued TTSInputs of the same response into one synthesis call - the reference's rules
        (TTS/qwen3_tts_handler.py:751-810): same (turn_id, turn_revision, cancel_generation, response_key); stop at
        SESSION_END / PIPELINE_END / EndOfResponse, at anything that is not a TTSInput, and at a TTSInput whose language
        code differs from the one collected so far; text-only AssistantOutputEvents of the same response that sit between
        the absorbed inputs are taken out and forwarded BEFORE the audio (protocol order text -> audio)."""
        q = self.queue_in
        text, lang = current.text, current.language_code
        if not hasattr(q, "mutex") or not hasattr(q, "queue"):
            return text, lang
        parts = [text.strip()] if text and text.strip() else []
        text_events: list = []

        def same_response(item: Any) -> bool:
            return (item.turn_id == current.turn_id and item.turn_revision == current.turn_revision
                    and item.cancel_generation == current.cancel_generation and item.response_key == current.response_key)

        with q.mutex:
            while q.queue:
                nxt = q.queue[0]
                if is_control_message(nxt, SESSION_END.kind):
                    break
                if isinstance(nxt, bytes) and nxt == PIPELINE_END:
                    break
                if isinstance(nxt, EndOfResponse):
                    break
                if isinstance(nxt, AssistantOutputEvent):
                    if not same_response(nxt) or any(not isinstance(part, AssistantTextPart) for part in nxt.parts):
                        break
                    text_events.append(q.queue.popleft())
                    continue
                if not isinstance(nxt, TTSInput):
                    break
                if not same_response(nxt):
                    break
                if lang is not None and nxt.language_code is not None and nxt.language_code != lang:
                    break
                q.queue.popleft()
                if nxt.text.strip():
                    parts.append(nxt.text.strip())
                if lang is None:
                    lang = nxt.language_code
        for event in text_events:
            self.queue_out.put(event)
        return " ".join(parts).strip(), lang

    def _build_prompt(self, text: str) -> TTSPrompt:
        """One utterance's prefill rows in the mode the reference's process() would pick (:847-858)."""
        eng = self.pool.engine("tts")
        fe = eng.text_frontend
        has_layout = getattr(fe, "layout", None) is not None
        kw: dict = {}
        lang = None
        if has_layout:
            voice = getattr(self, "_voice", None)
            if voice is None:                            # (handlers built without setup(), or a session override: resolve on first use)
                voice = self._voice = resolve_voice(fe.layout, getattr(self, "speaker", None), getattr(self, "language", None))
            lang = voice[1]
            kw["language"] = lang
        instruct = getattr(self, "instruct", None)
        if self._has_voice_clone_reference():
            row, ref_rows = self._voice_reference()
            kw.update(speaker_row=row, ref_rows=ref_rows, ref_text=getattr(self, "ref_text", None) if ref_rows is not None else None)
        else:
            kind = self._model_type()
            if kind == "custom_voice":
                if has_layout:
                    if getattr(fe.layout, "speaker_ids", None) and not self._voice[0]:
                        raise ValueError("CustomVoice generation requires a speaker. Set qwen3_tts_speaker or use a voice-clone model with ref_audio.")
                    kw["speaker"] = self._voice[0]
                if instruct:
                    kw["instruct"] = instruct
            elif kind == "voice_design":
                if instruct:
                    kw["instruct"] = instruct
            else:
                raise ValueError("Qwen3-TTS Base model requires a voice-clone reference. Provide qwen3_tts_ref_audio or qwen3_tts_ref_spk, "
                                 "or use a CustomVoice/VoiceDesign model.")
        if not getattr(self, "non_streaming_mode", True):
            rows, trailing = fe.embed(text, eng.max_prompt, streaming=True, **kw)
            return TTSPrompt(rows, trailing)
        return TTSPrompt(fe.embed(text, eng.max_prompt, **kw))

    def _build_prompts(self, text: str) -> list[tuple[str, TTSPrompt]]:
        """One prompt per piece of text that fits the engine's prefill budget.  An over-long input is split at sentence ends, then at
        spaces (never truncated: the cut-off tail would silently not be spoken); the pieces are synthesised one after the other."""
        try:
            return [(text, self._build_prompt(text))]
        except PromptTooLong:
            if len(text) < 2:        # nothing left to split: the fixed rows alone exceed the budget
                raise
        cut = -1
        mid = len(text) // 2
        for pat in (r"[.!?\u3002\uff01\uff1f]\s", r"[,;:]\s", r"\s"):
            pos = [m.end() for m in re.finditer(pat, text)]
            if pos:
                cut = min(pos, key=lambda p: abs(p - mid))
                if 0 < cut < len(text):
                    break
        if not 0 < cut < len(text):
            cut = mid
        left, right = text[:cut].strip(), text[cut:].strip()
        return (self._build_prompts(left) if left else []) + (self._build_prompts(right) if right else [])

    def process(self, tts_input: Any) -> Iterator[Any]:
        # stale-turn handling exactly as the reference's process() (TTS/qwen3_tts_handler.py:812-836): the tracker's commit()
        # returns None - its result is never a filter
        speculative_turns = getattr(self, "speculative_turns", None)
        if isinstance(tts_input, EndOfResponse):
            if speculative_turns and not speculative_turns.is_latest_after_reopen_grace(tts_input.turn_id, tts_input.turn_revision):
                if tts_input.response_key is None:
                    return
                tts_input.cleanup_only = True
            yield AUDIO_RESPONSE_DONE
            return
        if speculative_turns and not speculative_turns.is_latest_after_reopen_grace(tts_input.turn_id, tts_input.turn_revision):
            logger.debug("Dropping stale TTS input for turn=%s rev=%s", tts_input.turn_id, tts_input.turn_revision)
            return
        if speculative_turns:
            speculative_turns.commit(tts_input.turn_id, tts_input.turn_revision)
        text, _lang = self._coalesce_pending_tts_input(tts_input)
        text = text or "Hello."          # the reference synthesises "Hello." for an empty response text
        if getattr(tts_input, "runtime_config", None) is not None or getattr(tts_input, "response", None) is not None:
            self._apply_session_voice_override(self._model_type(), tts_input.runtime_config, tts_input.response)
        gen = self.cancel_scope.generation if self.cancel_scope is not None else None
        start = perf_counter()
        try:
            blocker = AudioBlocker(self.blocksize)
            first, total = True, 0
            for piece, prompt in self._build_prompts(text):
                budget = self._estimate_max_new_tokens(piece)
                # min_new_tokens: benchmark knob (fixed-length replies, eos masked); 0 on the serving path
                min_frames = min(budget, int(getattr(self, "min_new_tokens", 0) or 0))
                chunks, cancel = (self.pool.synthesize(prompt, budget, min_frames=min_frames) if min_frames else self.pool.synthesize(prompt, budget))
                for chunk in chunks:
                    if gen is not None and self.cancel_scope.is_stale(gen):
                        cancel.set()
                        logger.info("TTS generation cancelled (interruption)")
                        return
                    if first:
                        logger.info("MI355X Qwen3-TTS TTFA: %.3fs", perf_counter() - start)
                        if tts_input.speech_stopped_at_s is not None:
                            logger.info("Last speech detected to first speech out: %.3fs", perf_counter() - tts_input.speech_stopped_at_s)
                        first = False
                    for block in blocker.push(np.asarray(chunk, dtype=np.int16)):
                        total += len(block)
                        started = getattr(chunks, "started", None)
                        if started is not None and not started.is_set():
                            started.set()        # first audible block: from here on the pool may pace this reply by its audio lead
                        yield block
            for block in blocker.flush():
                total += len(block)
                yield block
            dt = perf_counter() - start
            logger.info("MI355X Qwen3-TTS generated %.2fs audio in %.2fs (RTF: %.2f)", total / 16000, dt, (total / 16000) / dt if dt > 0 else 0)
        except Exception as e:  # noqa: BLE001  (reference policy: generation failure -> no audio, thread keeps running)
            logger.error("Qwen3-TTS generation failed: %s", e, exc_info=True)
