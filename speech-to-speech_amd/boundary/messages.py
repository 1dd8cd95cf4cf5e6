"""Field- and behaviour-compatible restatement of the reference's inter-stage messages.

The reference package cannot be imported offline (``openai`` / ``nltk`` ... are absent - SURVEY.md 7 step 0), so the
boundary types the MI355X handlers consume and produce are restated here with the same class names, tags and field
names (reference ``pipeline/messages.py:45-88, 200-223, 360-361``; ``pipeline/control.py``).  When the real package is
importable, ``boundary/__init__.py`` re-exports ITS classes instead and these are unused.
"""

from __future__ import annotations

from dataclasses import dataclass
from time import perf_counter
from typing import Any, Final, Literal, Optional

import numpy as np
from pydantic import BaseModel, ConfigDict, Field


class PipelineMessage(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    tag: str


class VADAudio(PipelineMessage):
    tag: Literal["vad_audio"] = "vad_audio"
    audio: np.ndarray
    runtime_config: Any = None
    mode: Literal["progressive", "final"] | None = None
    turn_id: str | None = None
    turn_revision: int | None = None
    processing_delay_s: float = 0.0
    created_at_s: float = Field(default_factory=perf_counter)


class PartialTranscription(PipelineMessage):
    tag: Literal["partial_transcription"] = "partial_transcription"
    text: str
    turn_id: str | None = None
    turn_revision: int | None = None


class Transcription(PipelineMessage):
    tag: Literal["transcription"] = "transcription"
    text: str
    language_code: Optional[str] = None
    turn_id: str | None = None
    turn_revision: int | None = None
    speech_stopped_at_s: float | None = None


class TranscriptionFailure(PipelineMessage):
    tag: Literal["transcription_failure"] = "transcription_failure"
    message: str
    turn_id: str | None = None
    turn_revision: int | None = None
    speech_stopped_at_s: float | None = None


class TTSInput(PipelineMessage):
    tag: Literal["tts_input"] = "tts_input"
    text: str
    language_code: Optional[str] = None
    runtime_config: Any = None
    response: Any = None
    turn_id: str | None = None
    turn_revision: int | None = None
    speech_stopped_at_s: float | None = None
    cancel_generation: int | None = None
    response_key: str | None = Field(default=None, exclude=True, repr=False)
    prefetch_transaction: Any = Field(default=None, exclude=True, repr=False)


class GenerateResponseRequest(PipelineMessage):
    """Triggers LLM generation (reference pipeline/messages.py:334-355; the fields the echo stub reads)."""

    tag: Literal["generate_response"] = "generate_response"
    response_key: str = Field(default="", exclude=True, repr=False)
    runtime_config: Any = None
    response: Any = None
    audio: Any = None
    language_code: Optional[str] = None
    turn_id: str | None = None
    turn_revision: int | None = None
    speech_stopped_at_s: float | None = None
    prefetch_transaction: Any = Field(default=None, exclude=True, repr=False)


class LLMResponseChunk(PipelineMessage):
    """One ordered group of assistant output parts (reference pipeline/messages.py:137-157)."""

    tag: Literal["llm_response_chunk"] = "llm_response_chunk"
    parts: list = Field(default_factory=list)
    text: str = ""
    language_code: Optional[str] = None
    tools: list = Field(default_factory=list)
    runtime_config: Any = None
    response: Any = None
    turn_id: str | None = None
    turn_revision: int | None = None
    speech_stopped_at_s: float | None = None
    cancel_generation: int | None = None
    response_key: str | None = Field(default=None, exclude=True, repr=False)
    prefetch_transaction: Any = Field(default=None, exclude=True, repr=False)


class EndOfResponse(PipelineMessage):
    tag: Literal["end_of_response"] = "end_of_response"
    turn_id: str | None = None
    turn_revision: int | None = None
    cancel_generation: int | None = None
    response_key: str | None = Field(default=None, exclude=True, repr=False)
    error: str | None = None
    cleanup_only: bool = False


class AudioOutput(PipelineMessage):
    tag: Literal["audio_output"] = "audio_output"
    audio: bytes | np.ndarray
    cancel_generation: int | None = None
    response_key: str | None = Field(default=None, exclude=True, repr=False)
    cleanup_only: bool = False


class AssistantTextPart(BaseModel):
    """One ordered assistant text part (reference pipeline/messages.py:94-98)."""

    type: Literal["text"] = "text"
    text: str


class PipelineEvent(BaseModel):
    """Base of the protocol-neutral events the realtime router consumes (reference pipeline/events.py:23-33).  They share
    the TTS queue with their audio; handlers forward them untouched so text / tool / audio order is preserved."""

    type: str


class AssistantOutputEvent(PipelineEvent):
    """Ordered assistant output awaiting API-specific serialisation (reference pipeline/events.py:105-131; only the
    fields the TTS handler reads)."""

    type: Literal["assistant_text"] = "assistant_text"
    parts: list = Field(default_factory=list)
    text: str = ""
    turn_id: str | None = None
    turn_revision: int | None = None
    cancel_generation: int | None = None
    response_key: str | None = Field(default=None, exclude=True, repr=False)


AUDIO_RESPONSE_DONE: Final[bytes] = b"__RESPONSE_DONE__"
PIPELINE_END: Final[bytes] = b"END"


@dataclass(frozen=True)
class PipelineControlMessage:
    kind: str


SESSION_END: Final[PipelineControlMessage] = PipelineControlMessage("session_end")


def is_control_message(item: object, kind: str) -> bool:
    return isinstance(item, PipelineControlMessage) and item.kind == kind


class CancelScope:
    """Generation-counter cancellation (reference pipeline/cancel_scope.py): ``cancel()`` bumps the generation, workers
    capture ``generation`` when a response starts and stop as soon as ``is_stale(captured)``."""

    def __init__(self) -> None:
        self._gen = 0
        self._discarding = False
        self._discarded_generation: int | None = None

    @property
    def generation(self) -> int:
        return self._gen

    def cancel(self) -> None:
        self._discarded_generation = self._gen
        self._gen = (self._gen + 1) & 0xFFFFFFFF
        self._discarding = True

    def response_done(self, generation: int | None = None) -> None:
        if generation is not None and self._discarded_generation is not None and generation not in {self._discarded_generation, self._gen}:
            return
        self._discarding = False
        self._discarded_generation = None

    def new_response(self) -> None:
        self._discarding = False
        self._discarded_generation = None

    def is_stale(self, gen: int) -> bool:
        return gen != self._gen

    @property
    def discarding(self) -> bool:
        return self._discarding

    def reset(self) -> None:
        self._gen = 0
        self._discarding = False
        self._discarded_generation = None
