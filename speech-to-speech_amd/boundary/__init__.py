"""The drop-in boundary: the reference's own classes when ``speech_to_speech`` is importable, else the in-repo
field/behaviour-compatible restatements (messages.py, base_handler.py)."""

try:  # pragma: no cover - the reference package is not importable offline
    from speech_to_speech.baseHandler import BaseHandler  # type: ignore
    from speech_to_speech.pipeline.cancel_scope import CancelScope  # type: ignore
    from speech_to_speech.pipeline.control import SESSION_END, PipelineControlMessage, is_control_message  # type: ignore
    from speech_to_speech.pipeline.events import AssistantOutputEvent, PipelineEvent  # type: ignore
    from speech_to_speech.pipeline.messages import AssistantTextPart, GenerateResponseRequest, LLMResponseChunk  # type: ignore
    from speech_to_speech.pipeline.messages import (AUDIO_RESPONSE_DONE, PIPELINE_END, AudioOutput, EndOfResponse,  # type: ignore
                                                    PartialTranscription, Transcription, TranscriptionFailure, TTSInput, VADAudio)
    from speech_to_speech.STT.base_stt_handler import BaseSTTHandler  # type: ignore

    USING_REFERENCE_PACKAGE = True
except Exception:  # noqa: BLE001
    from .base_handler import BaseHandler, BaseSTTHandler
    from .messages import (AUDIO_RESPONSE_DONE, PIPELINE_END, SESSION_END, AssistantOutputEvent, AssistantTextPart, AudioOutput, CancelScope,
                           EndOfResponse, GenerateResponseRequest, LLMResponseChunk, PartialTranscription, PipelineControlMessage, PipelineEvent, Transcription, TranscriptionFailure,
                           TTSInput, VADAudio, is_control_message)

    USING_REFERENCE_PACKAGE = False

__all__ = ["AUDIO_RESPONSE_DONE", "PIPELINE_END", "SESSION_END", "AssistantOutputEvent", "AssistantTextPart", "PipelineEvent", "AudioOutput", "GenerateResponseRequest", "LLMResponseChunk", "BaseHandler", "BaseSTTHandler", "CancelScope",
           "EndOfResponse", "PartialTranscription", "PipelineControlMessage", "Transcription", "TranscriptionFailure", "TTSInput",
           "VADAudio", "is_control_message", "USING_REFERENCE_PACKAGE"]
