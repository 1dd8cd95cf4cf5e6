"""Backend-registry boundary (reference ``backend_registry.py:46-233``): ``BackendSpec`` / ``HandlerContext`` /
``normalize_dataclass_config`` restated with the same names and semantics, plus the three MI355X entries a maintainer adds
to ``STT_BACKENDS`` / ``TTS_BACKENDS``.  Factories import their handler module lazily (the registry import stays light -
reference tests/test_backend_registry.py:468-480)."""

from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass, field, fields
from importlib import import_module
from queue import Queue
from threading import Event
from typing import Any, Callable, Mapping

try:  # the real registry types when the reference package is importable
    from speech_to_speech.backend_registry import BackendCapabilities, BackendSpec, HandlerContext, normalize_dataclass_config  # type: ignore
    USING_REFERENCE_PACKAGE = True
except Exception:  # noqa: BLE001
    USING_REFERENCE_PACKAGE = False

    @dataclass(frozen=True)
    class BackendCapabilities:
        bypasses_transcription_notifier: bool = False
        supports_audio_input: bool = False
        supports_llm_proxy: bool = False

    @dataclass(frozen=True)
    class HandlerContext:
        stop_event: Event
        queue_in: Queue
        queue_out: Queue
        text_output_queue: Queue
        should_listen: Event
        cancel_scope: Any
        speculative_turns: Any
        pipeline_index: int
        sample_rate: int
        enable_live_transcription: bool
        live_transcription_update_interval: float

    def normalize_dataclass_config(config: Any, prefix: str | None = None) -> dict[str, Any]:
        """Strip the ``<prefix>_`` marker from field names and fold ``gen_*`` fields into ``gen_kwargs``."""
        normalized: dict[str, Any] = {}
        generation: dict[str, Any] = {}
        marker = f"{prefix}_" if prefix else None
        for f in fields(config):
            name, value = f.name, deepcopy(getattr(config, f.name))
            if marker and name.startswith(marker):
                name = name[len(marker):]
            if name == "gen_kwargs" and isinstance(value, Mapping):
                generation.update(value)
            elif name.startswith("gen_"):
                generation[name[len("gen_"):]] = value
            else:
                normalized[name] = value
        normalized["gen_kwargs"] = generation
        return normalized

    @dataclass(frozen=True)
    class BackendSpec:
        name: str
        kind: str
        config_type: type
        create_handler: Callable[[Any, Mapping[str, Any]], Any]
        config_prefix: str | None = None
        normalize_config: Callable[[Any], dict] | None = None
        required_extra: str | None = None
        capabilities: BackendCapabilities = field(default_factory=BackendCapabilities)

        def normalize(self, config: Any) -> dict[str, Any]:
            if not isinstance(config, self.config_type):
                raise TypeError(f"Backend {self.name!r} expects {self.config_type.__name__}, got {type(config).__name__}.")
            if self.normalize_config is not None:
                return self.normalize_config(config)
            return normalize_dataclass_config(config, self.config_prefix)


from ..arguments import (EchoLLMHandlerArguments, MI355XParakeetSTTHandlerArguments, MI355XQwen3TTSHandlerArguments,  # noqa: E402
                         MI355XWhisperSTTHandlerArguments)


def _load(module: str, cls: str):
    return getattr(import_module(module), cls)


def _create_stt(module: str, cls: str, extra_from_context: bool):
    def factory(context: Any, config: Mapping[str, Any]):
        kwargs = dict(config)
        if extra_from_context:
            kwargs["enable_live_transcription"] = context.enable_live_transcription
            kwargs["live_transcription_update_interval"] = context.live_transcription_update_interval
        handler = _load(module, cls)(context.stop_event, queue_in=context.queue_in, queue_out=context.queue_out, setup_kwargs=kwargs)
        handler.speculative_turns = context.speculative_turns
        handler.pipeline_index = context.pipeline_index
        return handler

    return factory


def _create_tts(context: Any, config: Mapping[str, Any]):
    kwargs = dict(config)
    kwargs["cancel_scope"] = context.cancel_scope
    kwargs["speculative_turns"] = context.speculative_turns
    handler = _load("speech_to_speech_amd.handlers.qwen3_tts", "MI355XQwen3TTSHandler")(
        context.stop_event, queue_in=context.queue_in, queue_out=context.queue_out, setup_args=(context.should_listen,), setup_kwargs=kwargs)
    handler.pipeline_index = context.pipeline_index
    return handler


MI355X_STT_BACKENDS = (
    BackendSpec("mi355x-whisper", "stt", MI355XWhisperSTTHandlerArguments,
                _create_stt("speech_to_speech_amd.handlers.whisper", "MI355XWhisperSTTHandler", False), config_prefix="stt"),
    BackendSpec("mi355x-parakeet", "stt", MI355XParakeetSTTHandlerArguments,
                _create_stt("speech_to_speech_amd.handlers.parakeet", "MI355XParakeetSTTHandler", True), config_prefix="parakeet_tdt"),
)
MI355X_TTS_BACKENDS = (BackendSpec("mi355x-qwen3", "tts", MI355XQwen3TTSHandlerArguments, _create_tts, config_prefix="qwen3_tts"),)


def _create_echo_llm(context: Any, config: Mapping[str, Any]):
    kwargs = dict(config)
    kwargs["cancel_scope"] = context.cancel_scope
    kwargs["speculative_turns"] = context.speculative_turns
    handler = _load("speech_to_speech_amd.handlers.echo_llm", "EchoLLMHandler")(context.stop_event, queue_in=context.queue_in, queue_out=context.queue_out,
                                                                                setup_kwargs=kwargs)
    handler.pipeline_index = context.pipeline_index
    return handler


# the stub LLM of BASELINE config C1 / the benchmarks (precedent tests/test_backend_registry.py:153-170)
ECHO_LLM_BACKENDS = (BackendSpec("echo", "llm", EchoLLMHandlerArguments, _create_echo_llm, config_prefix="echo_llm"),)


def register_mi355x_backends(stt_registry: dict, tts_registry: dict, llm_registry: dict | None = None) -> None:
    """The lines a maintainer adds after STT_BACKENDS / TTS_BACKENDS / LLM_BACKENDS are built."""
    for spec in MI355X_STT_BACKENDS:
        stt_registry[spec.name] = spec
    for spec in MI355X_TTS_BACKENDS:
        tts_registry[spec.name] = spec
    if llm_registry is not None:
        for spec in ECHO_LLM_BACKENDS:
            llm_registry[spec.name] = spec
