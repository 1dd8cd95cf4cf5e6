"""``--llm echo``: the stub language model BASELINE config C1 and the benchmarks need.  The reference ships no
stub LLM; its registry makes one a single ``BackendSpec`` (precedent: tests/test_backend_registry.py:153-170).  Contract
honoured (LLM/base_openai_compatible_language_model.py:1032-1110, 810-832, 880-899): for every ``GenerateResponseRequest`` the
cancel generation is captured first, the reply is the last user text of ``request.runtime_config.chat``, emitted as ONE
``LLMResponseChunk`` carrying the request's turn ids / runtime config / response key / prefetch transaction, followed by an
``EndOfResponse`` with the same keys.  ``LMOutputProcessor`` then turns the chunk's text into ``TTSInput``
(LLM/lm_output_processor.py:181-222)."""

from __future__ import annotations

from typing import Any, Iterator

from ..boundary import BaseHandler, EndOfResponse, LLMResponseChunk


def _content_text(content: Any) -> str:
    if content is None:
        return ""
    if isinstance(content, str):
        return content
    if isinstance(content, (list, tuple)):
        parts = []
        for c in content:
            t = c.get("text") or c.get("transcript") if isinstance(c, dict) else (getattr(c, "text", None) or getattr(c, "transcript", None))
            if t:
                parts.append(str(t))
        return " ".join(parts)
    return str(getattr(content, "text", content))


def last_user_text(chat: Any) -> str:
    """Last user message of a conversation: the reference's ``Chat`` (``.buffer`` of realtime conversation items with
    ``role`` / ``content`` parts), a plain list of ``{"role", "content"}`` dicts, or anything with ``to_list()``."""
    if chat is None:
        return ""
    items = getattr(chat, "buffer", None)
    if items is None and hasattr(chat, "to_list"):
        items = chat.to_list()
    if items is None:
        items = chat
    for it in reversed(list(items)):
        role = it.get("role") if isinstance(it, dict) else getattr(it, "role", None)
        if role == "user":
            return _content_text(it.get("content") if isinstance(it, dict) else getattr(it, "content", None)).strip()
    return ""


class EchoLLMHandler(BaseHandler):
    def setup(self, prefix: str = "", cancel_scope: Any = None, speculative_turns: Any = None, **_ignored: Any) -> None:
        self.prefix, self.cancel_scope, self.speculative_turns = prefix, cancel_scope, speculative_turns

    def process(self, request: Any) -> Iterator[Any]:
        gen = self.cancel_scope.generation if getattr(self, "cancel_scope", None) is not None else None
        keys = dict(turn_id=request.turn_id, turn_revision=request.turn_revision, cancel_generation=gen, response_key=request.response_key)
        tracker = getattr(self, "speculative_turns", None)
        stale = tracker is not None and request.turn_id is not None and hasattr(tracker, "is_latest") and not tracker.is_latest(
            request.turn_id, request.turn_revision)
        if not stale:
            text = last_user_text(getattr(request.runtime_config, "chat", None))
            yield LLMResponseChunk(text=f"{getattr(self, 'prefix', '')}{text}", language_code=request.language_code, runtime_config=request.runtime_config,
                                   response=request.response, speech_stopped_at_s=request.speech_stopped_at_s,
                                   prefetch_transaction=request.prefetch_transaction, **keys)
        yield EndOfResponse(**keys)
