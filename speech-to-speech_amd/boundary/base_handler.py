"""Behaviour-compatible restatement of the reference's handler runtime (``baseHandler.py:24-187`` and
``STT/base_stt_handler.py:16-213``): one thread per handler, ``queue_in -> should_process_input -> process() generator ->
queue_out``, SESSION_END / PIPELINE_END handling, audio outputs wrapped into ``AudioOutput``, exceptions in ``process``
logged and swallowed.  Used only when the real ``speech_to_speech`` package is not importable."""

from __future__ import annotations

import logging
from queue import Empty, Queue
from threading import Event
from time import perf_counter
from typing import Any, Iterator

from .messages import (PIPELINE_END, SESSION_END, AudioOutput, EndOfResponse, PartialTranscription, PipelineControlMessage, PipelineEvent, VADAudio,
                       Transcription, TTSInput, is_control_message)

logger = logging.getLogger(__name__)


class BaseHandler:
    def __init__(self, stop_event: Event, queue_in: Queue, queue_out: Queue, setup_args: tuple = (), setup_kwargs: dict | None = None) -> None:
        self.stop_event = stop_event
        self.queue_in = queue_in
        self.queue_out = queue_out
        self.pipeline_index: int | None = None
        self.setup(*setup_args, **(setup_kwargs or {}))
        self._times: list[float] = []

    def setup(self, *args: Any, **kwargs: Any) -> None:
        pass

    def process(self, input: Any) -> Iterator[Any]:
        raise NotImplementedError

    def should_process_input(self, item: Any) -> bool:
        if isinstance(item, TTSInput) and item.prefetch_transaction is not None:
            tx = item.prefetch_transaction
            while not self.stop_event.is_set() and not tx.wait_until_resolved(0.05):
                pass
            if not tx.claimed:
                return False
        cancel_scope = getattr(self, "cancel_scope", None)
        gen = getattr(item, "cancel_generation", None)
        if cancel_scope is not None and gen is not None and not isinstance(item, EndOfResponse) and cancel_scope.is_stale(gen):
            logger.debug("%s: dropping stale input for cancel generation %s", self.__class__.__name__, gen)
            return False
        return True

    def should_emit_output(self, output: Any) -> bool:
        return True

    def before_emit_output(self, output: Any) -> None:
        pass

    def output_for_queue(self, output: Any, source_input: Any) -> Any:
        gen = getattr(source_input, "cancel_generation", None)
        key = getattr(source_input, "response_key", None)
        cleanup_only = getattr(source_input, "cleanup_only", False)
        if (gen is not None or key is not None or cleanup_only) and (isinstance(output, bytes) or hasattr(output, "tobytes")):
            return AudioOutput(audio=output, cancel_generation=gen, response_key=key, cleanup_only=cleanup_only)
        return output

    def run(self) -> None:
        while not self.stop_event.is_set():
            try:
                item = self.queue_in.get(timeout=0.1)
            except Empty:
                continue
            if isinstance(item, PipelineControlMessage) and is_control_message(item, SESSION_END.kind):
                try:
                    self.on_session_end()
                except Exception as e:  # noqa: BLE001
                    logger.error("%s: error in on_session_end(): %s", self.__class__.__name__, e, exc_info=True)
                self.queue_out.put(item)
                continue
            if isinstance(item, bytes) and item == PIPELINE_END:
                break
            if isinstance(item, PipelineControlMessage):
                continue
            if not self.should_process_input(item):
                continue
            # response events share the TTS queue with their audio: forwarding them here preserves the model's exact
            # text / tool / audio order (reference baseHandler.py:139-143)
            if isinstance(item, PipelineEvent):
                self.queue_out.put(item)
                continue
            start = perf_counter()
            try:
                for output in self.process(item):
                    if not self.should_emit_output(output):
                        start = perf_counter()
                        continue
                    self._times.append(perf_counter() - start)
                    self.before_emit_output(output)
                    self.queue_out.put(self.output_for_queue(output, item))
                    start = perf_counter()
            except Exception as e:  # noqa: BLE001  (reference policy: log, drop the item, keep the thread alive)
                logger.error("%s: error in process(): %s: %s", self.__class__.__name__, type(e).__name__, e, exc_info=True)
        self.cleanup()
        self.queue_out.put(PIPELINE_END)

    @property
    def last_time(self) -> float:
        return self._times[-1]

    def cleanup(self) -> None:
        pass

    def on_session_end(self) -> None:
        pass


class BaseSTTHandler(BaseHandler):
    """Speculative-turn stale filtering around any STT (reference STT/base_stt_handler.py): inputs / outputs whose
    (turn_id, turn_revision) is no longer the tracker's latest are dropped; a Transcription marks its revision final so
    later partials of the same revision are discarded."""

    speculative_turns: Any = None
    final_revision_settle_s: float = 0.0
    _MAX_COMPLETED_FINAL_REVISIONS = 2048

    def _completed(self) -> dict:
        if not hasattr(self, "_completed_final"):
            self._completed_final: dict = {}
        return self._completed_final

    def _key(self, item: Any):
        t, r = getattr(item, "turn_id", None), getattr(item, "turn_revision", None)
        return None if t is None or r is None else (t, r)

    def _is_latest(self, item: Any, *, stability: bool, pending_reopen: bool) -> bool:
        tr = self.speculative_turns
        key = self._key(item)
        if tr is None or key is None:
            return True
        if stability and hasattr(tr, "is_latest_after_stability_window"):
            age = max(0.0, perf_counter() - getattr(item, "created_at_s", perf_counter()))
            delay = max(0.0, getattr(item, "processing_delay_s", 0.0) - age)
            return tr.is_latest_after_stability_window(key[0], key[1], max(self.final_revision_settle_s, delay))
        if pending_reopen and hasattr(tr, "is_latest_after_pending_reopen"):
            return tr.is_latest_after_pending_reopen(key[0], key[1])
        return tr.is_latest(key[0], key[1])

    def _has_queued_final_for_revision_locked(self, item: Any) -> bool:
        key = self._key(item)
        if key is None:
            return False
        return any(isinstance(q, VADAudio) and q.mode == "final" and self._key(q) == key for q in self.queue_in.queue)

    def _has_queued_final_for_revision(self, item: Any) -> bool:
        if not hasattr(self.queue_in, "mutex") or not hasattr(self.queue_in, "queue"):
            return False
        with self.queue_in.mutex:
            return self._has_queued_final_for_revision_locked(item)

    def _drop_stale_queued_inputs(self) -> int:
        """Purge queued VADAudio that can no longer matter (reference base_stt_handler.py:105-131): inputs of an already
        finalised revision, progressives with their own final queued behind them, inputs of a superseded revision."""
        if self.speculative_turns is None or not hasattr(self.queue_in, "mutex") or not hasattr(self.queue_in, "queue"):
            return 0
        dropped = 0
        with self.queue_in.mutex:
            kept = []
            while self.queue_in.queue:
                q = self.queue_in.queue.popleft()
                if isinstance(q, VADAudio) and (self._key(q) in self._completed() or (q.mode == "progressive" and self._has_queued_final_for_revision_locked(q))
                                                or not self._is_latest(q, stability=False, pending_reopen=False)):
                    dropped += 1
                else:
                    kept.append(q)
            self.queue_in.queue.extend(kept)
            if dropped:
                self.queue_in.not_full.notify_all()
        return dropped

    def should_process_input(self, item: Any) -> bool:
        key = self._key(item)
        if key is not None and key in self._completed():
            self._drop_stale_queued_inputs()
            return False
        if getattr(item, "mode", None) == "progressive" and self._has_queued_final_for_revision(item):
            return False          # its own final is already waiting: transcribing the partial would only delay it
        if not self._is_latest(item, stability=getattr(item, "mode", None) == "final", pending_reopen=True):
            self._drop_stale_queued_inputs()
            return False
        return True

    def on_session_end(self) -> None:
        if hasattr(self, "_completed_final"):
            self._completed_final.clear()

    def should_emit_output(self, output: Any) -> bool:
        key = self._key(output)
        if isinstance(output, PartialTranscription) and key is not None and key in self._completed():
            return False
        return self._is_latest(output, stability=False, pending_reopen=True)

    def before_emit_output(self, output: Any) -> None:
        key = self._key(output)
        if isinstance(output, Transcription) and key is not None:
            done = self._completed()
            done[key] = True
            while len(done) > self._MAX_COMPLETED_FINAL_REVISIONS:
                done.pop(next(iter(done)))
