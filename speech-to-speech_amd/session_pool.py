"""Per-GPU session pool: the unit of data parallelism of the hot path is one conversation session, so every GPU runs
ONE pool that owns one replica of each engine and batches the STT / TTS work of all pipeline units (handler threads)
mapped to it.  This is the new component the reference does not have (it instantiates one model per pipeline unit,
s2s_pipeline.py:474-477, 554-565); it sits entirely behind the per-unit handler API.

Threading: handler threads only enqueue requests and block on a future / chunk queue; every GPU call happens on the
pool's single worker thread (one HIP stream), so the engines need no locking.  Scheduling per loop turn:
  1. all pending final-STT requests (up to max_batch) are batched and run - keeps time-to-first-audio low;
  2. the active TTS batch advances by one streamed chunk (chunk_size codec frames);
  3. if no TTS batch is active, pending TTS requests (up to max_batch) are prefilled as a new batch.
Multi-GPU = one pool per device (``SessionPool.for_device``), sessions are pinned to a pool when their handlers are
built; there is no cross-GPU traffic.
"""

from __future__ import annotations

import logging
import queue
import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Callable, Iterator

import numpy as np

logger = logging.getLogger(__name__)

_STOP = object()


@dataclass
class _STTRequest:
    model: str                      # "whisper" | "parakeet"
    audio: np.ndarray
    options: dict
    future: Future = field(default_factory=Future)
    t_submit: float = field(default_factory=time.perf_counter)


@dataclass
class _TTSRequest:
    prompt: Any
    max_new_tokens: int
    chunks: "queue.Queue[Any]" = field(default_factory=queue.Queue)
    cancelled: threading.Event = field(default_factory=threading.Event)
    t_submit: float = field(default_factory=time.perf_counter)


class SessionPool:
    _pools: dict[str, "SessionPool"] = {}
    _pools_lock = threading.Lock()

    @classmethod
    def for_device(cls, device: str, **kw) -> "SessionPool":
        with cls._pools_lock:
            if device not in cls._pools:
                cls._pools[device] = cls(device, **kw)
            return cls._pools[device]

    @classmethod
    def shutdown_all(cls) -> None:
        with cls._pools_lock:
            pools, cls._pools = list(cls._pools.values()), {}
        for p in pools:
            p.shutdown()

    def __init__(self, device: str, max_batch: int = 64, batch_window_ms: float = 4.0, tts_chunk_size: int = 8, tts_ctx_frames: int = 25):
        self.device = device
        self.max_batch = int(max_batch)
        self.batch_window_s = batch_window_ms * 1e-3
        self.tts_chunk_size, self.tts_ctx_frames = int(tts_chunk_size), int(tts_ctx_frames)
        self._engines: dict[str, Any] = {}
        self._factories: dict[str, Callable[[], Any]] = {}
        self._stt_q: "queue.Queue[Any]" = queue.Queue()
        self._tts_q: "queue.Queue[Any]" = queue.Queue()
        self._wake = threading.Event()
        self._engine_lock = threading.Lock()
        self.stats = {"stt_batches": 0, "stt_requests": 0, "tts_batches": 0, "tts_requests": 0, "max_stt_batch": 0}
        self._thread = threading.Thread(target=self._run, name=f"s2s-pool-{device}", daemon=True)
        self._stopping = False
        self._thread.start()

    # ---- engines ---------------------------------------------------------------------------------
    def ensure_engine(self, name: str, factory: Callable[[], Any]) -> None:
        """Build (once per GPU) the engine a handler needs; called from handler.setup() on the main thread.  The build
        itself runs on the worker thread so that every GPU call of the process stays on one thread."""
        with self._engine_lock:
            if name in self._engines:
                return
            fut: Future = Future()
            self._stt_q.put(("build", name, factory, fut))
            self._wake.set()
        fut.result()

    def engine(self, name: str) -> Any:
        return self._engines[name]

    # ---- client API (handler threads) -----------------------------------------------------------
    def transcribe(self, model: str, audio: np.ndarray, **options) -> dict:
        req = _STTRequest(model, np.array(audio, dtype=np.float32, copy=True), options)
        self._stt_q.put(req)
        self._wake.set()
        return req.future.result()

    def try_transcribe(self, model: str, audio: np.ndarray, max_queue: int, **options) -> dict | None:
        """Progressive (live partial) requests are best-effort, like the reference's try-lock-and-skip
        (STT/parakeet_tdt_handler.py:264-295): when the pool is already backed up the partial is skipped."""
        if self._stt_q.qsize() >= max_queue:
            return None
        return self.transcribe(model, audio, **options)

    def synthesize(self, prompt: Any, max_new_tokens: int) -> tuple[Iterator[np.ndarray], threading.Event]:
        """-> (iterator over int16 16 kHz chunks of one utterance, cancel event)."""
        req = _TTSRequest(prompt, int(max_new_tokens))
        self._tts_q.put(req)
        self._wake.set()

        def it():
            while True:
                item = req.chunks.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item

        return it(), req.cancelled

    def shutdown(self) -> None:
        self._stopping = True
        self._stt_q.put(_STOP)
        self._wake.set()
        self._thread.join(timeout=10)
        for e in self._engines.values():
            try:
                e.close()
            except Exception:  # noqa: BLE001
                pass

    # ---- worker ---------------------------------------------------------------------------------
    def _drain(self, q: "queue.Queue[Any]", limit: int) -> list:
        items = []
        while len(items) < limit:
            try:
                items.append(q.get_nowait())
            except queue.Empty:
                break
        return items

    def _run(self) -> None:
        tts_active: tuple[Iterator, list[_TTSRequest]] | None = None
        while True:
            if tts_active is None and self._stt_q.empty() and self._tts_q.empty():
                self._wake.wait(timeout=0.05)
                self._wake.clear()
                if self._stopping and self._stt_q.empty():
                    return
                continue
            # 1. STT (and engine builds)
            if not self._stt_q.empty():
                if tts_active is None and self.batch_window_s > 0:
                    time.sleep(self.batch_window_s)      # let concurrent sessions' finals join the batch
                items = self._drain(self._stt_q, self.max_batch)
                reqs: list[_STTRequest] = []
                for it in items:
                    if it is _STOP:
                        self._stopping = True
                    elif isinstance(it, tuple) and it[0] == "build":
                        _, name, factory, fut = it
                        try:
                            self._engines[name] = factory()
                            fut.set_result(True)
                        except BaseException as e:  # noqa: BLE001
                            fut.set_exception(e)
                    else:
                        reqs.append(it)
                for model in sorted({r.model for r in reqs}):
                    self._run_stt([r for r in reqs if r.model == model], model)
            # 2./3. TTS
            if tts_active is not None:
                gen, batch = tts_active
                try:
                    chunks = next(gen)
                    for r, ch in zip(batch, chunks):
                        if ch is not None and not r.cancelled.is_set():
                            r.chunks.put(ch)
                    if all(r.cancelled.is_set() for r in batch):
                        raise StopIteration
                except StopIteration:
                    for r in batch:
                        r.chunks.put(None)
                    tts_active = None
                except BaseException as e:  # noqa: BLE001
                    logger.error("TTS batch failed: %s", e, exc_info=True)
                    for r in batch:
                        r.chunks.put(e)
                    tts_active = None
            elif not self._tts_q.empty():
                if self.batch_window_s > 0:
                    time.sleep(self.batch_window_s)
                batch = [r for r in self._drain(self._tts_q, self.max_batch) if not r.cancelled.is_set()]
                if batch:
                    self.stats["tts_batches"] += 1
                    self.stats["tts_requests"] += len(batch)
                    eng = self._engines["tts"]
                    gen = eng.stream([r.prompt for r in batch], [r.max_new_tokens for r in batch], chunk_size=self.tts_chunk_size,
                                     ctx_frames=self.tts_ctx_frames)
                    tts_active = (gen, batch)
            if self._stopping and tts_active is None and self._stt_q.empty() and self._tts_q.empty():
                return

    def _run_stt(self, reqs: list[_STTRequest], model: str) -> None:
        if not reqs:
            return
        self.stats["stt_batches"] += 1
        self.stats["stt_requests"] += len(reqs)
        self.stats["max_stt_batch"] = max(self.stats["max_stt_batch"], len(reqs))
        try:
            results = self._engines[model].transcribe_batch([r.audio for r in reqs], [r.options for r in reqs])
            for r, res in zip(reqs, results):
                r.future.set_result(res)
        except BaseException as e:  # noqa: BLE001
            logger.error("%s batch failed: %s", model, e, exc_info=True)
            for r in reqs:
                if not r.future.done():
                    r.future.set_exception(e)
