"""Per-GPU session pool: the unit of data parallelism of the hot path is one conversation session, so every GPU runs
ONE pool that owns one replica of each engine and batches the STT / TTS work of all pipeline units (handler threads)
mapped to it.  This is the new component the reference does not have (it instantiates one model per pipeline unit,
s2s_pipeline.py:474-477, 554-565); it sits entirely behind the per-unit handler API.

Threading: handler threads only enqueue requests and block on a future / chunk queue.  GPU work runs on the pool's
own threads, each engine instance is only ever touched by one of them, so the engines need no locking:
  * the STT worker batches all pending final-STT requests (up to max_batch) and runs them on its stream;
  * ``tts_lanes`` TTS lanes (thread + HIP stream + engine instance each; lanes > 0 are replicas that share lane 0's
    weights and own only their KV caches / workspace) each take a batch of pending TTS requests (up to max_batch),
    prefill it and stream it chunk by chunk (chunk_size codec frames) to the requesting handlers.
The autoregressive TTS kernels are latency-bound and leave most of the chip idle, so the lanes' turn-batches and the
STT worker's encoder passes overlap on the device (measured in bench.py).
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
    live: bool = False              # a live partial (best effort): served behind every queued final, in the rows a batch has left


@dataclass
class _TTSRequest:
    prompt: Any
    max_new_tokens: int
    min_frames: int = 0             # eos is masked until the utterance has this many frames (throughput runs: SURVEY 8d forces full replies)
    chunks: "queue.Queue[Any]" = field(default_factory=queue.Queue)
    cancelled: threading.Event = field(default_factory=threading.Event)
    t_submit: float = field(default_factory=time.perf_counter)
    # set by the consumer once it has handed out its first AUDIBLE block (the handler trims leading silence): until then the lane does not
    # pace this session by its audio lead - a reply that starts with silent chunks would otherwise reach its first audio at playback speed
    started: threading.Event = field(default_factory=threading.Event)


class _ChunkStream:
    """What `synthesize` hands to a handler thread: an iterator over int16 chunks; `started` is the request's event (see _TTSRequest)."""

    def __init__(self, req: "_TTSRequest"):
        self._req = req
        self.started = req.started

    def __iter__(self):
        return self

    def __next__(self):
        item = self._req.chunks.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item


def release_pool_on_failure(setup):
    """Decorator for a handler's setup(): a setup that raises after it attached itself to a pool (a bad checkpoint, an unknown speaker) lets
    go of the pool again - otherwise the stale attachment keeps the device's pool (and its engines) alive for ever."""
    import functools

    @functools.wraps(setup)
    def wrapper(self, *a, **kw):
        try:
            return setup(self, *a, **kw)
        except BaseException:
            pool = getattr(self, "pool", None)
            if pool is not None and hasattr(pool, "detach"):
                self.pool = None
                pool.detach(self)
            raise

    return wrapper


class SessionPool:
    _pools: dict[str, "SessionPool"] = {}
    _pools_lock = threading.Lock()

    @classmethod
    def for_device(cls, device: str, **kw) -> "SessionPool":
        with cls._pools_lock:
            if device not in cls._pools:
                cls._pools[device] = cls(device, **kw)
            else:
                pool = cls._pools[device]
                if "tts_lanes" in kw:        # the pool may have been created by the STT handler: grow to the TTS handler's lane count
                    pool._ensure_lanes(int(kw["tts_lanes"]))
                for k, v in kw.items():      # one pool per GPU: the first handler's settings stand; say so instead of silently ignoring
                    cur = {"max_batch": pool.max_batch, "batch_window_ms": pool.batch_window_s * 1e3, "tts_chunk_size": pool.tts_chunk_size,
                           "tts_ctx_frames": pool.tts_ctx_frames}.get(k)
                    if cur is not None and abs(float(cur) - float(v)) > 1e-9:
                        logger.warning("SessionPool(%s) already exists with %s=%s; ignoring %s=%s from a later handler", device, k, cur, k, v)
            return cls._pools[device]

    def attach(self, user: Any) -> None:
        """EVERY user of a pool registers itself: the handlers (whether they obtained the pool through `for_device` or were handed it as
        `pool=`) and direct `for_device` callers such as the server (any object serves as the token).  `detach` from the user's cleanup()
        releases the engines (weights, KV arenas, graphs) once the LAST user of a `for_device` pool is gone - the pool-level equivalent of the
        reference handlers' `del self.model` (STT/parakeet_tdt_handler.py:636-640, TTS/qwen3_tts_handler.py:1011-1034).  A pool somebody
        constructed directly (`SessionPool(...)`: tests, bench.py) belongs to that caller and is never shut down from here."""
        with self._pools_lock:
            self.__dict__.setdefault("_attached", set()).add(id(user))

    def detach(self, user: Any) -> bool:
        """-> True when this was the last attached user of a `for_device` pool and the pool has been shut down (engines closed, device memory released)."""
        with self._pools_lock:
            att = self.__dict__.setdefault("_attached", set())
            if id(user) not in att:
                return False
            att.discard(id(user))
            last = not att and self._pools.get(self.device) is self
            if last:
                del self._pools[self.device]
        if last:
            self.shutdown()
        return last

    def configure(self, *, tts_lead_s: float | None = None, tts_admit_wait_s: float | None = None, stt_max_batch: int | None = None) -> None:
        """Scheduling knobs a handler may set after the pool exists (the lanes read them every round).  One pool per GPU: a later handler
        asking for something else is told so, the first explicit setting stands."""
        for name, val, unset in (("tts_lead_s", tts_lead_s, float("inf")), ("tts_admit_wait_s", tts_admit_wait_s, 0.0), ("stt_max_batch", stt_max_batch, 0)):
            if val is None:
                continue
            val = type(unset)(val)
            cur = getattr(self, name)
            if cur == unset or cur == val:
                setattr(self, name, val)
            else:
                logger.warning("SessionPool(%s) already runs with %s=%s; ignoring %s=%s from a later handler", self.device, name, cur, name, val)

    @classmethod
    def shutdown_all(cls) -> None:
        with cls._pools_lock:
            pools, cls._pools = list(cls._pools.values()), {}
        for p in pools:
            p.shutdown()

    def __init__(self, device: str, max_batch: int = 64, batch_window_ms: float = 4.0, tts_chunk_size: int = 8, tts_ctx_frames: int = 25,
                 tts_lanes: int = 3, continuous: bool = True, stt_priority: int = 0, stt_workers: int = 1, tts_lead_s: float = float("inf"), tts_fill_first: int = 0, tts_lane_cus: int = 8, stt_max_batch: int = 0,
                 tts_admit_wait_s: float = 0.0, stt_stages: int = 1, tts_nursery_slots: int = 0, tts_nursery_priority: int = 0, stt_slots: int = 0,
                 stt_slot_steps: int = 4):
        self.device = device
        # > 0: engines that offer it (Whisper: slots_step, stt_continuous.py) run their token loop continuously - requests join after their
        # encoder pass (second stream) and leave at their own eos instead of being served batch by batch.  Written at the end of round 4
        # without GPU time left to measure it: off by default.
        self.stt_slots = max(0, int(stt_slots))
        self.stt_slot_steps = max(1, int(stt_slot_steps))
        self._stt_sq: dict[str, "queue.Queue[Any]"] = {}
        self._stt_sthreads: dict[str, threading.Thread] = {}
        self.stt_priority = int(stt_priority)    # HIP stream priority of the STT worker (-1 = high: STT sits in front of every session's first audio)
        # optional pacing of the continuous lanes: a round of continuation chunks runs only when some session's delivered audio leads its
        # playback by less than this (a round with a NEW cohort - time-to-first-audio - always runs at once).  Default inf = eager rounds:
        # measured on C3 (profiles/r03_open_loop_ab.txt) pacing at 1-2 s raises the rows per round but does NOT raise the STT service rate,
        # which is what bounds the sustained session count - p95 TTFA at 900 sessions 958 ms eager vs 3.3 s paced at 2 s.
        self.tts_lead_s = float(tts_lead_s)
        # ... and a queued request joins the loop once it has waited this long (at once when the lane is idle or a round is due anyway):
        # bounds what pacing adds to a session's time-to-first-audio
        self.tts_admit_wait_s = float(tts_admit_wait_s)
        # STT as a two-stage pipeline (engines with stt_begin / stt_finish / replica, i.e. Whisper): the worker thread runs log-mel + encoder
        # of batch k + 1 on its stream while a second thread runs the token loop of batch k on another stream, each batch on its own engine
        # replica (weights shared).  The encoder fills the chip for 5 ms per session, the token loop is a chain of 35 x 256 latency-bound
        # launches (72 ms whatever the batch).  MEASURED NEGATIVE on C3 (profiles/r04_open_loop_ab.txt): with the wide TTS lane on a third
        # stream the token loop's 9 000 dependent launches wait behind the encoder's and the frame loop's workgroups - p95 TTFA at 1 100
        # sessions 1 085 ms with two stages vs 848 ms with one.  Default 1 = one stage; the option stays for lighter loads.
        self.stt_stages = max(1, int(stt_stages))
        # > 0: every session's FIRST chunk runs in a small, eager frame loop of this many slots (an engine replica on its own thread and
        # stream); the session is then moved (s2s_tts_migrate) into the wide loop, whose rounds are paced by audio lead alone - so the wide
        # loop's pace no longer sits in anybody's time-to-first-audio.  0: one loop does both (admission wait `tts_admit_wait_s`).
        self.tts_nursery_slots = max(0, int(tts_nursery_slots))
        self.tts_nursery_priority = int(tts_nursery_priority)      # HIP stream priority of the nursery loop (-1 = high: it sits in every session's first audio)
        self.stt_workers = max(1, int(stt_workers))
        self._stt_extra_reps: dict[str, dict[int, Any]] = {}      # model -> {extra worker k: its replica, built with the engine}
        self._stt_reps: dict[str, list] = {}          # model -> engines of the pipeline (the pool's engine + replicas)
        self._stt_free: dict[str, "queue.Queue[int]"] = {}
        self._stt_dq: "queue.Queue[Any]" = queue.Queue()
        self._stt_dthread: threading.Thread | None = None
        # lane admission policy: 0 = the least loaded lanes take new sessions (spreads them over all lanes); T > 0 = fill first: lane k only
        # takes new sessions once every lane below it holds at least T live sessions (fewer frame loops running, fuller rows)
        self.tts_fill_first = int(tts_fill_first)
        # largest final-STT batch (0: the engine's max_batch).  A cap keeps a backlog from locking the pool into long STT batches whose
        # sessions then all start their TTS at the same instant (bursty prefills against the next batch's encoder)
        self.stt_max_batch = int(stt_max_batch)
        self.tts_lane_cus = 8                    # (round 3's CU-masked lane streams measured negative and were removed; the argument is accepted and ignored)
        self.continuous = bool(continuous)       # TTS lanes admit / retire sessions at chunk boundaries (tts_continuous.py); False: turn-batches
        self._lane_live: dict[int, int] = {}     # live sessions per continuous lane (load balancing between lanes)
        self.tts_lanes = max(1, int(tts_lanes))
        self.max_batch = int(max_batch)
        self.batch_window_s = batch_window_ms * 1e-3
        self.tts_chunk_size, self.tts_ctx_frames = int(tts_chunk_size), int(tts_ctx_frames)
        self._engines: dict[str, Any] = {}
        self._lane_replicas: list[Any] = []
        self._factories: dict[str, Callable[[], Any]] = {}
        # STT queues.  _stt_q: final transcriptions, shared by all STT workers.  _stt_pq: live partials, shared; a batch takes them only into the
        # rows the queued finals leave free (finals sit in front of a session's first audio, partials are best effort).  _stt_q0: what ONLY the
        # first worker may serve - engine builds, host calls, the stop token and, with several workers, requests for engines that cannot be
        # replicated or that the slot loop serves (the extra workers used to take such items and append them again, which reordered
        # the FIFO and spun a core)
        self._stt_q: "queue.Queue[Any]" = queue.Queue()
        self._stt_pq: "queue.Queue[Any]" = queue.Queue()
        self._stt_q0: "queue.Queue[Any]" = queue.Queue()
        self._tts_q: "queue.Queue[Any]" = queue.Queue()
        self._wake = threading.Event()
        self._engine_lock = threading.Lock()
        self._building: dict[str, Future] = {}
        self.stats = {"stt_batches": 0, "stt_requests": 0, "tts_batches": 0, "tts_requests": 0, "max_stt_batch": 0, "tts_lanes_used": 0,
                      "stt_busy_s": 0.0, "tts_rounds": 0, "tts_round_s": 0.0, "tts_admit_s": 0.0, "tts_row_rounds": {}}
        self._stats_lock = threading.Lock()
        self._thread = threading.Thread(target=self._run, name=f"s2s-pool-{device}", daemon=True)
        self._lanes: list[threading.Thread] = []
        self._lane_ready: list[threading.Event] = []
        self._warm_requested = False
        self._stopping = False
        self._thread.start()
        # extra STT workers (thread + HIP stream + engine replica each): final-STT batches of different workers overlap on the device, which
        # halves the queueing in front of every session's first audio when one worker is saturated (bench.py open-loop probe)
        self.stt_workers = max(1, int(stt_workers))
        self._stt_extra = [threading.Thread(target=self._run_stt_extra, args=(k,), name=f"s2s-pool-{device}-stt{k}", daemon=True) for k in range(1, self.stt_workers)]
        for t in self._stt_extra:
            t.start()
        self._ensure_lanes(self.tts_lanes)

    def ensure_stt_workers(self, n: int) -> None:
        """Grow the pool to `n` STT workers (a handler that knows its load asks for them after the pool exists: live Parakeet partials are
        launch-bound passes, three workers overlap them - profiles/r06_open_loop_ab.txt).  Never shrinks; replicas of engines that already
        exist are built here, not in serving time."""
        with self._pools_lock:
            old, n = self.stt_workers, max(1, int(n))
            if n <= old:
                return
            self.stt_workers = n
            new = [threading.Thread(target=self._run_stt_extra, args=(k,), name=f"s2s-pool-{self.device}-stt{k}", daemon=True) for k in range(old, n)]
            self._stt_extra.extend(new)
        for model in list(self._engines):
            self._stt_extra_build(model)
        for t in new:
            t.start()

    def _ensure_lanes(self, n: int) -> None:
        while len(self._lanes) < max(1, n):
            i = len(self._lanes)
            t = threading.Thread(target=self._run_tts_lane, args=(i,), name=f"s2s-pool-{self.device}-tts{i}", daemon=True)
            self._lane_live.setdefault(i, 0)         # (every lane's key exists before any lane iterates the dict)
            self._lane_ready.append(threading.Event())
            self._lanes.append(t)
            t.start()
        self.tts_lanes = len(self._lanes)

    # ---- engines ---------------------------------------------------------------------------------
    def ensure_engine(self, name: str, factory: Callable[[], Any]) -> None:
        """Build (once per GPU) the engine a handler needs; called from handler.setup() on the main thread.  The build
        itself runs on the worker thread so that every GPU call of the process stays on one thread."""
        with self._engine_lock:
            if name in self._engines:
                return
            fut = self._building.get(name)            # a second caller waits for the build already in flight instead of starting another
            if fut is None:
                fut = self._building[name] = Future()
                self._stt_q0.put(("build", name, factory, fut))
                self._wake.set()
        try:
            fut.result()
        finally:
            with self._engine_lock:
                self._building.pop(name, None)

    def engine(self, name: str) -> Any:
        return self._engines[name]

    def warm_once(self, name: str, fn: Callable[[], Any]) -> bool:
        """Run a handler's warm-up pass for engine `name` unless another handler of this pool already did: the model is shared by every
        pipeline unit of the GPU, so it is warmed once - the reference warms once per handler because every handler owns its model
        (backend_registry.py:206-264).  -> True when `fn` ran."""
        with self._engine_lock:
            done = self.__dict__.setdefault("_warmed", set())
            if name in done:
                return False
            done.add(name)
        try:
            fn()
        except BaseException:
            with self._engine_lock:
                done.discard(name)
            raise
        return True

    def warm_tts_lanes(self, timeout: float = 300.0) -> None:
        """Build every lane's engine (replicas share lane 0's weights) and run its warm-up utterance now, so that the first
        real request does not pay for it - the reference's handlers load and warm up synchronously in setup()
        (backend_registry.py:206-264; TTS/qwen3_tts_handler.py setup -> model.warmup)."""
        self._warm_requested = True
        for ev in list(self._lane_ready):
            ev.wait(timeout)

    def warm_stt(self, model: str, timeout: float = 300.0) -> None:
        """Build the replicas of the extra STT workers and of the two-stage STT pipeline now, not on the first request (an engine build allocates,
        pins host memory and synchronises the device: inside serving time that is a late audio block for somebody)."""
        self._stt_extra_build(model)
        if self.stt_stages <= 1:
            return
        fut: Future = Future()
        self._stt_q0.put(("call", lambda: self._stt_pipeline_build(model), fut))
        self._wake.set()
        fut.result(timeout)

    # ---- client API (handler threads) -----------------------------------------------------------
    def _stt_route(self, req: _STTRequest) -> "queue.Queue[Any]":
        if self.stt_workers > 1:
            eng = self._engines.get(req.model)
            if not hasattr(eng, "replica") or (self.stt_slots > 0 and hasattr(eng, "slots_step")):
                return self._stt_q0
        return self._stt_pq if req.live else self._stt_q

    def stt_backlog(self) -> int:
        """Transcription requests waiting for a batch (finals + live partials)."""
        return self._stt_q.qsize() + self._stt_pq.qsize() + self._stt_q0.qsize()

    def transcribe(self, model: str, audio: np.ndarray, live: bool = False, **options) -> dict:
        """Blocks until the transcript of `audio` is there.  live=True marks a live partial: same computation, lower priority."""
        if self._stopping:
            raise RuntimeError("SessionPool is shut down")
        req = _STTRequest(model, np.array(audio, dtype=np.float32, copy=True), options, live=bool(live))
        self._stt_route(req).put(req)
        self._wake.set()
        return req.future.result()

    def try_transcribe(self, model: str, audio: np.ndarray, max_queue: int, **options) -> dict | None:
        """Load shedding by POOL backlog (round 2 - 5 default for live partials; off since round 6: the reference skips a partial only while
        the SAME unit's compute lock is held, STT/parakeet_tdt_handler.py:264-295 - see handlers/parakeet.py).  Kept for callers that want it."""
        if self.stt_backlog() >= max_queue:
            return None
        return self.transcribe(model, audio, live=True, **options)

    def synthesize(self, prompt: Any, max_new_tokens: int, min_frames: int = 0) -> tuple[Iterator[np.ndarray], threading.Event]:
        """-> (iterator over int16 16 kHz chunks of one utterance, cancel event).  min_frames > 0 masks eos until the utterance
        has that many codec frames (benchmarks that need replies of a fixed length; 0 on the serving path)."""
        if self._stopping:
            raise RuntimeError("SessionPool is shut down")
        req = _TTSRequest(prompt, int(max_new_tokens), int(min_frames))
        self._tts_q.put(req)
        self._wake.set()

        return _ChunkStream(req), req.cancelled

    def _fail_pending(self) -> None:
        """Requests still queued at shutdown are failed, so no handler thread blocks forever on a future / chunk queue."""
        err = RuntimeError("SessionPool is shut down")
        for q in (self._stt_q0, self._stt_q, self._stt_pq, self._tts_q):
            while True:
                try:
                    it = q.get_nowait()
                except queue.Empty:
                    break
                if isinstance(it, _STTRequest) and not it.future.done():
                    it.future.set_exception(err)
                elif isinstance(it, _TTSRequest):
                    it.chunks.put(err)
                elif isinstance(it, tuple) and it[0] == "build" and not it[3].done():
                    it[3].set_exception(err)
                elif isinstance(it, tuple) and it[0] == "call" and not it[2].done():
                    it[2].set_exception(err)

    def shutdown(self, close_engines: bool = True) -> None:
        """close_engines=False leaves the engines registered with the pool open (they belong to the caller); lane replicas the pool
        built itself are always closed."""
        self._stopping = True
        self._fail_pending()
        self._stt_q0.put(_STOP)
        for _ in self._lanes:
            self._tts_q.put(_STOP)
        self._wake.set()
        self._thread.join(timeout=10)
        for t in self._stt_extra + self._lanes + ([self._stt_dthread] if self._stt_dthread is not None else []) + list(self._stt_sthreads.values()):
            t.join(timeout=10)
        for sq in self._stt_sq.values():
            for it in self._drain(sq, 1 << 20):
                if isinstance(it, _STTRequest) and not it.future.done():
                    it.future.set_exception(RuntimeError("SessionPool is shut down"))
        while True:
            try:
                item = self._stt_dq.get_nowait()
            except queue.Empty:
                break
            for r in item[2]:
                if not r.future.done():
                    r.future.set_exception(RuntimeError("SessionPool is shut down"))
        self._fail_pending()
        for e in (list(self._engines.values()) if close_engines else []) + self._lane_replicas:
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
        """STT worker (also builds the engines, so that model loading stays on one thread).  Like the TTS lanes it issues all its GPU
        work - including the host-to-device copies of the utterances - on its own HIP stream: anything on the legacy default stream
        would collide with (and poison) a lane's stream capture."""
        ctx = self._lane_stream_ctx(self.stt_priority)
        ctx.__enter__()
        while True:
            self._wake.clear()
            first = self._stt_take(own=True)
            if first is None:
                if self._stopping:
                    return
                self._wake.wait(0.05)
                continue
            if self.batch_window_s > 0 and isinstance(first, _STTRequest):
                time.sleep(self.batch_window_s)      # let concurrent sessions' finals join the batch
            cap = self.max_batch
            if isinstance(first, _STTRequest):
                cap = min(cap, int(getattr(self._engines.get(first.model), "max_batch", cap)))     # never hand an engine more rows than it was built for
                if self.stt_max_batch > 0:
                    cap = min(cap, self.stt_max_batch)
            items = self._stt_fill([first], cap, own=True)
            reqs: list[_STTRequest] = []
            for it in items:
                if it is _STOP:
                    self._stopping = True
                elif isinstance(it, tuple) and it[0] == "call":      # host-side set-up that must run on this thread (warm_stt)
                    _, fn, fut = it
                    try:
                        fn()
                        self._sync_current_stream()
                        fut.set_result(True)
                    except BaseException as e:  # noqa: BLE001
                        fut.set_exception(e)
                elif isinstance(it, tuple) and it[0] == "build":
                    _, name, factory, fut = it
                    try:
                        self._engines[name] = factory()
                        self._stt_pipeline_build(name)   # (replicas are built HERE, with the engine - never while serving)
                        self._stt_extra_build(name)
                        self._sync_current_stream()      # uploads / conversions of the build are complete before any lane uses the weights
                        fut.set_result(True)
                    except BaseException as e:  # noqa: BLE001
                        fut.set_exception(e)
                else:
                    reqs.append(it)
            for model in sorted({r.model for r in reqs}):
                mine = [r for r in reqs if r.model == model]
                if self.stt_slots > 0 and hasattr(self._engines.get(model), "slots_step"):
                    self._stt_slots_submit(mine, model)
                    continue
                if self.stt_stages > 1 and model not in self._stt_reps:
                    self._stt_pipeline_build(model)      # (an engine registered directly, e.g. by bench.py: first use)
                if self._stt_reps.get(model):
                    self._run_stt_staged(mine, model)
                else:
                    self._run_stt(mine, model)
            if self._stopping and self.stt_backlog() == 0:
                return

    def _stt_take(self, own: bool) -> Any:
        """Next STT item without blocking: the first worker's own queue (own=True), then finals, then live partials."""
        for q in ((self._stt_q0,) if own else ()) + (self._stt_q, self._stt_pq):
            try:
                return q.get_nowait()
            except queue.Empty:
                pass
        return None

    def _stt_fill(self, items: list, cap: int, own: bool) -> list:
        """Fill a batch that starts with `items` up to `cap` entries: finals before live partials (a partial never takes a final's row)."""
        for q in ((self._stt_q0,) if own else ()) + (self._stt_q, self._stt_pq):
            if len(items) < cap:
                items += self._drain(q, cap - len(items))
        return items

    # ---- continuous token loop (slots) ------------------------------------------------------------
    def _stt_slots_submit(self, reqs: list[_STTRequest], model: str) -> None:
        if model not in self._stt_sq:
            self._stt_sq[model] = queue.Queue()
            t = threading.Thread(target=self._run_stt_slots, args=(model,), name=f"s2s-pool-{self.device}-sttloop-{model}", daemon=True)
            self._stt_sthreads[model] = t
            t.start()
        for r in reqs:
            self._stt_sq[model].put(r)

    def _run_stt_slots(self, model: str) -> None:
        """The running token loop of one STT engine (own thread, its stream for the loop + a side stream for the encoder passes)."""
        from .stt_continuous import ContinuousSTT

        ctx = self._lane_stream_ctx(self.stt_priority)
        ctx.__enter__()
        eng = self._engines[model]
        hooks: dict[str, Any] = {}
        try:
            import torch

            dev = torch.device(self.device)
            if dev.type == "cuda" and torch.cuda.is_available():
                side = torch.cuda.Stream(device=dev, priority=self.stt_priority)

                def make_event():
                    ev = torch.cuda.Event()
                    ev.record()
                    return ev

                hooks = {"enc_stream": lambda: torch.cuda.stream(side), "make_event": make_event,
                         "wait_event": lambda ev: torch.cuda.current_stream().wait_event(ev)}
        except Exception:  # noqa: BLE001
            hooks = {}
        cs = ContinuousSTT(eng, steps_per_round=self.stt_slot_steps, max_group=self.stt_max_batch or 16, **hooks)
        q = self._stt_sq[model]
        backlog: list[_STTRequest] = []
        t_busy = None
        while True:
            try:
                if cs.idle and not backlog:
                    if t_busy is not None:
                        with self._stats_lock:
                            self.stats["stt_busy_s"] += time.perf_counter() - t_busy
                        t_busy = None
                    try:
                        backlog.append(q.get(timeout=0.05))
                    except queue.Empty:
                        if self._stopping:
                            return
                        continue
                    if self.batch_window_s > 0:
                        time.sleep(self.batch_window_s)
                    t_busy = time.perf_counter()
                backlog += self._drain(q, 1 << 20)
                if self._stopping:
                    raise RuntimeError("SessionPool is shut down")
                if backlog and cs.can_encode():
                    take, rest = backlog[: cs.max_group], backlog[cs.max_group :]
                    left = cs.encode(take)
                    for j, err in cs.rejected:
                        if not j.future.done():
                            j.future.set_exception(err)
                    n_in = len(take) - len(left) - len(cs.rejected)
                    if n_in > 0:
                        with self._stats_lock:
                            self.stats["stt_batches"] += 1
                            self.stats["stt_requests"] += n_in
                            self.stats["max_stt_batch"] = max(self.stats["max_stt_batch"], n_in)
                    backlog = left + rest
                for job, res in cs.step():
                    job.future.set_result(res)
            except BaseException as e:  # noqa: BLE001
                if not self._stopping:
                    logger.error("%s token loop failed: %s", model, e, exc_info=True)
                for j in cs.abort() + backlog:
                    if not j.future.done():
                        j.future.set_exception(e)
                backlog = []
                if self._stopping:
                    return

    # ---- two-stage STT ---------------------------------------------------------------------------
    def _stt_pipeline_build(self, model: str) -> None:
        eng = self._engines.get(model)
        if self.stt_stages <= 1 or model in self._stt_reps or eng is None:
            return
        if not all(hasattr(eng, a) for a in ("stt_begin", "stt_finish", "replica")):
            self._stt_reps[model] = []
            return
        cap = min(self.max_batch, int(getattr(eng, "max_batch", self.max_batch)))
        if self.stt_max_batch > 0:
            cap = min(cap, self.stt_max_batch)
        reps = [eng]
        try:
            for _ in range(self.stt_stages - 1):
                try:
                    rep = eng.replica(max_batch=cap)
                except TypeError:
                    rep = eng.replica()
                reps.append(rep)
                self._lane_replicas.append(rep)
        except BaseException as e:  # noqa: BLE001
            logger.error("STT pipeline: could not build the %s replicas (%s); one stage", model, e, exc_info=True)
            self._stt_reps[model] = []
            return
        self._stt_reps[model] = reps
        self._stt_free[model] = queue.Queue()
        for i in range(len(reps)):
            self._stt_free[model].put(i)
        if self._stt_dthread is None:
            self._stt_dthread = threading.Thread(target=self._run_stt_decode, name=f"s2s-pool-{self.device}-sttdec", daemon=True)
            self._stt_dthread.start()

    def _run_stt_staged(self, reqs: list[_STTRequest], model: str) -> None:
        """Stage 1 (this thread, its stream): features + encoder of the batch on a free replica, then hand it to the decode thread."""
        if not reqs:
            return
        with self._stats_lock:
            self.stats["stt_batches"] += 1
            self.stats["stt_requests"] += len(reqs)
            self.stats["max_stt_batch"] = max(self.stats["max_stt_batch"], len(reqs))
        idx = None
        try:
            while idx is None:
                try:
                    idx = self._stt_free[model].get(timeout=0.05)
                except queue.Empty:
                    if self._stopping:
                        raise RuntimeError("SessionPool is shut down")
            t_b = time.perf_counter()
            eng = self._stt_reps[model][idx]
            state = eng.stt_begin([r.audio for r in reqs], [r.options for r in reqs])
            ev = None
            try:
                import torch

                if torch.cuda.is_available():
                    ev = torch.cuda.Event()
                    ev.record()
            except Exception:  # noqa: BLE001
                ev = None
            self._stt_dq.put((model, idx, reqs, state, ev, t_b))
        except BaseException as e:  # noqa: BLE001
            logger.error("%s encoder stage failed: %s", model, e, exc_info=True)
            if idx is not None:
                self._stt_free[model].put(idx)
            for r in reqs:
                if not r.future.done():
                    r.future.set_exception(e)

    def _run_stt_decode(self) -> None:
        """Stage 2 (own thread + stream): the token loop of every encoded batch, in order."""
        ctx = self._lane_stream_ctx(self.stt_priority)
        ctx.__enter__()
        while True:
            try:
                item = self._stt_dq.get(timeout=0.05)
            except queue.Empty:
                if self._stopping:
                    return
                continue
            model, idx, reqs, state, ev, t_b = item
            try:
                if ev is not None:
                    import torch

                    torch.cuda.current_stream().wait_event(ev)
                results = self._stt_reps[model][idx].stt_finish(state)
                with self._stats_lock:
                    self.stats["stt_busy_s"] += time.perf_counter() - t_b
                for r, res in zip(reqs, results):
                    r.future.set_result(res)
            except BaseException as e:  # noqa: BLE001
                logger.error("%s decode stage failed: %s", model, e, exc_info=True)
                for r in reqs:
                    if not r.future.done():
                        r.future.set_exception(e)
            finally:
                self._stt_free[model].put(idx)

    def _stt_extra_build(self, model: str) -> None:
        """Replicas of the extra STT workers are built with the engine (replica finalize allocates and uses the legacy stream; the
        first features() call pins memory - neither belongs into serving time).  Engines registered directly are replicated lazily."""
        eng = self._engines.get(model)
        if self.stt_workers <= 1 or eng is None or not hasattr(eng, "replica") or not hasattr(eng, "transcribe_batch"):
            return
        reps = self._stt_extra_reps.setdefault(model, {})
        for k in range(1, self.stt_workers):
            if k not in reps:
                try:
                    reps[k] = eng.replica()
                    self._lane_replicas.append(reps[k])
                except BaseException as e:  # noqa: BLE001
                    logger.error("STT worker %d: could not build its %s replica with the engine (%s); it will try on first use", k, model, e)

    def _run_stt_extra(self, k: int) -> None:
        """STT worker k > 0: same batching as the first worker, on replicas that share the engines' weights.  Engine builds, and requests for
        engines that cannot be replicated, stay with the first worker."""
        ctx = self._lane_stream_ctx(self.stt_priority)
        ctx.__enter__()
        replicas: dict[str, Any] = {}
        while not self._stopping:
            first = self._stt_take(own=False)        # (only replicable, batch-served models reach the shared queues: _stt_route)
            if first is None:
                time.sleep(0.005)
                continue
            if self.batch_window_s > 0:
                time.sleep(self.batch_window_s)
            model = first.model
            try:
                if model not in replicas:
                    pre = self._stt_extra_reps.get(model, {}).get(k)
                    if pre is not None:
                        replicas[model] = pre
                    else:
                        replicas[model] = self._engines[model].replica()
                        self._lane_replicas.append(replicas[model])
            except BaseException as e:  # noqa: BLE001
                logger.error("STT worker %d could not build its %s replica: %s", k, model, e, exc_info=True)
                self._stt_q0.put(first)              # the first worker serves it on the engine itself
                self._wake.set()
                return
            eng = replicas[model]
            reqs, back = [first], []
            cap = min(self.max_batch, int(getattr(eng, "max_batch", self.max_batch)))
            if self.stt_max_batch > 0:
                cap = min(cap, self.stt_max_batch)
            for it in self._stt_fill([], cap - 1, own=False):
                (reqs if it.model == model else back).append(it)
            for it in back:                          # (another model's request: rare - two STT models on one pool)
                self._stt_route(it).put(it)
            self._run_stt(reqs, model, eng)

    @staticmethod
    def _drop_cancelled(reqs: list) -> list:
        """Requests cancelled while they waited leave the backlog WITH a terminal item on their chunk queue: a consumer that cancels and
        keeps iterating must see the end of its stream, not block."""
        keep = []
        for r in reqs:
            if r.cancelled.is_set():
                r.chunks.put(None)
            else:
                keep.append(r)
        return keep

    def _lane_engine(self, lane: int) -> Any:
        """Lane 0 streams on the pool's TTS engine; further lanes on replicas that share its weights (built on first use,
        on the lane's own thread).  An engine without ``replica()`` (a test double) is served by lane 0 only."""
        eng = self._engines["tts"]
        if lane == 0:
            return eng
        if not hasattr(eng, "replica"):
            return None
        rep = eng.replica()
        self._lane_replicas.append(rep)
        return rep

    def _lane_stream_ctx(self, priority: int = 0, cus_per_8: int = 8):
        """Every lane issues its GPU work on its own HIP stream (no-op for devices torch does not know, e.g. test doubles)."""
        import contextlib

        try:
            import torch

            dev = torch.device(self.device)
            if dev.type == "cuda" and torch.cuda.is_available():
                torch.cuda.set_device(dev)
                # (no device-wide synchronise here: it would collide with another lane's stream capture; the worker that builds an
                # engine synchronises ITS stream before the build is reported done)
                return torch.cuda.stream(torch.cuda.Stream(device=dev, priority=priority))
        except Exception:  # noqa: BLE001
            pass
        return contextlib.nullcontext()

    def _sync_current_stream(self) -> None:
        try:
            import torch

            dev = torch.device(self.device)
            if dev.type == "cuda" and torch.cuda.is_available():
                torch.cuda.current_stream(dev).synchronize()
        except Exception:  # noqa: BLE001
            pass

    def _lane_start(self, lane: int, warm: bool):
        """-> (engine, entered stream context) of a lane, or (None, None) when the TTS engine cannot be replicated."""
        try:
            eng = self._lane_engine(lane)
        except BaseException as e:  # noqa: BLE001
            logger.error("TTS lane %d could not build its engine replica: %s", lane, e, exc_info=True)
            eng = None
        if eng is None:
            return None, None
        ctx = self._lane_stream_ctx(0, self.tts_lane_cus)
        ctx.__enter__()
        with self._stats_lock:
            self.stats["tts_lanes_used"] += 1
        if warm and hasattr(eng, "warmup"):
            try:
                eng.warmup()
            except BaseException as e:  # noqa: BLE001
                logger.error("TTS lane %d warm-up failed: %s", lane, e, exc_info=True)
        return eng, ctx

    def _run_tts_lane_continuous(self, lane: int, eng: Any, first: Any, ready: threading.Event | None = None) -> None:
        """One lane as a continuously batched engine: at every chunk boundary finished sessions leave, queued requests are prefilled into
        free slots, and the frame loop goes on for everybody (speech-to-speech_amd/tts_continuous.py).  Several lanes share the request
        queue; a busy lane only takes new requests while it is among the least loaded ones, idle lanes block on the queue.

        Pacing (round 4): the frame loop's cost per round grows far slower than its row count (csrc/gemm_decw.hip), so rounds should be
        FULL rather than frequent.  A round runs when (a) some session's delivered audio leads its playback by less than ``tts_lead_s``,
        or (b) a queued request has waited ``tts_admit_wait_s`` (or the lane is idle: a lone session starts at once).  With both at their
        defaults (inf / 0) every round runs as soon as the previous one has been handed out (eager)."""
        from .tts_continuous import ROW_BUCKETS, ContinuousTTS

        if self.tts_nursery_slots > 0 and hasattr(eng, "migrate_from") and hasattr(eng, "replica"):
            return self._run_tts_lane_wide(lane, eng, first, ready)
        backlog: list = [first] if first is not None else []
        cb = None
        try:
            cb = ContinuousTTS(eng, chunk_size=self.tts_chunk_size)
            if hasattr(eng, "prepare_admission"):
                eng.prepare_admission()              # pinned staging of the prompt rows: allocated now, never while serving
                cb.prepare()
            for r in [r for r in ROW_BUCKETS if r <= cb.cap] or [cb.cap]:
                eng.step(r, 1)                       # captures the frame graph of every row count while all slots are idle
            self._sync_current_stream()
        except BaseException as e:  # noqa: BLE001  (a lane that cannot start must not leave its requests - or warm_tts_lanes - hanging)
            logger.error("TTS lane %d could not start: %s", lane, e, exc_info=True)
            for job in backlog:
                job.chunks.put(e)
            self._lane_live[lane] = 1 << 30          # never the least loaded lane again
            return
        finally:
            if ready is not None:
                ready.set()
        stop = False
        while True:
            try:
                if self._stopping:       # shutdown: live sessions are failed NOW (their engine may be handed to another pool right after)
                    err = RuntimeError("SessionPool is shut down")
                    for job in cb.abort() + backlog:
                        job.chunks.put(err)
                    self._sync_current_stream()
                    return
                self._lane_live[lane] = cb.live + len(backlog)
                T = self.tts_fill_first
                lower_full = T <= 0 or all(self._lane_live.get(j, 0) >= T for j in range(lane))
                if not cb.live and not backlog:
                    if stop or self._stopping:
                        return
                    if not lower_full:               # fill first: an idle upper lane leaves the queue to the lanes below it
                        time.sleep(0.002)
                        continue
                    try:
                        it = self._tts_q.get(timeout=0.05)
                    except queue.Empty:
                        continue
                    if it is _STOP:
                        return
                    backlog.append(it)
                    if self.batch_window_s > 0:
                        time.sleep(self.batch_window_s)
                others = [v for k, v in list(self._lane_live.items()) if k != lane]
                room = cb.free_slots - len(backlog)
                mine = cb.live + len(backlog)
                least = not others or mine <= min(others) + 8
                if T > 0:        # fill first: up to T sessions per lane in lane order; once every lane holds T, the least loaded ones take the rest
                    n_lanes = max(self.tts_lanes, lane + 1)
                    all_full = mine >= T and all(self._lane_live.get(j, 0) >= T for j in range(n_lanes))
                    take = lower_full and (mine < T or (all_full and least))
                    room = min(room, T - mine) if mine < T else room
                else:
                    take = least
                if room > 0 and not stop and take:
                    for it in self._drain(self._tts_q, room):
                        if it is _STOP:
                            stop = True
                            break
                        backlog.append(it)
                backlog = self._drop_cancelled(backlog)
                now = time.perf_counter()
                can_admit = bool(backlog) and cb.free_slots > 0
                waited = now - min(r.t_submit for r in backlog) if backlog else 0.0
                due = cb.live > 0 and (stop or cb.min_lead_s(now) <= self.tts_lead_s)
                admit_now = can_admit and (cb.live == 0 or due or stop or waited >= self.tts_admit_wait_s)
                if not admit_now and not due:
                    time.sleep(0.002)                # everybody is far enough ahead of their playback: leave the device to STT
                    continue
                n0 = len(backlog)
                t_a = time.perf_counter()
                if admit_now:
                    backlog = cb.admit(backlog)
                    for job, err in cb.rejected:     # a request the engine cannot take fails alone
                        job.chunks.put(err)
                    n0 -= len(cb.rejected)
                    cb.rejected.clear()
                t_r = time.perf_counter()
                rows = cb.rows()
                for job, chunk in cb.step():
                    if chunk is None or not job.cancelled.is_set():
                        job.chunks.put(chunk)
                with self._stats_lock:
                    if n0 > len(backlog):
                        self.stats["tts_batches"] += 1
                        self.stats["tts_requests"] += n0 - len(backlog)
                        self.stats["tts_admit_s"] += t_r - t_a
                    self.stats["tts_rounds"] += 1
                    self.stats["tts_round_s"] += time.perf_counter() - t_r
                    self.stats["tts_row_rounds"][rows] = self.stats["tts_row_rounds"].get(rows, 0) + 1
            except BaseException as e:  # noqa: BLE001  (engine failure: every live and queued session of THIS lane is failed, the lane goes on)
                logger.error("TTS lane %d failed: %s", lane, e, exc_info=True)
                for job in cb.abort() + backlog:
                    job.chunks.put(e)
                backlog = []

    def _run_tts_lane_wide(self, lane: int, eng: Any, first: Any, ready: threading.Event | None = None) -> None:
        """A lane as TWO frame loops: a nursery (small engine replica, own thread + stream, eager) that admits the requests and produces every
        session's first chunk, and this thread's wide loop that adopts the sessions afterwards and runs only when some session's audio lead
        drops below `tts_lead_s` (tts_continuous.py NurseryTTS / ContinuousTTS.adopt, csrc/tts.cpp s2s_tts_migrate)."""
        from .tts_continuous import ROW_BUCKETS, ContinuousTTS, NurseryTTS

        state = {"stop": False, "nursery_done": False, "error": None}
        grad_q: "queue.Queue[Any]" = queue.Queue()
        cb = nb = None
        try:
            cb = ContinuousTTS(eng, chunk_size=self.tts_chunk_size)
            neng = eng.replica(max_batch=min(self.tts_nursery_slots, int(eng.max_batch)))
            self._lane_replicas.append(neng)
            if hasattr(eng, "prepare_admission"):
                cb.prepare()
            for r in [r for r in ROW_BUCKETS if r <= cb.cap] or [cb.cap]:
                eng.step(r, 1)
            self._sync_current_stream()
        except BaseException as e:  # noqa: BLE001
            logger.error("TTS lane %d could not start: %s", lane, e, exc_info=True)
            if first is not None:
                first.chunks.put(e)
            self._lane_live[lane] = 1 << 30
            if ready is not None:
                ready.set()
            return
        nursery_ready = threading.Event()

        def nursery_loop():
            nonlocal nb
            ctx = self._lane_stream_ctx(self.tts_nursery_priority)
            ctx.__enter__()
            backlog: list = [first] if first is not None else []
            try:
                nb = NurseryTTS(neng, chunk_size=self.tts_chunk_size)
                if hasattr(neng, "prepare_admission"):
                    neng.prepare_admission()
                    nb.prepare()
                for r in [r for r in ROW_BUCKETS if r <= nb.cap] or [nb.cap]:
                    neng.step(r, 1)
                self._sync_current_stream()
            except BaseException as e:  # noqa: BLE001
                state["error"] = e
                logger.error("TTS lane %d: the nursery could not start: %s", lane, e, exc_info=True)
                for job in backlog:
                    job.chunks.put(e)
                state["nursery_done"] = True
                nursery_ready.set()
                return
            nursery_ready.set()
            while True:
                try:
                    if self._stopping:
                        err = RuntimeError("SessionPool is shut down")
                        for job in nb.abort() + backlog:
                            job.chunks.put(err)
                        self._sync_current_stream()
                        break
                    nb.reclaim()
                    self._lane_live[lane] = cb.live + nb.live + len(backlog)
                    if not nb.cohorts and not backlog:
                        if state["stop"] or self._stopping:
                            break
                        try:
                            it = self._tts_q.get(timeout=0.02)
                        except queue.Empty:
                            continue
                        if it is _STOP:
                            state["stop"] = True
                            continue
                        backlog.append(it)
                        if self.batch_window_s > 0:
                            time.sleep(self.batch_window_s)
                    room = nb.free_slots - len(backlog)
                    if room > 0 and not state["stop"]:
                        for it in self._drain(self._tts_q, room):
                            if it is _STOP:
                                state["stop"] = True
                                break
                            backlog.append(it)
                    backlog = self._drop_cancelled(backlog)
                    n0 = len(backlog)
                    t_a = time.perf_counter()
                    backlog = nb.admit(backlog)
                    for job, err in nb.rejected:
                        job.chunks.put(err)
                    n0 -= len(nb.rejected)
                    nb.rejected.clear()
                    if not nb.cohorts:
                        time.sleep(0.002)            # (every slot is parked, waiting for the wide loop to take its sessions over)
                        continue
                    t_r = time.perf_counter()
                    for job, chunk in nb.step():
                        if chunk is None or not job.cancelled.is_set():
                            job.chunks.put(chunk)
                    for g in nb.take_graduates():
                        grad_q.put(g)
                    with self._stats_lock:
                        self.stats["tts_batches"] += 1 if n0 > len(backlog) else 0
                        self.stats["tts_requests"] += max(0, n0 - len(backlog))
                        self.stats["tts_admit_s"] += t_r - t_a
                        self.stats["tts_nursery_rounds"] = self.stats.get("tts_nursery_rounds", 0) + 1
                        self.stats["tts_nursery_s"] = self.stats.get("tts_nursery_s", 0.0) + time.perf_counter() - t_r
                except BaseException as e:  # noqa: BLE001
                    logger.error("TTS lane %d nursery failed: %s", lane, e, exc_info=True)
                    for job in nb.abort() + backlog:
                        job.chunks.put(e)
                    backlog = []
            state["nursery_done"] = True

        nt = threading.Thread(target=nursery_loop, name=f"s2s-pool-{self.device}-tts{lane}-nursery", daemon=True)
        nt.start()
        nursery_ready.wait(300)
        if ready is not None:
            ready.set()
        pending: list = []
        while True:
            try:
                if self._stopping:
                    nt.join(timeout=10)
                    while True:
                        try:
                            pending.append(grad_q.get_nowait())
                        except queue.Empty:
                            break
                    err = RuntimeError("SessionPool is shut down")
                    for job in cb.abort() + [j for g in pending for j in g.jobs]:
                        job.chunks.put(err)
                    self._sync_current_stream()
                    return
                while True:
                    try:
                        pending.append(grad_q.get_nowait())
                    except queue.Empty:
                        break
                if pending and nb is not None:
                    pending = cb.adopt(pending, nb)
                if not cb.live:
                    if state["nursery_done"] and not pending and grad_q.empty():
                        return
                    time.sleep(0.002)
                    continue
                now = time.perf_counter()
                if not (state["stop"] or cb.min_lead_s(now) <= self.tts_lead_s):
                    time.sleep(0.002)                # everybody is far enough ahead of their playback: leave the device to STT and the nursery
                    continue
                t_r = time.perf_counter()
                rows = cb.rows()
                for job, chunk in cb.step():
                    if chunk is None or not job.cancelled.is_set():
                        job.chunks.put(chunk)
                with self._stats_lock:
                    self.stats["tts_rounds"] += 1
                    self.stats["tts_round_s"] += time.perf_counter() - t_r
                    self.stats["tts_row_rounds"][rows] = self.stats["tts_row_rounds"].get(rows, 0) + 1
            except BaseException as e:  # noqa: BLE001
                logger.error("TTS lane %d failed: %s", lane, e, exc_info=True)
                for job in cb.abort() + [j for g in pending for j in g.jobs]:
                    job.chunks.put(e)
                pending = []

    def _run_tts_lane(self, lane: int) -> None:
        eng, ctx = None, None
        while True:
            try:
                first = self._tts_q.get(timeout=0.05)
            except queue.Empty:
                if self._stopping:
                    return
                if self._warm_requested and eng is None and "tts" in self._engines and not self._lane_ready[lane].is_set():
                    eng, ctx = self._lane_start(lane, warm=True)
                    if eng is not None and self.continuous and hasattr(eng, "admit"):
                        self._run_tts_lane_continuous(lane, eng, None, self._lane_ready[lane])
                        return
                    self._lane_ready[lane].set()
                    if eng is None and lane > 0:
                        return                       # no replica possible: lane 0 serves everything
                continue
            if first is _STOP:
                return
            if eng is None:
                eng, ctx = self._lane_start(lane, warm=False)
                self._lane_ready[lane].set()
                if eng is None:                      # hand the request back to the lanes that have an engine and retire
                    self._tts_q.put(first)
                    return
            if self.continuous and hasattr(eng, "admit"):
                self._run_tts_lane_continuous(lane, eng, first)
                return
            if self.batch_window_s > 0:
                time.sleep(self.batch_window_s)
            drained = self._drain(self._tts_q, min(self.max_batch, int(getattr(eng, "max_batch", self.max_batch))) - 1)
            for it in drained:
                if it is _STOP:
                    self._tts_q.put(_STOP)           # leave it for shutdown handling after this batch
            batch = self._drop_cancelled([first] + [d for d in drained if d is not _STOP])
            if not batch:
                continue
            with self._stats_lock:
                self.stats["tts_batches"] += 1
                self.stats["tts_requests"] += len(batch)
            try:
                kw = {"min_frames": max(r.min_frames for r in batch)} if any(r.min_frames for r in batch) else {}
                gen = eng.stream([r.prompt for r in batch], [r.max_new_tokens for r in batch], chunk_size=self.tts_chunk_size,
                                 ctx_frames=self.tts_ctx_frames, **kw)
                for chunks in gen:
                    for r, ch in zip(batch, chunks):
                        if ch is not None and not r.cancelled.is_set():
                            r.chunks.put(ch)
                    if all(r.cancelled.is_set() for r in batch):
                        break
                for r in batch:
                    r.chunks.put(None)
            except BaseException as e:  # noqa: BLE001
                logger.error("TTS batch failed: %s", e, exc_info=True)
                for r in batch:
                    r.chunks.put(e)

    def _run_stt(self, reqs: list[_STTRequest], model: str, engine: Any = None) -> None:
        if not reqs:
            return
        with self._stats_lock:
            self.stats["stt_batches"] += 1
            self.stats["stt_requests"] += len(reqs)
            self.stats["max_stt_batch"] = max(self.stats["max_stt_batch"], len(reqs))
        try:
            t_b = time.perf_counter()
            results = (engine or self._engines[model]).transcribe_batch([r.audio for r in reqs], [r.options for r in reqs])
            with self._stats_lock:
                self.stats["stt_busy_s"] += time.perf_counter() - t_b
            for r, res in zip(reqs, results):
                r.future.set_result(res)
        except BaseException as e:  # noqa: BLE001
            logger.error("%s batch failed: %s", model, e, exc_info=True)
            for r in reqs:
                if not r.future.done():
                    r.future.set_exception(e)
