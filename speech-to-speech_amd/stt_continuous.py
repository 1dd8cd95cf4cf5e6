"""Continuous batching of the Whisper token loop (host scheduler; device side: csrc/whisper.cpp ``s2s_whisper_slots_*``).

The reference transcribes every final with its own ``model.generate()`` (STT/whisper_stt_handler.py:225-282).  The session pool's
batched worker runs log-mel -> encoder -> token loop for one batch and only then looks at the queue again, so under load a transcript
waits for the whole batch in front of it, and the token loop - 35 steps of ~260 dependent launches whatever the batch - is paid once
per batch of 8 .. 25 requests.  Here the token loop never stops: its rows are SLOTS.  A group of queued requests gets free slots, its
log-mel + encoder pass runs on a SECOND stream (the encoder's scratch and the loop's state are disjoint buffers of the same engine)
and writes the cross K|V into those slots' rows; when the pass has completed the group joins the loop at the next round boundary; a
session leaves at ITS eos / token budget and its transcript is delivered at once.  One encoder group is in flight at a time (the
front-end's pinned staging block is reused per group).

STATUS: written at the end of round 4 without GPU time left - covered by the CPU scheduler tests (stub engine) and by
``tests/test_whisper_slots_gpu.py``, which is skipped unless ``S2S_WHISPER_SLOTS=1``.  ``SessionPool(stt_slots=...)`` is off by default.
"""

from __future__ import annotations

import contextlib
import time
from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

ROW_BUCKETS = (8, 16, 32, 64, 128)


def _bucket(rows: int, cap: int) -> int:
    for b in ROW_BUCKETS:
        if rows <= b <= cap:
            return b
    return cap


@dataclass
class _Live:
    job: Any
    prompt: list
    max_new: int
    t_admit: float


@dataclass
class _Group:
    jobs: list
    slot0: int
    prompts: list
    max_new: list
    event: Any                    # .query() -> True once the encoder pass has completed (None: synchronous engine)
    keep: Any = None              # the feature tensor: referenced until the group is admitted
    t0: float = field(default_factory=time.perf_counter)


class ContinuousSTT:
    """One running greedy loop over ``engine.max_batch`` slots.  Jobs are opaque to the scheduler except for ``job.audio`` and
    ``job.options`` (language / task / max_new_tokens / vocab, as ``WhisperEngine.transcribe_batch`` takes them); finished jobs are
    returned by ``step()`` as ``(job, result_dict)``.

    ``enc_stream``: context-manager factory entered around the encoder pass (``torch.cuda.stream(side_stream)`` on a GPU, nullcontext
    in the CPU tests); ``make_event``: -> an event recorded on the current stream (``query()``), or None when the engine is
    synchronous; ``wait_event``: makes the loop's stream wait for a group's event before its admission kernel."""

    def __init__(self, engine, steps_per_round: int = 4, max_group: int = 16, enc_stream: Callable[[], Any] | None = None,
                 make_event: Callable[[], Any] | None = None, wait_event: Callable[[Any], None] | None = None):
        self.eng = engine
        self.cap = int(engine.max_batch)
        self.steps_per_round = max(1, int(steps_per_round))
        self.max_group = max(1, int(max_group))
        self.enc_stream = enc_stream or contextlib.nullcontext
        self.make_event = make_event or (lambda: None)
        self.wait_event = wait_event or (lambda ev: None)
        self.free: list[tuple[int, int]] = [(0, self.cap)]          # sorted, disjoint [lo, hi) ranges
        self.rows: dict[int, _Live] = {}
        self.group: _Group | None = None                            # the encoder group in flight
        self.vocab = None
        self.rejected: list[tuple[Any, BaseException]] = []        # jobs of the last encode() call that the running loop cannot serve
        self.stats = {"groups": 0, "admitted": 0, "rounds": 0, "steps": 0, "max_live": 0, "row_rounds": {}}

    # ---- slots -------------------------------------------------------------------------------------
    @property
    def live(self) -> int:
        return len(self.rows)

    @property
    def n_free(self) -> int:
        return sum(hi - lo for lo, hi in self.free)

    def _take(self, n: int) -> tuple[int, int] | None:
        """Lowest free range, at most n slots of it (the group must be contiguous: one encoder pass writes slot0 .. slot0 + k - 1)."""
        if not self.free:
            return None
        lo, hi = self.free[0]
        k = min(n, hi - lo)
        if lo + k == hi:
            self.free.pop(0)
        else:
            self.free[0] = (lo + k, hi)
        return lo, k

    def _give(self, lo: int, n: int) -> None:
        self.free.append((lo, lo + n))
        self.free.sort()
        merged: list[tuple[int, int]] = []
        for a, b in self.free:
            if merged and merged[-1][1] >= a:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        self.free = merged

    # ---- encoder groups ------------------------------------------------------------------------------
    def can_encode(self) -> bool:
        return self.group is None and self.n_free > 0

    def encode(self, jobs: Sequence[Any]) -> list:
        """Start the encoder pass of (a prefix of) `jobs` into free slots; returns the jobs that did not fit (no free slots left in the
        lowest range, group cap) - the caller re-queues them.  Jobs whose options the running loop cannot serve fail alone."""
        assert self.group is None, "one encoder group at a time"
        jobs, rejected = list(jobs), []
        ok = []
        for j in jobs:
            vocab = j.options.get("vocab")
            if self.vocab is None and vocab is not None and not self.rows:
                self.eng.slots_config(vocab)
                self.vocab = vocab
            if vocab is None or (vocab is not self.vocab and vocab != self.vocab):
                rejected.append((j, ValueError("the running Whisper loop serves one vocabulary; this request names another")))
                continue
            n_new = int(j.options.get("max_new_tokens", 128))
            prompt = list(vocab.prompt(j.options.get("language"), j.options.get("task", "transcribe")))
            if not (1 <= len(prompt) <= 8) or n_new < 1 or len(prompt) + n_new > int(self.eng.dims.max_target):
                rejected.append((j, ValueError(f"prefix of {len(prompt)} tokens + max_new_tokens {n_new} does not fit {self.eng.dims.max_target} target positions")))
                continue
            ok.append((j, prompt, n_new))
        self.rejected = rejected
        if not ok:
            return []
        # one prefix length per group (the admission kernel takes [n][P]); other lengths wait for the next group
        P = len(ok[0][1])
        mine = [t for t in ok if len(t[1]) == P][: self.max_group]
        got = self._take(len(mine))
        if got is None:
            return [t[0] for t in ok]
        slot0, k = got
        taken = {id(t) for t in mine[:k]}
        mine, left = mine[:k], [t[0] for t in ok if id(t) not in taken]
        with self.enc_stream():
            keep = self.eng.slots_encode([t[0].audio for t in mine], slot0)
            ev = self.make_event()
        self.group = _Group([t[0] for t in mine], slot0, [t[1] for t in mine], [t[2] for t in mine], ev, keep)
        self.stats["groups"] += 1
        return left

    def _admit_ready(self, block: bool = False) -> None:
        g = self.group
        if g is None:
            return
        if g.event is not None and not block and not g.event.query():
            return
        if g.event is not None:
            self.wait_event(g.event)
        self.eng.slots_admit(g.slot0, g.prompts, g.max_new)
        now = time.perf_counter()
        for i, j in enumerate(g.jobs):
            self.rows[g.slot0 + i] = _Live(j, g.prompts[i], g.max_new[i], now)
        self.stats["admitted"] += len(g.jobs)
        self.stats["max_live"] = max(self.stats["max_live"], len(self.rows))
        self.group = None

    # ---- the loop ------------------------------------------------------------------------------------
    def step(self) -> list[tuple[Any, dict]]:
        """One round: admit the encoder group if its pass has completed (or block on it when nothing else is live), run
        `steps_per_round` greedy steps over the live rows, deliver the sessions that finished."""
        self._admit_ready(block=not self.rows)
        if not self.rows:
            return []
        rows = _bucket(max(self.rows) + 1, self.cap)
        self.eng.slots_step(rows, self.steps_per_round)
        pack = self.eng.slots_collect(rows)
        self.stats["rounds"] += 1
        self.stats["steps"] += self.steps_per_round
        self.stats["row_rounds"][rows] = self.stats["row_rounds"].get(rows, 0) + 1
        done = []
        for slot in sorted(self.rows):
            if int(pack[slot][0]):
                r = self.rows.pop(slot)
                done.append((slot, r.job, self.eng.slot_result(pack[slot], r.prompt, r.max_new)))
        # release runs of adjacent slots with one call each
        i = 0
        while i < len(done):
            k = i
            while k + 1 < len(done) and done[k + 1][0] == done[k][0] + 1:
                k += 1
            self.eng.slots_release(done[i][0], k - i + 1)
            self._give(done[i][0], k - i + 1)
            i = k + 1
        return [(j, res) for _, j, res in done]

    @property
    def idle(self) -> bool:
        return not self.rows and self.group is None

    def abort(self) -> list:
        """Drop every live / encoding session (engine failure, shutdown); returns their jobs."""
        jobs = [r.job for r in self.rows.values()] + (self.group.jobs if self.group else [])
        try:
            if self.rows or self.group:
                self.eng.slots_release(0, self.cap)
        except Exception:  # noqa: BLE001
            pass
        self.rows, self.group, self.free = {}, None, [(0, self.cap)]
        return jobs
