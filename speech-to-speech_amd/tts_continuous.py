"""Continuous batching of the Qwen3-TTS frame loop on one engine (one GPU stream, one thread).

The reference streams every session independently (TTS/qwen3_tts_handler.py:695-749: one `_stream()` generator per request).
Batching sessions into turn-batches that start and end together (round 2) made a request that arrived just after a batch had
started wait for a free lane - the queueing, not the device, capped the sustained session count.  Here the rows ("slots") of ONE
frame loop are shared: at every chunk boundary (`chunk_size` codec frames) finished sessions leave, queued sessions are prefilled
into free slots and join, and the frame loop goes on for everybody - every slot carries its own frame index / eos gate / budget on
the device (csrc/tts.cpp: s2s_tts_admit / _step / _read_codes / _release).

Sessions admitted together form a COHORT on a contiguous slot range.  The codec decoder's streaming state (conv history rows,
pre-transformer cache) is per row, but its first chunk has a different geometry than a continuation (causal transposed convs trim
their first output group), so the codec runs once per cohort and chunk on the cohort's rows (s2s_codec_decode_rows) - cohorts of
different ages stream side by side.  A session's codes and samples do not depend on its slot, its cohort or its batch-mates
(tests/test_tts_continuous_gpu.py compares a session admitted mid-stream with the same session run alone and with the oracle).
"""

from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Any, Iterator, Sequence

import numpy as np
import torch

ROW_BUCKETS = (32, 64, 128)      # frame-loop row counts with a captured graph (2 * rows must be <= 128 or a multiple of 128: predictor prefill pass)


@dataclass
class TTSJob:
    """What the scheduler needs from a request (the pool's _TTSRequest has the same fields)."""

    prompt: Any
    max_new_tokens: int
    min_frames: int = 0
    cancelled: threading.Event = field(default_factory=threading.Event)


@dataclass
class _Cohort:
    slot0: int
    jobs: list
    frames: int = 0                 # frames generated (= decoded) so far
    ended: list = field(default_factory=list)     # per job: end marker delivered
    t_first: float | None = None    # wall time its first chunk was handed out (playback starts then)


class ContinuousTTS:
    def __init__(self, engine, chunk_size: int = 8, max_cohorts: int = 24):
        self.eng = engine
        self.cap = int(engine.max_batch)
        self.chunk = int(chunk_size)
        self.max_cohorts = int(max_cohorts)
        self.record_codes = False        # tests: append every chunk's codes [n_frames][G] (numpy) to job.codes
        self.free: list[tuple[int, int]] = [(0, self.cap)]      # (start, length), sorted by start
        self.cohorts: list[_Cohort] = []
        self.next_utt = 1
        self._pinned: dict[tuple, torch.Tensor] = {}
        self._events: list = []
        self.stats = {"admitted": 0, "cohorts": 0, "rounds": 0, "max_live": 0, "row_rounds": {}}
        self.eng.release(0, self.cap)                           # every slot idle

    def prepare(self) -> None:
        """Allocate every pinned staging row and event the scheduler can ever need (the session pool calls this while the lane warms
        up: a pinned allocation in the middle of serving stalls the device for every stream)."""
        for ci in range(self.max_cohorts):
            self._host(("pcm", ci), (1, 1), torch.int16)
            self._host(("nfr", ci), (1,), torch.int32)
            self._host(("nval", ci), (1,), torch.int32)
            self._events.append(torch.cuda.Event())

    # ---- slots ------------------------------------------------------------------------------------
    @property
    def live(self) -> int:
        return sum(len(c.jobs) for c in self.cohorts)

    @property
    def free_slots(self) -> int:
        return sum(n for _, n in self.free)

    def _take(self, want: int) -> tuple[int, int]:
        """Lowest free range that holds `want` slots, else the largest one (partly filled requests come back as a remainder)."""
        best = None
        for i, (s, n) in enumerate(self.free):
            if n >= want:
                best = i
                break
        if best is None:
            best = max(range(len(self.free)), key=lambda i: self.free[i][1])
        s, n = self.free[best]
        k = min(n, want)
        if k == n:
            self.free.pop(best)
        else:
            self.free[best] = (s + k, n - k)
        return s, k

    def _give(self, s: int, n: int) -> None:
        self.free.append((s, n))
        self.free.sort()
        merged: list[tuple[int, int]] = []
        for a, b in self.free:
            if merged and merged[-1][0] + merged[-1][1] == a:
                merged[-1] = (merged[-1][0], merged[-1][1] + b)
            else:
                merged.append((a, b))
        self.free = merged

    # ---- admission --------------------------------------------------------------------------------
    def admit(self, jobs: Sequence[Any]) -> list:
        """Prefill as many of `jobs` as there are free slots (contiguous ranges -> cohorts); returns the jobs that did not fit."""
        jobs = [j for j in jobs if not j.cancelled.is_set()]
        while jobs and self.free and len(self.cohorts) < self.max_cohorts:
            s, k = self._take(len(jobs))
            group, jobs = list(jobs[:k]), list(jobs[k:])
            budgets = [max(1, min(int(j.max_new_tokens), self.eng.max_frames)) for j in group]
            self.eng.admit([j.prompt for j in group], s, budgets, [int(j.min_frames) for j in group], utterance_id0=self.next_utt)
            self.next_utt += k
            self.cohorts.append(_Cohort(s, group, 0, [False] * k))
            self.stats["admitted"] += k
            self.stats["cohorts"] += 1
        self.stats["max_live"] = max(self.stats["max_live"], self.live)
        return list(jobs)

    # ---- one chunk for everybody --------------------------------------------------------------------
    def _host(self, key: tuple, shape, dtype) -> torch.Tensor:
        """Pinned staging rows, allocated ONCE per cohort index at the largest size it can ever need (a pinned allocation synchronises
        the whole device - every other stream, the STT worker's included - so nothing is allocated while serving); returned as a
        contiguous view of the requested shape."""
        flat = self._pinned.get(key)
        if flat is None:
            width = (2 * self.chunk * self.eng.dims.total_upsample + 2) // 3 if key[0] == "pcm" else 1
            flat = self._pinned[key] = torch.empty((self.cap * width,), dtype=dtype, pin_memory=True)
        n = 1
        for d in shape:
            n *= int(d)
        return flat[:n].view(*shape)

    def rows(self) -> int:
        hi = max((c.slot0 + len(c.jobs) for c in self.cohorts), default=0)
        for r in ROW_BUCKETS:
            if hi <= r and r <= self.cap:
                return r
        return self.cap

    def step(self) -> Iterator[tuple[Any, np.ndarray | None]]:
        """`chunk_size` more frames for every live session; yields (job, int16 16 kHz chunk) and, once per job, (job, None) at its end.
        Cohorts are decoded youngest first, and each one's audio is handed out as soon as ITS copies have landed."""
        if not self.cohorts:
            return
        eng, n = self.eng, self.chunk
        rows = self.rows()
        self.stats["rounds"] += 1
        self.stats["row_rounds"][rows] = self.stats["row_rounds"].get(rows, 0) + 1
        eng.step(rows, n)
        pending = []
        for ci, c in enumerate(sorted(self.cohorts, key=lambda c: c.frames)):
            k = len(c.jobs)
            codes, nfr, nvalid = eng.read_codes(c.slot0, k, c.frames, n)
            wav = eng.codec_decode_rows(codes, c.slot0, c.frames)
            pcm = eng.resample_int16(wav, nvalid)       # samples past a session's last valid frame are silence to the resampler
            h_pcm = self._host(("pcm", ci), pcm.shape, torch.int16)
            h_nfr = self._host(("nfr", ci), (k,), torch.int32)
            h_nval = self._host(("nval", ci), (k,), torch.int32)
            h_pcm.copy_(pcm, non_blocking=True)
            h_nfr.copy_(nfr, non_blocking=True)
            h_nval.copy_(nvalid, non_blocking=True)
            h_codes = None
            if self.record_codes:
                h_codes = torch.empty(codes.shape, dtype=torch.int32, pin_memory=True)      # (tests only)
                h_codes.copy_(codes, non_blocking=True)
            while len(self._events) <= ci:
                self._events.append(torch.cuda.Event())
            ev = self._events[ci]
            ev.record()
            pending.append((c, h_pcm, (h_nfr, h_nval), ev, (pcm, nfr, codes, wav, nvalid), h_codes))
        import time as _time

        for c, h_pcm, h_nfr, ev, _keep, h_codes in pending:
            ev.synchronize()
            if c.t_first is None:
                c.t_first = _time.perf_counter()
            pcm_h, nfr_h, nval_h = h_pcm.numpy(), h_nfr[0].numpy(), h_nfr[1].numpy()       # (views of the pinned rows)
            for i, job in enumerate(c.jobs):
                if c.ended[i]:
                    continue
                if h_codes is not None:
                    if not hasattr(job, "codes"):
                        job.codes = []
                    job.codes.append(h_codes[i].numpy().copy())
                budget = max(1, min(int(job.max_new_tokens), eng.max_frames))
                valid = int(min(nfr_h[i], budget)) - c.frames
                if job.cancelled.is_set():
                    valid = 0
                if valid >= n:
                    yield job, pcm_h[i].copy()
                elif valid > 0:      # the utterance ended inside this chunk: only the samples of its valid frames
                    yield job, pcm_h[i][: (2 * int(nval_h[i]) + 2) // 3].copy()
                if valid < n or c.frames + n >= budget:
                    c.ended[i] = True
                    yield job, None
            c.frames += n
        for c in [c for c in self.cohorts if all(c.ended)]:
            eng.release(c.slot0, len(c.jobs))
            self._give(c.slot0, len(c.jobs))
            self.cohorts.remove(c)

    FRAME_S = 0.08                   # one codec frame = 80 ms of audio (12.5 Hz)

    def min_lead_s(self, now: float) -> float:
        """Smallest audio lead over the live cohorts: seconds of already delivered audio that playback has not reached yet
        (first chunk handed out at t_first, frames * 80 ms delivered).  A cohort still waiting for its first chunk has lead -inf."""
        lead = float("inf")
        for c in self.cohorts:
            lead = min(lead, float("-inf") if c.t_first is None else c.t_first + c.frames * self.FRAME_S - now)
        return lead

    def drain(self) -> Iterator[tuple[Any, np.ndarray | None]]:
        while self.cohorts:
            yield from self.step()

    def abort(self, err: BaseException | None = None) -> list:
        """Drop every live session (engine failure / shutdown); returns their jobs."""
        jobs = [j for c in self.cohorts for j, e in zip(c.jobs, c.ended) if not e]
        for c in self.cohorts:
            try:
                self.eng.release(c.slot0, len(c.jobs))
            except Exception:  # noqa: BLE001
                pass
        self.cohorts = []
        self.free = [(0, self.cap)]
        return jobs
