"""Continuous batching of the Qwen3-TTS frame loop on one engine (one GPU stream, one thread).

The reference streams every session independently (TTS/qwen3_tts_handler.py:695-749: one `_stream()` generator per request).
Batching sessions into turn-batches that start and end together (round 2) made a request that arrived just after a batch had
started wait for a free lane - the queueing, not the device, capped the sustained session count.  Here the rows ("slots") of ONE
frame loop are shared: at every chunk boundary (`chunk_size` codec frames) finished sessions leave, queued sessions are prefilled
into free slots and join, and the frame loop goes on for everybody - every slot carries its own frame index / eos gate / budget on
the device (csrc/tts.cpp: s2s_tts_admit / _step / _read_codes / _release).

Round 4: the loop is WIDE (up to 512 slots on the row-block decode projections of csrc/gemm_decw.hip) and a round's launch count no
longer depends on how many admission groups are alive:

* sessions admitted together form a COHORT on a contiguous slot range only for their FIRST chunk (the codec decoder's first chunk has
  its own geometry: causal transposed convs trim their first output group) - one s2s_codec_decode_rows(frames_done = 0) per new cohort;
* from their second chunk on, sessions are independent ROWS: all continuing rows of the lane go through ONE codec pass per round
  (s2s_tts_read_codes_rows + s2s_codec_decode_rows_at with per-row stream positions), whatever their ages;
* a session's slot is released the moment IT ends, and new sessions take the lowest free slots, so the rows of the frame loop stay
  compact and the round runs at the smallest captured row count that holds them.

A session's codes and samples do not depend on its slot, its cohort or its batch-mates (tests/test_tts_continuous_gpu.py compares
sessions admitted mid-stream with the same sessions run alone and with the oracle).
"""

from __future__ import annotations

import threading
import time
from dataclasses import dataclass, field
from typing import Any, Iterator, Sequence

import numpy as np
import torch

ROW_BUCKETS = (32, 64, 128, 192, 256, 384, 512)      # frame-loop row counts with a captured graph


def _stream_wait(ev) -> None:
    """Order the current stream behind `ev` (recorded on another lane's stream); host stand-ins without a GPU are waited for directly."""
    if ev is None:
        return
    if torch.cuda.is_available():
        torch.cuda.current_stream().wait_event(ev)
    else:
        ev.synchronize()


@dataclass
class TTSJob:
    """What the scheduler needs from a request (the pool's _TTSRequest has the same fields)."""

    prompt: Any
    max_new_tokens: int
    min_frames: int = 0
    cancelled: threading.Event = field(default_factory=threading.Event)


@dataclass
class _Cohort:
    """Sessions admitted together, until their first chunk has been decoded."""

    slot0: int
    jobs: list


@dataclass
class _Graduates:
    """Sessions whose first chunk a NurseryTTS has delivered, waiting to be adopted by the wide loop: a contiguous run of nursery slots."""

    jobs: list
    slot0: int                      # nursery slots [slot0, slot0 + len(jobs))
    frames: int                     # frames generated (= decoded) so far
    n_pos: int                      # talker positions to carry over (longest prompt of the run + frames)
    t_first: float
    event: Any                      # recorded on the nursery's stream behind the run's last launch
    x: Any = None                   # snapshot of the run's next talker input rows (the nursery loop goes on over the parked slots)


@dataclass
class _Row:
    """A session from its second chunk on."""

    job: Any
    slot: int
    frames: int                     # frames generated (= decoded) so far
    t_first: float                  # wall time its first chunk was handed out (playback starts then)


class ContinuousTTS:
    def __init__(self, engine, chunk_size: int = 8, max_cohorts: int = 8):
        self.eng = engine
        self.cap = int(engine.max_batch)
        self.chunk = int(chunk_size)
        self.max_cohorts = int(max_cohorts)     # admission groups per round (each costs one first-chunk codec pass)
        self.record_codes = False        # tests: append every chunk's codes [n_frames][G] (numpy) to job.codes
        self.free: list[tuple[int, int]] = [(0, self.cap)]      # (start, length), sorted by start
        self.cohorts: list[_Cohort] = []         # admitted, first chunk not yet decoded
        self.rows_live: dict[int, _Row] = {}     # slot -> continuing session
        self.rejected: list[tuple[Any, BaseException]] = []     # jobs refused at admission (bad prompt): the pool fails only these
        self.next_utt = 1
        self._pinned: dict[tuple, torch.Tensor] = {}
        self._events: dict[Any, Any] = {}
        self.stats = {"admitted": 0, "cohorts": 0, "rounds": 0, "max_live": 0, "row_rounds": {}, "codec_passes": 0}
        self.eng.release(0, self.cap)                           # every slot idle

    def prepare(self) -> None:
        """Allocate every pinned staging row and event the scheduler can ever need (the session pool calls this while the lane warms
        up: a pinned allocation in the middle of serving stalls the device for every stream)."""
        for key in ["cont"] + list(range(self.max_cohorts)):
            self._host(("pcm", key), (1, 1), torch.int16)
            self._host(("nfr", key), (1,), torch.int32)
            self._host(("nval", key), (1,), torch.int32)
            self._event(key)
        self._host(("frames", "cont"), (1,), torch.int32)

    # ---- slots ------------------------------------------------------------------------------------
    @property
    def live(self) -> int:
        return sum(len(c.jobs) for c in self.cohorts) + len(self.rows_live)

    @property
    def free_slots(self) -> int:
        return sum(n for _, n in self.free)

    def _take(self, want: int) -> tuple[int, int]:
        """The LOWEST free range, up to `want` slots of it (keeps the occupied rows compact: the frame loop runs over [0, highest
        occupied slot], rounded up to a captured row count)."""
        s, n = self.free[0]
        k = min(n, want)
        if k == n:
            self.free.pop(0)
        else:
            self.free[0] = (s + k, n - k)
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
    def _check(self, job: Any) -> BaseException | None:
        """A request the engine cannot take fails ALONE (it must not take the lane's live sessions with it)."""
        try:
            emb = getattr(job.prompt, "embeds", None)
            if emb is None:              # (not an embedding prompt - e.g. a test double's: the engine's own checks apply)
                return None
            if emb.ndim != 2 or emb.shape[1] != self.eng.dims.t_hidden:
                return ValueError(f"TTS prompt rows have width {emb.shape[-1]}, the engine expects {self.eng.dims.t_hidden}")
            if not 2 <= emb.shape[0] <= self.eng.max_prompt:
                return ValueError(f"TTS prompt of {emb.shape[0]} rows (engine: 2 .. {self.eng.max_prompt})")
            if int(job.max_new_tokens) < 1:
                return ValueError("max_new_tokens must be >= 1")
        except Exception as e:  # noqa: BLE001
            return e
        return None

    def admit(self, jobs: Sequence[Any]) -> list:
        """Prefill as many of `jobs` as there are free slots (contiguous ranges -> cohorts); returns the jobs that did not fit.  Jobs the
        engine cannot take are moved to ``self.rejected`` with their error."""
        ok = []
        for j in jobs:
            if j.cancelled.is_set():
                continue
            err = self._check(j)
            if err is not None:
                self.rejected.append((j, err))
            else:
                ok.append(j)
        jobs = ok
        while jobs and self.free and len(self.cohorts) < self.max_cohorts:
            s, k = self._take(len(jobs))
            group, jobs = list(jobs[:k]), list(jobs[k:])
            budgets = [max(1, min(int(j.max_new_tokens), self.eng.max_frames)) for j in group]
            try:
                self.eng.admit([j.prompt for j in group], s, budgets, [int(j.min_frames) for j in group], utterance_id0=self.next_utt)
            except Exception as e:  # noqa: BLE001  (per cohort: the sessions already live keep streaming)
                self._give(s, k)
                self.rejected.extend((j, e) for j in group)
                continue
            self.next_utt += k
            self.cohorts.append(_Cohort(s, group))
            self.stats["admitted"] += k
            self.stats["cohorts"] += 1
        self.stats["max_live"] = max(self.stats["max_live"], self.live)
        return list(jobs)

    # ---- one chunk for everybody --------------------------------------------------------------------
    def _host(self, key: tuple, shape, dtype) -> torch.Tensor:
        """Pinned staging rows, allocated ONCE per pass at the largest size it can ever need (a pinned allocation synchronises
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

    def _event(self, key):
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        return ev

    def rows(self) -> int:
        hi = max([c.slot0 + len(c.jobs) for c in self.cohorts] + [s + 1 for s in self.rows_live], default=0)
        for r in ROW_BUCKETS:
            if hi <= r and r <= self.cap:
                return r
        return self.cap

    def _stage(self, key, pcm, nfr, nvalid, codes):
        k = pcm.shape[0]
        h_pcm = self._host(("pcm", key), pcm.shape, torch.int16)
        h_nfr = self._host(("nfr", key), (k,), torch.int32)
        h_nval = self._host(("nval", key), (k,), torch.int32)
        h_pcm.copy_(pcm, non_blocking=True)
        h_nfr.copy_(nfr, non_blocking=True)
        h_nval.copy_(nvalid, non_blocking=True)
        h_codes = None
        if self.record_codes:
            h_codes = torch.empty(codes.shape, dtype=torch.int32, pin_memory=True)      # (tests only)
            h_codes.copy_(codes, non_blocking=True)
        ev = self._event(key)
        ev.record()
        return h_pcm, h_nfr, h_nval, h_codes, ev

    def _deliver(self, job, frames0: int, n: int, nfr: int, nval: int, pcm_row, codes_row):
        """-> (items to yield for this job, ended).  frames0: frames the session had before this chunk."""
        out = []
        if codes_row is not None:
            if not hasattr(job, "codes"):
                job.codes = []
            job.codes.append(codes_row.copy())
        budget = max(1, min(int(job.max_new_tokens), self.eng.max_frames))
        valid = int(min(nfr, budget)) - frames0
        if job.cancelled.is_set():
            valid = 0
        if valid >= n:
            out.append((job, pcm_row.copy()))
        elif valid > 0:      # the utterance ended inside this chunk: only the samples of its valid frames
            out.append((job, pcm_row[: (2 * int(nval) + 2) // 3].copy()))
        ended = valid < n or frames0 + n >= budget
        if ended:
            out.append((job, None))
        return out, ended

    def step(self) -> Iterator[tuple[Any, np.ndarray | None]]:
        """`chunk_size` more frames for every live session; yields (job, int16 16 kHz chunk) and, once per job, (job, None) at its end.
        One codec pass for ALL continuing rows, one per newly admitted cohort; each pass's audio is handed out as soon as ITS copies
        have landed."""
        if not self.cohorts and not self.rows_live:
            return
        eng, n = self.eng, self.chunk
        rows = self.rows()
        self.stats["rounds"] += 1
        self.stats["row_rounds"][rows] = self.stats["row_rounds"].get(rows, 0) + 1
        eng.step(rows, n)
        cont = None
        if self.rows_live:
            # Continuation pass over the slot range that holds every continuing row.  Slots of the range without a continuing row (idle,
            # or a cohort whose first chunk follows below) are computed with position 0 and ignored: their codec state is rewritten by
            # the first chunk of the stream that owns / will own the slot - which is why this pass runs BEFORE the first-chunk passes.
            lo, hi = min(self.rows_live), max(self.rows_live) + 1
            k = hi - lo
            h_fr = self._host(("frames", "cont"), (k,), torch.int32)
            h_fr.zero_()
            fr_np = h_fr.numpy()
            for s, r in self.rows_live.items():
                fr_np[s - lo] = r.frames
            d_fr = h_fr.to(eng.device, non_blocking=True)
            codes, nfr, nvalid = eng.read_codes_rows(lo, k, d_fr, n)
            wav = eng.codec_decode_rows_at(codes, lo, d_fr, max(r.frames for r in self.rows_live.values()))
            pcm = eng.resample_int16(wav, nvalid)
            cont = (lo, k, self._stage("cont", pcm, nfr, nvalid, codes), (pcm, nfr, codes, wav, nvalid, d_fr))
            self.stats["codec_passes"] += 1
        firsts = []
        for ci, c in enumerate(self.cohorts):
            k = len(c.jobs)
            codes, nfr, nvalid = eng.read_codes(c.slot0, k, 0, n)
            wav = eng.codec_decode_rows(codes, c.slot0, 0)
            pcm = eng.resample_int16(wav, nvalid)       # samples past a session's last valid frame are silence to the resampler
            firsts.append((c, self._stage(ci, pcm, nfr, nvalid, codes), (pcm, nfr, codes, wav, nvalid)))
            self.stats["codec_passes"] += 1

        ended_slots: list[int] = []
        if cont is not None:
            lo, k, (h_pcm, h_nfr, h_nval, h_codes, ev), _keep = cont
            ev.synchronize()
            pcm_h, nfr_h, nval_h = h_pcm.numpy(), h_nfr.numpy(), h_nval.numpy()       # (views of the pinned rows)
            for s in sorted(self.rows_live):
                r = self.rows_live[s]
                i = s - lo
                items, ended = self._deliver(r.job, r.frames, n, nfr_h[i], nval_h[i], pcm_h[i], h_codes[i].numpy() if h_codes is not None else None)
                yield from items
                r.frames += n
                if ended:
                    ended_slots.append(s)
        for c, (h_pcm, h_nfr, h_nval, h_codes, ev), _keep in firsts:
            ev.synchronize()
            t_first = time.perf_counter()
            pcm_h, nfr_h, nval_h = h_pcm.numpy(), h_nfr.numpy(), h_nval.numpy()
            for i, job in enumerate(c.jobs):
                items, ended = self._deliver(job, 0, n, nfr_h[i], nval_h[i], pcm_h[i], h_codes[i].numpy() if h_codes is not None else None)
                yield from items
                if ended:
                    ended_slots.append(c.slot0 + i)
                else:
                    self._continue(job, c.slot0 + i, n, t_first)
        self._first_chunks_done()
        self.cohorts = []
        # a finished session's slot is free at once (contiguous runs released with one call)
        ended_slots.sort()
        i = 0
        while i < len(ended_slots):
            j = i
            while j + 1 < len(ended_slots) and ended_slots[j + 1] == ended_slots[j] + 1:
                j += 1
            s0, cnt = ended_slots[i], j - i + 1
            eng.release(s0, cnt)
            self._give(s0, cnt)
            for s in ended_slots[i : j + 1]:
                self.rows_live.pop(s, None)
            i = j + 1

    def _continue(self, job, slot: int, frames: int, t_first: float) -> None:
        """A session that goes on after its first chunk: from now on an independent row of this loop (NurseryTTS hands it on instead)."""
        self.rows_live[slot] = _Row(job, slot, frames, t_first)

    def _first_chunks_done(self) -> None:
        pass

    # ---- adoption of sessions whose first chunk ran in a nursery loop ---------------------------------
    def adopt(self, grads: Sequence[_Graduates], nursery: "NurseryTTS") -> list:
        """Move graduated sessions into free slots of THIS loop (s2s_tts_migrate on this stream, behind the nursery's event); returns what
        did not fit.  The nursery gets its slots back together with an event that marks the end of the copies."""
        left = []
        for g in grads:
            jobs, src = list(g.jobs), g.slot0
            waited = False
            while jobs and self.free:
                s, k = self._take(len(jobs))
                if not waited:
                    _stream_wait(g.event)
                    waited = True
                off = src - g.slot0
                if g.x is not None and getattr(g.x, "is_cuda", False):
                    g.x.record_stream(torch.cuda.current_stream())      # (allocated on the nursery's stream, read on this one)
                xs = None if g.x is None else g.x[off : off + k]
                self.eng.migrate_from(nursery.eng, src, s, k, g.n_pos, g.frames, xs.contiguous() if hasattr(xs, "contiguous") else xs)
                ev = torch.cuda.Event()
                ev.record()
                for i, job in enumerate(jobs[:k]):
                    self.rows_live[s + i] = _Row(job, s + i, g.frames, g.t_first)
                nursery.give_back(src, k, ev)
                jobs, src = jobs[k:], src + k
                self.stats["admitted"] += k
            if jobs:
                left.append(_Graduates(jobs, src, g.frames, g.n_pos, g.t_first, g.event, None if g.x is None else g.x[src - g.slot0 :]))
        self.stats["max_live"] = max(self.stats["max_live"], self.live)
        return left

    FRAME_S = 0.08                   # one codec frame = 80 ms of audio (12.5 Hz)
    HUNGRY_FRAMES = 32               # a session whose consumer has not reported audible audio yet is served eagerly for this many frames

    def min_lead_s(self, now: float) -> float:
        """Smallest audio lead over the live sessions: seconds of already delivered audio that playback has not reached yet
        (first chunk handed out at t_first, frames * 80 ms delivered).  A cohort still waiting for its first chunk has lead -inf."""
        if self.cohorts:
            return float("-inf")
        lead = float("inf")
        for r in self.rows_live.values():
            st = getattr(r.job, "started", None)
            if st is not None and not st.is_set() and r.frames < self.HUNGRY_FRAMES:
                return float("-inf")         # (leading silence is trimmed by the handler: nothing audible has left yet)
            lead = min(lead, r.t_first + r.frames * self.FRAME_S - now)
        return lead

    def drain(self) -> Iterator[tuple[Any, np.ndarray | None]]:
        while self.cohorts or self.rows_live:
            yield from self.step()

    def abort(self, err: BaseException | None = None) -> list:
        """Drop every live session (engine failure / shutdown); returns their jobs."""
        jobs = [j for c in self.cohorts for j in c.jobs] + [r.job for r in self.rows_live.values()]
        try:
            self.eng.release(0, self.cap)
        except Exception:  # noqa: BLE001
            pass
        self.cohorts = []
        self.rows_live = {}
        self.free = [(0, self.cap)]
        return jobs


class NurseryTTS(ContinuousTTS):
    """The small, EAGER frame loop that produces every session's first chunk (time-to-first-audio), on its own engine replica / stream /
    thread.  Sessions are admitted, run `chunk_size` frames at the smallest row count that holds them, get their first chunk decoded and
    delivered - and are then handed to the wide, paced ContinuousTTS (`graduates`), which adopts them with s2s_tts_migrate.  A slot is
    reusable once the wide loop has returned it (`give_back`: the copies out of it have completed)."""

    def __init__(self, engine, chunk_size: int = 8, max_cohorts: int = 8):
        super().__init__(engine, chunk_size, max_cohorts)
        self.graduates: list[_Graduates] = []
        self._run: list[tuple[Any, int, float]] = []          # (job, slot, t_first) of the round being delivered
        self._returned: list[tuple[int, int, Any]] = []
        self._ret_lock = threading.Lock()

    @property
    def live(self) -> int:
        return sum(len(c.jobs) for c in self.cohorts)

    def give_back(self, slot0: int, n: int, event: Any) -> None:
        """Called by the adopting loop's thread: slots [slot0, slot0 + n) have been copied out (complete when `event` fires)."""
        with self._ret_lock:
            self._returned.append((slot0, n, event))

    def reclaim(self) -> None:
        """Nursery thread: make returned slots admissible again (their next prefill is ordered behind the copies by the event)."""
        with self._ret_lock:
            ret, self._returned = self._returned, []
        for slot0, n, ev in ret:
            _stream_wait(ev)
            self.eng.release(slot0, n)
            self._give(slot0, n)

    def _continue(self, job, slot: int, frames: int, t_first: float) -> None:
        self._run.append((job, slot, t_first))

    def _first_chunks_done(self) -> None:
        """Group the survivors of this round into contiguous slot runs and publish them behind ONE event."""
        if not self._run:
            return
        self._run.sort(key=lambda t: t[1])
        runs, i = [], 0
        while i < len(self._run):
            j = i
            while j + 1 < len(self._run) and self._run[j + 1][1] == self._run[j][1] + 1:
                j += 1
            jobs = [t[0] for t in self._run[i : j + 1]]
            lens = [int(getattr(getattr(job.prompt, "embeds", None), "shape", (self.eng.max_prompt,))[0]) for job in jobs]
            s0 = self._run[i][1]
            # snapshot the next input rows, THEN park the slots: this loop's later rounds step over them (a parked slot keeps its cache rows,
            # counters, codes and codec state; only its row of x is recomputed by every frame)
            x = self.eng.read_x(s0, len(jobs)) if hasattr(self.eng, "read_x") else None
            self.eng.release(s0, len(jobs))
            runs.append((jobs, s0, min(max(lens), self.eng.max_prompt) + self.chunk, self._run[i][2], x))
            i = j + 1
        ev = torch.cuda.Event()
        ev.record()
        for jobs, s0, n_pos, t_first, x in runs:
            self.graduates.append(_Graduates(jobs, s0, self.chunk, n_pos, t_first, ev, x))
        self._run = []

    def take_graduates(self) -> list:
        out, self.graduates = self.graduates, []
        return out
