"""Row a6 on the GPU: the live-partial path of ``--stt mi355x-parakeet`` - ``PooledParakeetModel.transcribe(timestamps=True)`` and the
``SentenceWindow`` policy (reference STT/smart_progressive_streaming.py:69-157) - on the REAL float32 engine behind a SessionPool,
against the same adapter code driven by the CPU oracle (oracle/parakeet.py: log-mel -> FastConformer -> TDT greedy on that window alone).

The float32 engine is step-exact against the oracle, so everything derived from its steps must be IDENTICAL: the window's text, every
``timestamp["segment"]`` entry (text, start, end = encoder frame x 80 ms) and - over a turn growing in 0.5 s increments past the window
cap - every (fixed, active) pair, including the calls where sentences are fixed and the shortened window is transcribed again.  Several
sessions run their partials concurrently so that the pool batches windows of different lengths into one encoder pass (what a tick of
live sessions looks like)."""

import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import logmel as OL
from oracle import parakeet as OP
from speech_to_speech_amd.handlers.parakeet import PooledParakeetModel, SentenceWindow
from speech_to_speech_amd.frontend import pack_pcm
from speech_to_speech_amd.parakeet import ParakeetEngine
from speech_to_speech_amd.session_pool import SessionPool
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

pytestmark = pytest.mark.gpu


class _Codec:
    """token id -> word; ids divisible by 5 end a sentence (so that random-weight transcripts have sentence structure)."""

    def decode(self, ids):
        return " ".join(f"w{i}" + ("." if i % 5 == 0 else "") for i in ids)


class _OracleModel:
    """The same ``model`` surface (transcribe(audio, timestamps)) computed by the CPU oracle on that window alone."""

    def __init__(self, dims, sd, tok):
        self.dims, self.sd, self.tok, self.calls = dims, sd, tok, 0

    def steps(self, audio):
        a = np.asarray(audio, dtype=np.float32)
        if len(a) < 1280:
            a = np.pad(a, (0, 1280 - len(a)))
        feats = torch.from_numpy(OL.parakeet_logmel(a, 128, t_total=len(a) // 160 + 1)[0].astype(np.float32))[None]
        with torch.no_grad():
            enc, elen = OP.encode(self.sd, feats, torch.tensor([len(a) // 160]), self.dims.n_heads, self.dims.layers)
            toks, durs, frames = OP.tdt_greedy(self.sd, enc, elen, self.dims.vocab, self.dims.blank)
        keep = [k for k, t in enumerate(toks[0]) if t != self.dims.blank]
        return [toks[0][k] for k in keep], [frames[0][k] * 0.08 for k in keep], [(frames[0][k] + max(durs[0][k], 1)) * 0.08 for k in keep]

    def transcribe(self, audio, timestamps=False):
        self.calls += 1
        ids, starts, ends = self.steps(audio)
        text = self.tok.decode(ids).strip()
        if not timestamps:
            return SimpleNamespace(text=text)
        segs, cur = [], []                      # (restated here, not imported: a segment closes after a sentence-final token, the tail is one more)
        for j, tid in enumerate(ids):
            cur.append(j)
            if self.tok.decode([tid]).strip().endswith("."):
                segs.append(cur)
                cur = []
        if cur:
            segs.append(cur)
        return SimpleNamespace(text=text, timestamp={"segment": [{"segment": self.tok.decode([ids[j] for j in g]).strip(), "start": float(starts[g[0]]),
                                                                  "end": float(ends[g[-1]])} for g in segs]})


@pytest.fixture(scope="module")
def rig():
    dims = ParakeetDims.test_tiny()
    sd = parakeet_state_dict(dims, seed=12, gain=1.5)
    pool = SessionPool("cuda:0", max_batch=4, batch_window_ms=20.0)
    pool.ensure_engine("parakeet", lambda: ParakeetEngine(dims, sd, dtype=torch.float32, max_batch=4, max_seconds=12.0))
    yield dims, sd, pool, _Codec()
    pool.shutdown()


def test_timestamped_transcribe_matches_oracle_segments(rig):
    dims, sd, pool, tok = rig
    model, oracle = PooledParakeetModel(pool, tok), _OracleModel(dims, sd, tok)
    for i, n in enumerate((48000, 30500, 96000, 8000)):
        audio = synth_audio(n, 900 + i)
        got, want = model.transcribe(audio, timestamps=True), oracle.transcribe(audio, timestamps=True)
        segs = got.timestamp["segment"]
        print(f"{n / 16000:.2f} s: {len(want.text.split())} words, {len(segs)} segments, ends {[round(s['end'], 2) for s in segs]}")
        assert got.text == want.text
        assert [s["segment"] for s in segs] == [s["segment"] for s in want.timestamp["segment"]]
        assert [(round(s["start"], 4), round(s["end"], 4)) for s in segs] == [(round(s["start"], 4), round(s["end"], 4)) for s in want.timestamp["segment"]]
        assert model.transcribe(audio).text == want.text


def test_cached_features_are_bit_identical_to_the_whole_window(rig):
    """Row f3 (the part that IS reusable): s2s_logmel_parakeet_cached on a window that grows by 0.5 s (and once by an odd amount, once not at
    all) gives the features of s2s_logmel_parakeet on the whole window BIT FOR BIT - float32 and bf16 outputs, also as a row of a larger
    batch - while uploading only the new samples and computing only the 64-frame tiles the new samples touch; a new window start or audio that
    is not an extension of the last partial restarts the stream."""
    from speech_to_speech_amd.parakeet import ProgressiveStream

    dims, sd, pool, tok = rig
    eng = pool._engines["parakeet"]
    full = synth_audio(11 * 16000, 980)
    for dtype in (torch.float32, torch.bfloat16):
        st = ProgressiveStream()
        lens = list(range(8000, 11 * 16000 + 1, 8000))
        lens[5] += 1234                                   # an odd increment
        lens.insert(9, lens[8])                           # a partial without new samples
        for n in lens:
            T = max(8, n // 160 + 1) + (3 if n % 16000 == 0 else 0)       # (as a row of a batch whose longest item is longer)
            row = torch.empty((1, T, dims.n_mels), dtype=dtype, device="cuda:0")
            nf = torch.empty((1,), dtype=torch.int32, device="cuda:0")
            eng.stream_features(st, full[:n], 0, T, row, nf)
            pcm, ns = pack_pcm([full[:n]], eng.device)
            want, wnf = eng.frontend.parakeet(pcm, ns, T, out_dtype=dtype)
            assert int(nf[0]) == int(wnf[0]) == n // 160
            assert torch.equal(row, want), f"{dtype} n = {n}: cached features differ from the whole-window features"
        s_ = st.stats
        print(f"{dtype}: {s_['partials']} partials, uploaded {s_['samples_uploaded']} of {s_['samples_total']} samples ({s_['samples_uploaded'] / s_['samples_total']:.1%}), "
              f"computed {s_['frames_computed']} of {s_['frames_total']} frames ({s_['frames_computed'] / s_['frames_total']:.1%}), restarts {s_['restarts']}")
        assert s_["restarts"] == 1 and s_["samples_uploaded"] == lens[-1] and s_["frames_computed"] < 0.5 * s_["frames_total"]
        # not an extension (another signal / a moved window): the stream restarts instead of reusing anything
        other = synth_audio(40000, 981)
        row = torch.empty((1, 251, dims.n_mels), dtype=dtype, device="cuda:0")
        nf = torch.empty((1,), dtype=torch.int32, device="cuda:0")
        eng.stream_features(st, other, 0, 251, row, nf)
        pcm, ns = pack_pcm([other], eng.device)
        assert torch.equal(row, eng.frontend.parakeet(pcm, ns, 251, out_dtype=dtype)[0]) and st.stats["restarts"] == 2
        eng.stream_features(st, full[16000:64000], 16000, 301, torch.empty((1, 301, dims.n_mels), dtype=dtype, device="cuda:0"), nf)
        assert st.stats["restarts"] == 3
        # a buffer that changed BEHIND its tail (re-based / enhanced turn, or a tail of digital silence that looks the same) is
        # not an extension either - the whole uploaded prefix is checksummed, the stream restarts and the features are those of the new buffer
        quiet = full[:48000].copy()
        quiet[-4000:] = 0.0
        st2 = ProgressiveStream()
        row = torch.empty((1, 351, dims.n_mels), dtype=dtype, device="cuda:0")
        eng.stream_features(st2, quiet, 0, 351, row, nf)
        edited = np.concatenate([quiet, np.zeros(8000, dtype=np.float32)])
        edited[1000:3000] *= 0.5                           # same length prefix, same last 64 samples, different content before them
        row = torch.empty((1, 351, dims.n_mels), dtype=dtype, device="cuda:0")
        eng.stream_features(st2, edited, 0, 351, row, nf)
        pcm, ns = pack_pcm([edited], eng.device)
        assert st2.stats["restarts"] == 2 and torch.equal(row, eng.frontend.parakeet(pcm, ns, 351, out_dtype=dtype)[0])


def test_a_tick_of_growing_windows_in_one_call_is_bit_identical_per_window(rig):
    """s2s_logmel_parakeet_cached_rows (ABI v5): the partial windows of a tick - different lengths, different growth, one of them restarted with
    another signal, one without new samples - in ONE call with ONE upload of the new samples: every row equals s2s_logmel_parakeet on that window
    alone bit for bit (float32 and bf16), and the streams' counters say that only the new samples crossed PCIe."""
    from speech_to_speech_amd.parakeet import ProgressiveStream

    dims, sd, pool, tok = rig
    eng = pool._engines["parakeet"]
    B = 4
    turns = [synth_audio(10 * 16000, 700 + i) for i in range(B)]
    for dtype in (torch.float32, torch.bfloat16):
        streams = [ProgressiveStream() for _ in range(B)]
        lens = [4000, 12000, 30000, 8000]
        for tick in range(7):
            grow = [8000, 8000 + 777 * (tick % 2), 8000, 0 if tick == 3 else 8000]
            lens = [min(len(t), n + g) for t, n, g in zip(turns, lens, grow)] if tick else lens
            wins = [t[:n] for t, n in zip(turns, lens)]
            keys = [0] * B
            if tick == 4:                                  # row 2: the window moved (sentences were fixed) - another start, other samples
                turns[2] = synth_audio(10 * 16000, 790)
                lens[2] = 20000
                wins[2], keys[2] = turns[2][:20000], 5
            elif tick > 4:
                keys[2] = 5
            T = max(8, max(lens) // 160 + 1)
            out = torch.empty((B, T, dims.n_mels), dtype=dtype, device="cuda:0")
            nf = torch.empty((B,), dtype=torch.int32, device="cuda:0")
            eng.stream_features_rows(streams, wins, keys, T, out, nf)
            pcm, ns = pack_pcm(wins, eng.device)
            want, wnf = eng.frontend.parakeet(pcm, ns, T, out_dtype=dtype)
            assert torch.equal(nf, wnf)
            for b in range(B):
                assert torch.equal(out[b], want[b]), f"{dtype} tick {tick} row {b}: batched cached features differ from the whole-window features"
        st = [s.stats for s in streams]
        print(f"{dtype}: uploaded {[round(s_['samples_uploaded'] / s_['samples_total'], 3) for s_ in st]} of the samples, restarts {[s_['restarts'] for s_ in st]}")
        assert [s_["restarts"] for s_ in st] == [1, 1, 2, 1] and all(s_["samples_uploaded"] < 0.5 * s_["samples_total"] for s_ in st)
        assert st[0]["samples_uploaded"] == lens[0]


@pytest.mark.parametrize("incremental", [False, True])
def test_growing_turns_fix_sentences_like_the_oracle_driven_streamer(rig, incremental):
    """(incremental = True: every window is a ProgressiveStream - only new samples uploaded, only new log-mel tiles computed, same results.)
    Three sessions speak at once; each turn grows by 0.5 s per tick to 9 s with a 4 s window cap and a 1 s sentence buffer (the
    reference's 15 s / 2 s scaled to a test-sized turn).  Every tick's (fixed, active) of every session equals the oracle-driven
    SentenceWindow's, sentences get fixed along the way, and the pool served windows of different sessions in shared batches."""
    dims, sd, pool, tok = rig
    n_sessions, total, step = 3, 9 * 16000, 8000
    turns = [synth_audio(total, 950 + i) for i in range(n_sessions)]
    want = []
    for a in turns:
        w = SentenceWindow(_OracleModel(dims, sd, tok), max_window_s=4.0, sentence_buffer_s=1.0)
        want.append([w.update(a[:n]) for n in range(step, total + 1, step)])
        assert w.fixed, "the turn never fixed a sentence: the test would not cover the window move"
    before = dict(pool.stats)
    got = [[] for _ in turns]
    wins = [None] * n_sessions
    barrier = threading.Barrier(n_sessions)

    def session(i):
        w = SentenceWindow(PooledParakeetModel(pool, tok), max_window_s=4.0, sentence_buffer_s=1.0, incremental=incremental)
        wins[i] = w
        for n in range(step, total + 1, step):
            barrier.wait(timeout=120)           # a tick: every live session sends its partial now
            got[i].append(w.update(turns[i][:n]))

    ts = [threading.Thread(target=session, args=(i,)) for i in range(n_sessions)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    for i in range(n_sessions):
        assert got[i] == want[i], f"session {i}: first differing tick {next(k for k, (g, w_) in enumerate(zip(got[i], want[i])) if g != w_)}"
    n_req = pool.stats["stt_requests"] - before.get("stt_requests", 0) if "stt_requests" in pool.stats else None
    n_bat = pool.stats["stt_batches"] - before.get("stt_batches", 0)
    print(f"{n_sessions} sessions x {total // step} ticks: {n_req} window transcriptions in {n_bat} encoder passes (largest batch {pool.stats['max_stt_batch']}); "
          f"fixed sentences per session {[len(w[-1][0].split('.')) - 1 for w in want]}")
    assert pool.stats["max_stt_batch"] >= 2 and n_bat < n_sessions * (total // step)       # partials of a tick share encoder passes
    if incremental:
        for w in wins:
            s_ = w.stream.stats
            print(f"stream: {s_['partials']} partials in {s_['restarts']} windows, uploaded {s_['samples_uploaded'] / s_['samples_total']:.1%} of the samples, "
                  f"computed {s_['frames_computed'] / s_['frames_total']:.1%} of the log-mel frames a whole-window recompute handles")
            assert s_["partials"] >= total // step and s_["restarts"] >= 2 and s_["samples_uploaded"] < 0.6 * s_["samples_total"]


def test_three_stt_workers_on_replicas_answer_like_the_single_engine(rig):
    """SessionPool(stt_workers = 3) / ensure_stt_workers: finals and live partials of many sessions spread over three workers - each a thread, a HIP
    stream and a ParakeetEngine.replica() that shares the first engine's weights - give exactly the transcripts the single float32 engine gives
    (which is step-exact against the oracle, above); every worker served something; the replicas own no second copy of the uploaded weights."""
    dims, sd, pool0, tok = rig
    eng0 = pool0._engines["parakeet"]
    audios = [synth_audio(16000 + 4000 * (i % 7), 300 + i) for i in range(18)]
    want = [eng0.transcribe_batch([a], [None])[0]["ids"] for a in audios]
    pool = SessionPool("cuda:0", max_batch=4, batch_window_ms=2.0)
    try:
        pool.ensure_engine("parakeet", lambda: ParakeetEngine(dims, sd, dtype=torch.float32, max_batch=4, max_seconds=12.0))
        pool.ensure_stt_workers(3)
        reps = list(pool._stt_extra_reps["parakeet"].values())
        assert len(reps) == 2 and all(r is not pool._engines["parakeet"] for r in reps)
        base = pool._engines["parakeet"]
        assert all(r._weights[k].data_ptr() == base._weights[k].data_ptr() for r in reps for k in base._weights)        # shared, not copied
        served = {}
        for k, r in enumerate([base] + reps):
            orig = r.transcribe_batch

            def counted(a, o, orig=orig, k=k):
                served[k] = served.get(k, 0) + len(a)
                return orig(a, o)

            r.transcribe_batch = counted
        got = [None] * len(audios)

        def one(i):
            got[i] = pool.transcribe("parakeet", audios[i], live=bool(i % 2))["ids"]

        for rnd in range(3):                             # three bursts, so that every worker gets its share
            ts = [threading.Thread(target=one, args=(i,)) for i in range(len(audios))]
            [t.start() for t in ts]
            [t.join(timeout=120) for t in ts]
            assert got == want, f"burst {rnd}: a worker's transcript differs from the single engine's"
        print(f"requests served per worker: {served}")
        assert len(served) == 3 and sum(served.values()) == 3 * len(audios)
    finally:
        pool.shutdown()
