"""Boundary-conformance tests (CPU, fake engines): the MI355X handlers honour the reference's handler contract
(tests/test_backend_registry.py, test_stt_stale_filter.py, test_qwen3_tts_handler_backend.py:505-735,
915-964, test_tts_input_coalescing.py) and the session pool batches concurrent pipeline units."""

import importlib
import sys
import threading
import time
from queue import Queue
from threading import Event

import numpy as np
import pytest

from speech_to_speech_amd.boundary import (AUDIO_RESPONSE_DONE, PIPELINE_END, SESSION_END, AudioOutput, CancelScope, EndOfResponse,
                                           PartialTranscription, Transcription, TTSInput, VADAudio)
from speech_to_speech_amd.boundary import registry
from speech_to_speech_amd.handlers.qwen3_tts import MI355XQwen3TTSHandler, estimate_max_new_tokens
from speech_to_speech_amd.session_pool import SessionPool
from speech_to_speech_amd.text import IdTextCodec, WhisperVocab


class FakeSTTEngine:
    def __init__(self):
        self.batches = []

    def transcribe_batch(self, audios, options):
        self.batches.append(len(audios))
        time.sleep(0.01)
        return [{"ids": [len(a) % 97, 5], "language": "de", "starts_s": [0.0, 0.08]} for a in audios]

    def close(self):
        pass


class FakeTTSEngine:
    max_prompt = 64

    class text_frontend:  # noqa: N801
        @staticmethod
        def embed(text, n):
            return np.zeros((min(n, len(text) + 2), 8), np.float32)

    def stream(self, prompts, max_new_tokens, chunk_size=8, ctx_frames=25, min_frames=0):
        for k in range(3):
            time.sleep(0.005)
            yield [np.full(700, 3000 + k, np.int16) for _ in prompts]

    def close(self):
        pass


@pytest.fixture
def pool():
    p = SessionPool("fake:0", max_batch=8, batch_window_ms=20.0)
    p._engines.update(whisper=FakeSTTEngine(), parakeet=FakeSTTEngine(), tts=FakeTTSEngine())
    yield p
    p.shutdown()


def _whisper(pool, language="en"):
    from speech_to_speech_amd.handlers.whisper import MI355XWhisperSTTHandler

    h = object.__new__(MI355XWhisperSTTHandler)
    h.stop_event, h.queue_in, h.queue_out = Event(), Queue(), Queue()
    h.gen_kwargs, h.start_language, h.last_language = {}, language, None if language == "auto" else language
    h.vocab, h.tokenizer, h.pool, h._times = WhisperVocab(), IdTextCodec(), pool, []
    return h


def test_stt_final_yields_exactly_one_transcription_with_turn_fields(pool):
    h = _whisper(pool)
    va = VADAudio(audio=np.zeros(1600, np.float32), mode="final", turn_id="t1", turn_revision=3)
    out = list(h.process(va))
    assert len(out) == 1 and isinstance(out[0], Transcription)
    assert (out[0].turn_id, out[0].turn_revision, out[0].speech_stopped_at_s) == ("t1", 3, va.created_at_s)
    assert out[0].language_code == "en" and out[0].text == "t48 t5"


def test_stt_auto_language_reports_detected_code_with_suffix_and_partial_is_tagged(pool):
    h = _whisper(pool, language="auto")
    out = list(h.process(VADAudio(audio=np.zeros(800, np.float32), mode="final")))
    assert out[0].language_code == "de-auto"
    part = list(h.process(VADAudio(audio=np.zeros(800, np.float32), mode="progressive", turn_id="a", turn_revision=1)))
    assert len(part) == 1 and isinstance(part[0], PartialTranscription) and part[0].turn_id == "a"


def test_parakeet_failure_degrades_to_empty_transcript(pool):
    from speech_to_speech_amd.handlers.parakeet import MI355XParakeetSTTHandler

    class Broken(FakeSTTEngine):
        def transcribe_batch(self, audios, options):
            raise RuntimeError("device lost")

    pool._engines["parakeet"] = Broken()
    h = object.__new__(MI355XParakeetSTTHandler)
    h.language, h.tokenizer, h.pool = "en", IdTextCodec(), pool
    out = list(h.process(VADAudio(audio=np.zeros(1600, np.float32), mode="final", turn_id="x", turn_revision=0)))
    assert len(out) == 1 and out[0].text == "" and out[0].turn_id == "x"


def test_pool_batches_concurrent_pipeline_units(pool):
    h = [_whisper(pool) for _ in range(6)]
    res = [None] * 6

    def run(i):
        res[i] = list(h[i].process(VADAudio(audio=np.zeros(100 + i, np.float32), mode="final", turn_id=str(i), turn_revision=0)))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(r and r[0].turn_id == str(i) for i, r in enumerate(res))
    assert max(pool.engine("whisper").batches) >= 2 and sum(pool.engine("whisper").batches) == 6   # batched, nothing lost
    assert [r[0].text for r in res] == [f"t{(100 + i) % 97} t5" for i in range(6)]                  # no cross-session leak


def _tts(pool, cancel_scope=None):
    h = object.__new__(MI355XQwen3TTSHandler)
    h.stop_event, h.queue_in, h.queue_out, h._times = Event(), Queue(), Queue(), []
    h.should_listen = Event()
    h.cancel_scope, h.speculative_turns = cancel_scope, None
    h.streaming_chunk_size, h.max_new_tokens, h.blocksize, h.pool = 8, 1536, 512, pool
    return h


def test_tts_blocks_are_int16_512_and_end_of_response_maps_to_sentinel(pool):
    h = _tts(pool)
    blocks = list(h.process(TTSInput(text="Hello there.")))
    assert len(blocks) == 5 and all(b.dtype == np.int16 and b.shape == (512,) for b in blocks)   # 2100 samples -> 4 full + padded tail
    assert int(blocks[-1][-1]) == 0 and not h.should_listen.is_set()
    assert list(h.process(EndOfResponse())) == [AUDIO_RESPONSE_DONE]


def test_tts_cancel_generation_stops_between_chunks(pool):
    cs = CancelScope()
    h = _tts(pool, cs)
    out = []
    for b in h.process(TTSInput(text="Hello there.", cancel_generation=cs.generation)):
        out.append(b)
        cs.cancel()
    assert 1 <= len(out) <= 2     # stops at the next chunk boundary


def test_tts_coalesces_queued_inputs_of_same_response(pool):
    h = _tts(pool)
    first = TTSInput(text="One.", response_key="r1", cancel_generation=0, turn_id="t")
    h.queue_in.put(TTSInput(text="Two.", response_key="r1", cancel_generation=0, turn_id="t"))
    h.queue_in.put(TTSInput(text="Other.", response_key="r2", cancel_generation=0, turn_id="t"))
    text, _ = h._coalesce_pending_tts_input(first)
    assert text == "One. Two." and h.queue_in.qsize() == 1


def test_tts_coalescing_follows_the_reference_break_rules(pool):
    """TTS/qwen3_tts_handler.py:751-810: a different turn_revision or language code ends the merge, EndOfResponse /
    SESSION_END are never crossed, text-only AssistantOutputEvents of the same response are forwarded before the audio."""
    from speech_to_speech_amd.boundary import AssistantOutputEvent, AssistantTextPart

    h = _tts(pool)
    mk = lambda text, **kw: TTSInput(text=text, response_key="r", cancel_generation=0, turn_id="t", turn_revision=1, **kw)   # noqa: E731
    first = mk("One.", language_code="en")
    ev = AssistantOutputEvent(parts=[AssistantTextPart(text="Two.")], text="Two.", response_key="r", cancel_generation=0, turn_id="t", turn_revision=1)
    h.queue_in.put(ev)
    h.queue_in.put(mk("Two."))
    h.queue_in.put(mk("Drei.", language_code="de"))           # language changes: stop here
    text, lang = h._coalesce_pending_tts_input(first)
    assert (text, lang) == ("One. Two.", "en") and h.queue_in.qsize() == 1
    assert h.queue_out.get_nowait() is ev                      # forwarded ahead of the synthesis
    h2 = _tts(pool)
    h2.queue_in.put(TTSInput(text="Newer.", response_key="r", cancel_generation=0, turn_id="t", turn_revision=2))
    assert h2._coalesce_pending_tts_input(mk("One."))[0] == "One." and h2.queue_in.qsize() == 1      # other revision
    h3 = _tts(pool)
    h3.queue_in.put(EndOfResponse(turn_id="t", turn_revision=1))
    h3.queue_in.put(mk("After."))
    assert h3._coalesce_pending_tts_input(mk("One."))[0] == "One." and h3.queue_in.qsize() == 2      # never crosses EndOfResponse
    h4 = _tts(pool)
    h4.queue_in.put(SESSION_END)
    h4.queue_in.put(mk("After."))
    assert h4._coalesce_pending_tts_input(mk("One."))[0] == "One." and h4.queue_in.qsize() == 2


def _reference_tracker():
    """The reference's own SpeculativeTurnTracker (pipeline/speculative_turns.py has no third-party imports and loads
    standalone); skipped where /root/reference is not mounted (the GPU box)."""
    import importlib.util
    import os

    path = "/root/reference/src/speech_to_speech/pipeline/speculative_turns.py"
    if not os.path.exists(path):
        pytest.skip("reference checkout not available")
    spec = importlib.util.spec_from_file_location("_ref_speculative_turns", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod.SpeculativeTurnTracker()


def test_tts_emits_audio_with_the_reference_speculative_turn_tracker(pool):
    """commit() of the real tracker returns None: it must never be used as a filter (round-1 bug: every TTSInput with a
    turn id was dropped, so the registered backend never produced audio in the real pipeline)."""
    tracker = _reference_tracker()
    h = _tts(pool)
    h.speculative_turns = tracker
    rev = 1
    tracker.observe("turn-1", rev)
    blocks = list(h.process(TTSInput(text="Hello there.", turn_id="turn-1", turn_revision=rev)))
    assert len(blocks) == 5
    assert tracker.is_committed("turn-1", rev)


class _StaleTracker:
    """Every revision is stale (is_latest_after_reopen_grace -> False); commit must not be reached."""

    def is_latest_after_reopen_grace(self, turn_id, revision):
        return False

    def commit(self, turn_id, revision):
        raise AssertionError("stale input was committed")


def test_tts_stale_turn_handling_matches_the_reference(pool):
    h = _tts(pool)
    h.speculative_turns = _StaleTracker()
    assert list(h.process(TTSInput(text="Hello.", turn_id="t", turn_revision=1))) == []                  # dropped, no audio
    assert list(h.process(EndOfResponse(turn_id="t", turn_revision=1))) == []                           # stale and no response key: nothing
    eor = EndOfResponse(turn_id="t", turn_revision=1, response_key="k")
    assert list(h.process(eor)) == [AUDIO_RESPONSE_DONE] and eor.cleanup_only is True                   # cleanup-only completion


def test_run_loop_forwards_pipeline_events_in_order(pool):
    from speech_to_speech_amd.boundary import AssistantOutputEvent

    h = _tts(pool)
    t = threading.Thread(target=h.run)
    t.start()
    ev = AssistantOutputEvent(text="tool", response_key="k", cancel_generation=1)
    h.queue_in.put(ev)
    h.queue_in.put(PIPELINE_END)
    t.join(timeout=10)
    assert h.queue_out.get_nowait() is ev and h.queue_out.get_nowait() == PIPELINE_END


def test_run_loop_wraps_audio_and_forwards_session_end(pool):
    h = _tts(pool)
    t = threading.Thread(target=h.run)
    t.start()
    h.queue_in.put(TTSInput(text="Hi.", response_key="k", cancel_generation=7))
    h.queue_in.put(SESSION_END)
    h.queue_in.put(PIPELINE_END)
    t.join(timeout=10)
    items = []
    while not h.queue_out.empty():
        items.append(h.queue_out.get())
    audio = [i for i in items if isinstance(i, AudioOutput)]
    assert audio and all(a.cancel_generation == 7 and a.response_key == "k" for a in audio)
    assert SESSION_END in items and items[-1] == PIPELINE_END


def test_token_budget_known_answers():
    # reference tests/test_qwen3_tts_handler_backend.py:915-964
    assert estimate_max_new_tokens("Hello there.") == 360
    long_text = " ".join(["This is a deliberately long sentence for the Qwen3 TTS budget estimator."] * 12)
    assert 360 < estimate_max_new_tokens(long_text) <= 1536 and estimate_max_new_tokens(long_text) % 8 == 0
    assert estimate_max_new_tokens("我懂，心情不好时会让人特别疲惫。") == 360
    cjk = ("上海是一座充满活力的现代化大都市，既有繁华的金融中心和摩天大楼，也有老城厢的弄堂风情"
           "和江南水乡的韵味。这里交通便利，餐饮选择丰富，从精致西餐到地道小馆应有尽有。同时，上海"
           "还是文化与创新的交汇点，艺术展览、科技展会和国际活动频繁。如果你喜欢快节奏的生活和多元"
           "的氛围，上海会是个很吸引人的地方。你想了解哪方面的具体信息呢？")
    assert estimate_max_new_tokens(cjk) == 576
    assert estimate_max_new_tokens(long_text, cap=400) == 400
    assert estimate_max_new_tokens(" ".join(["This is a deliberately long sentence for the Qwen3 TTS budget estimator."] * 30), cap=2400) > 1536


def test_registry_entries_and_lazy_factories():
    names = {s.name: s for s in registry.MI355X_STT_BACKENDS + registry.MI355X_TTS_BACKENDS}
    assert set(names) == {"mi355x-whisper", "mi355x-parakeet", "mi355x-qwen3"}
    stt, tts = {}, {}
    registry.register_mi355x_backends(stt, tts)
    assert set(stt) == {"mi355x-whisper", "mi355x-parakeet"} and set(tts) == {"mi355x-qwen3"}
    cfg = names["mi355x-whisper"].normalize(names["mi355x-whisper"].config_type(stt_gen_max_new_tokens=64))
    assert cfg["model_name"] and cfg["gen_kwargs"]["max_new_tokens"] == 64 and "stt_model_name" not in cfg
    with pytest.raises(TypeError):
        names["mi355x-qwen3"].normalize(object())
    # the on-device sampler's options reach the handler as gen_kwargs (TTSEngine.configure_sampling)
    tcfg = names["mi355x-qwen3"].normalize(names["mi355x-qwen3"].config_type(qwen3_tts_gen_do_sample=True, qwen3_tts_gen_top_k=20))
    assert tcfg["gen_kwargs"]["do_sample"] is True and tcfg["gen_kwargs"]["top_k"] == 20 and tcfg["gen_kwargs"]["temperature"] == 0.9
    assert "gen_do_sample" not in tcfg and tcfg["lanes"] == 3
    # the scheduling knobs bench.py measures with are reachable from the CLI: pacing of the frame loop, the final-STT batch cap
    assert tcfg["lead_s"] == 0.0 and tcfg["admit_wait_ms"] == 0.0 and cfg["batch_cap"] == 0           # defaults: eager, uncapped
    hot = names["mi355x-qwen3"].normalize(names["mi355x-qwen3"].config_type(qwen3_tts_lanes=1, qwen3_tts_max_batch=512, qwen3_tts_lead_s=0.5, qwen3_tts_admit_wait_ms=250.0))
    assert (hot["lanes"], hot["max_batch"], hot["lead_s"], hot["admit_wait_ms"]) == (1, 512, 0.5, 250.0)
    assert names["mi355x-parakeet"].normalize(names["mi355x-parakeet"].config_type(parakeet_tdt_batch_cap=32))["batch_cap"] == 32
    # importing the registry must not import the handler modules or torch-heavy engines
    for m in [k for k in sys.modules if k.startswith("speech_to_speech_amd.handlers") or k.endswith(".whisper") or k.endswith(".tts")]:
        del sys.modules[m]
    importlib.reload(registry)
    assert not any(k.startswith("speech_to_speech_amd.handlers.") for k in sys.modules)


def test_tts_lanes_overlap_turn_batches():
    """Two TTS lanes (thread + stream + engine replica each) serve two turn-batches at the same time; a lane whose engine
    cannot be replicated retires and lane 0 serves everything."""

    class SlowTTS(FakeTTSEngine):
        instances = 0

        def replica(self):
            SlowTTS.instances += 1
            return SlowTTS()

        def stream(self, prompts, max_new_tokens, chunk_size=8, ctx_frames=25, min_frames=0):
            for k in range(4):
                time.sleep(0.05)
                yield [np.full(512, k, np.int16) for _ in prompts]

    p = SessionPool("fake:1", max_batch=1, batch_window_ms=1.0, tts_lanes=2)
    p._engines.update(tts=SlowTTS())
    try:
        t0 = time.perf_counter()
        its = [p.synthesize(object(), 8)[0] for _ in range(2)]
        got = [None, None]

        def drain(i):
            got[i] = [c.copy() for c in its[i]]

        ts = [threading.Thread(target=drain, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        assert all(len(g) == 4 for g in got)
        assert p.stats["tts_lanes_used"] == 2 and SlowTTS.instances == 1 and p.stats["tts_batches"] == 2
        assert dt < 0.35, f"two 0.2 s batches on two lanes took {dt:.2f} s (serial would be 0.4 s)"
    finally:
        p.shutdown()


# ---- live partials: timestamped transcribe + sentence-aware window ---------------------------------------
class _ScriptedPool:
    """Stand-in for the session pool: 'transcribes' a window by reading a script of (token id, start_s, end_s) of the whole
    utterance and returning the tokens that start inside the window (times relative to the window start)."""

    max_batch = 8

    def __init__(self, script, total_samples):
        self.script, self.total, self.calls = script, total_samples, []

    def transcribe(self, model, audio, **kw):
        off_s = float(audio[0])                                     # the test audio carries its own offset in seconds
        dur = len(audio) / 16000.0
        self.calls.append((round(off_s, 3), round(dur, 3)))
        toks = [(t, s - off_s, e - off_s) for t, s, e in self.script if off_s <= s < off_s + dur]
        return {"ids": [t for t, _, _ in toks], "starts_s": [s for _, s, _ in toks], "ends_s": [e for _, _, e in toks]}

    def try_transcribe(self, model, audio, max_queue, **kw):
        return self.transcribe(model, audio)


class _PunctCodec:
    """ids 0..9 -> words, ids >= 100 -> the word with a full stop."""

    def decode(self, ids):
        return " ".join(f"w{i % 100}" + ("." if i >= 100 else "") for i in ids)


def _ramp_audio(seconds):
    """audio[i] = i / 16000 so that any slice starts with its own offset in seconds (what _ScriptedPool reads)."""
    return (np.arange(int(seconds * 16000)) / 16000.0).astype(np.float32)


def _script():
    # one word per second for 24 s, sentence ends at 3.x, 8.x, 13.x, 18.x s
    return [(i + (100 if i in (3, 8, 13, 18) else 0), i + 0.2, i + 0.6) for i in range(24)]


def test_timestamped_transcribe_builds_sentence_segments():
    from speech_to_speech_amd.handlers.parakeet import PooledParakeetModel

    pool = _ScriptedPool(_script(), 24 * 16000)
    model = PooledParakeetModel(pool, _PunctCodec())
    res = model.transcribe(_ramp_audio(10.0), timestamps=True)
    segs = res.timestamp["segment"]
    assert [s["segment"] for s in segs] == ["w0 w1 w2 w3.", "w4 w5 w6 w7 w8.", "w9"]
    assert [round(s["end"], 2) for s in segs] == [3.6, 8.6, 9.6] and round(segs[1]["start"], 2) == 4.2
    assert res.text == "w0 w1 w2 w3. w4 w5 w6 w7 w8. w9"
    assert model.transcribe(_ramp_audio(2.0)).text == "w0 w1"


def test_incremental_sentence_window_marks_each_window_as_one_stream():
    """SentenceWindow(incremental=True) hands the pool ONE ProgressiveStream per window plus the window's start sample as its key: the key
    changes exactly when sentences are fixed and the window start moves (the engine then restarts the stream), the texts are those of the plain
    window (the stream only changes where the features come from - tests/test_parakeet_progressive_gpu.py checks that on the device)."""
    from speech_to_speech_amd.handlers.parakeet import PooledParakeetModel, SentenceWindow

    class _Recording(_ScriptedPool):
        def transcribe(self, model, audio, **kw):
            self.seen.append((kw.get("stream"), kw.get("stream_key"), len(audio)))
            return super().transcribe(model, audio)

    pool, plain_pool = _Recording(_script(), 24 * 16000), _ScriptedPool(_script(), 24 * 16000)
    pool.seen = []
    win = SentenceWindow(PooledParakeetModel(pool, _PunctCodec()), incremental=True)
    plain = SentenceWindow(PooledParakeetModel(plain_pool, _PunctCodec()))
    full = _ramp_audio(24.0)
    for n in range(8000, len(full) + 1, 8000):
        assert win.update(full[:n]) == plain.update(full[:n])
    streams = {id(s) for s, _, _ in pool.seen}
    assert len(streams) == 1 and pool.seen[0][0] is win.stream                    # one token object, the engine keeps the device state on it
    keys = [k for _, k, _ in pool.seen]
    assert keys[0] == 0 and keys == sorted(keys) and len(set(keys)) >= 2           # the window start only moves forward, and it did move
    for (_, k0, n0), (_, k1, n1) in zip(pool.seen, pool.seen[1:]):
        if k1 == k0:
            assert n1 > n0                                                         # same window: a longer prefix every time
    assert [c for c in pool.calls] == [c for c in plain_pool.calls]                # same windows transcribed, in the same order


def test_sentence_window_fixes_sentences_like_the_reference_streamer():
    """Growing audio in 0.5 s increments up to 24 s: below 15 s the window is the whole turn; at 15 s the sentences ending
    before (window - 2 s) are fixed and the window restarts at the last fixed end.  When the reference is present
    (this container), its own SmartProgressiveStreamingHandler is driven by the same adapter and must agree call by call."""
    import importlib.util
    import os

    from speech_to_speech_amd.handlers.parakeet import PooledParakeetModel, SentenceWindow

    ref_path = "/root/reference/src/speech_to_speech/STT/smart_progressive_streaming.py"
    ref = None
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location("ref_smart_progressive", ref_path)
        mod = importlib.util.module_from_spec(spec)
        import sys

        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        ref = mod.SmartProgressiveStreamingHandler(PooledParakeetModel(_ScriptedPool(_script(), 24 * 16000), _PunctCodec()))

    pool = _ScriptedPool(_script(), 24 * 16000)
    win = SentenceWindow(PooledParakeetModel(pool, _PunctCodec()))
    full = _ramp_audio(24.0)
    seen_fix = False
    for n in range(8000, len(full) + 1, 8000):
        got = win.update(full[:n])
        assert got is not None
        fixed, active = got
        if ref is not None:
            r = ref.transcribe_incremental(full[:n])
            assert (r.fixed_text, r.active_text) == (fixed, active), f"at {n / 16000:.1f} s"
        if n / 16000 < 15.0:
            assert fixed == "" and active.startswith("w0")
        if fixed:
            seen_fix = True
            assert fixed.startswith("w0 w1 w2 w3.") and not active.startswith("w0")
    assert seen_fix and win.fixed_end_s > 0
    assert win.update(full) is None                                   # no new samples -> nothing to do
    # at 15.0 s: window 15 s, cutoff 13 s -> sentences ending at 3.6 and 8.6 are fixed (13.6 is not), restart at 8.6 s
    assert (0.0, 15.0) in pool.calls and (8.6, 6.4) in pool.calls


def test_warm_tts_lanes_builds_and_warms_every_lane_before_the_first_request():
    warmed = []

    class WarmTTS(FakeTTSEngine):
        def replica(self):
            return WarmTTS()

        def warmup(self, prefill_len=100, frames=8):
            warmed.append(threading.current_thread().name)

    p = SessionPool("fake:2", max_batch=2, batch_window_ms=1.0, tts_lanes=3)
    p._engines.update(tts=WarmTTS())
    try:
        p.warm_tts_lanes(timeout=5.0)
        assert p.stats["tts_lanes_used"] == 3 and len(warmed) == 3 and len(set(warmed)) == 3
        it, _ = p.synthesize(object(), 8)
        assert len(list(it)) == 3
        assert len(warmed) == 3                     # serving does not warm again
    finally:
        p.shutdown()


# ---- stub LLM and OpenAI-compatible server mode ---------------------------------------------------
def test_echo_llm_backend_spec_honours_the_llm_contract():
    from types import SimpleNamespace

    from speech_to_speech_amd.boundary import GenerateResponseRequest, LLMResponseChunk

    llm = {}
    registry.register_mi355x_backends({}, {}, llm)
    spec = llm["echo"]
    assert spec.kind == "llm"
    cs = CancelScope()
    cs.cancel()                                  # generation 1: the stub must capture the CURRENT generation
    ctx = registry.HandlerContext(stop_event=Event(), queue_in=Queue(), queue_out=Queue(), text_output_queue=Queue(), should_listen=Event(), cancel_scope=cs,
                                  speculative_turns=None, pipeline_index=0, sample_rate=16000, enable_live_transcription=False,
                                  live_transcription_update_interval=0.5)
    h = spec.create_handler(ctx, spec.normalize(spec.config_type(echo_llm_prefix="")))
    chat = SimpleNamespace(buffer=[SimpleNamespace(role="user", content=[SimpleNamespace(text="first")]), SimpleNamespace(role="assistant", content="x"),
                                   {"role": "user", "content": [{"type": "input_text", "text": "hello there"}]}])
    req = GenerateResponseRequest(response_key="rk", runtime_config=SimpleNamespace(chat=chat), turn_id="t", turn_revision=2, language_code="en",
                                  speech_stopped_at_s=1.5)
    out = list(h.process(req))
    assert isinstance(out[0], LLMResponseChunk) and isinstance(out[1], EndOfResponse) and len(out) == 2
    assert (out[0].text, out[0].turn_id, out[0].turn_revision, out[0].cancel_generation, out[0].response_key, out[0].speech_stopped_at_s) == \
        ("hello there", "t", 2, 1, "rk", 1.5)
    assert (out[1].turn_id, out[1].turn_revision, out[1].cancel_generation, out[1].response_key) == ("t", 2, 1, "rk")


def test_openai_compatible_server_serves_the_reference_client_request(pool):
    """The request the reference's --stt openai client posts (STT/openai_compatible_handler.py:57-80: multipart `file` = mono PCM16
    WAV, `model`, `language`, `response_format`) is answered from the session pool; the TTS route streams int16 PCM."""
    from starlette.testclient import TestClient

    from speech_to_speech_amd.server import create_app, float_pcm16_wav
    from speech_to_speech_amd.tts import TTSPrompt

    app = create_app(pool, "whisper", tokenizer=IdTextCodec(), tts_prompt=lambda text: TTSPrompt(np.zeros((4, 8), np.float32)), tts_max_new_tokens=16)
    client = TestClient(app)
    wav = float_pcm16_wav((np.zeros(1697) * 32767).astype(np.int16))
    r = client.post("/v1/audio/transcriptions", data={"model": "nvidia/parakeet-tdt-0.6b-v3", "language": "de", "response_format": "json"},
                    files={"file": ("audio.wav", wav, "audio/wav")})
    assert r.status_code == 200 and r.json() == {"text": f"t{1697 % 97} t5", "language": "de"}
    r = client.post("/v1/audio/transcriptions", data={"response_format": "text"}, files={"file": ("audio.wav", wav, "audio/wav")})
    assert r.status_code == 200 and r.text == f"t{1697 % 97} t5"
    assert client.post("/v1/audio/transcriptions", data={"model": "x"}).status_code == 400
    r = client.post("/v1/audio/speech", json={"input": "Hello.", "response_format": "pcm"})
    pcm = np.frombuffer(r.content, dtype="<i2")
    assert r.status_code == 200 and len(pcm) == 3 * 700 and set(pcm.tolist()) == {3000, 3001, 3002}
    assert client.get("/health").json()["status"] == "ok"


def test_stt_shim_drops_progressive_behind_its_final_and_purges_stale_queue(pool):
    """BaseSTTHandler behaviours the reference has (STT/base_stt_handler.py:24-62,105-131,215-217): a progressive whose own final is
    already queued is skipped, stale queued inputs are purged when a stale item is met, session end clears the finalised revisions."""

    class Tracker:
        latest = {"t": 2}

        def is_latest(self, t, r):
            return self.latest.get(t, r) == r

        def is_latest_after_pending_reopen(self, t, r):
            return self.is_latest(t, r)

        def is_latest_after_stability_window(self, t, r, settle):
            return self.is_latest(t, r)

    h = _whisper(pool)
    h.speculative_turns = Tracker()
    mk = lambda mode, rev: VADAudio(audio=np.zeros(800, np.float32), mode=mode, turn_id="t", turn_revision=rev)   # noqa: E731
    h.queue_in.put(mk("final", 2))
    assert h.should_process_input(mk("progressive", 2)) is False           # its final is waiting behind it
    assert h.queue_in.qsize() == 1
    h.queue_in.put(mk("progressive", 1))                                   # superseded revision, queued
    assert h.should_process_input(mk("final", 1)) is False                 # stale input ...
    assert h.queue_in.qsize() == 1 and h.queue_in.queue[0].turn_revision == 2   # ... purges the stale queued one, keeps the live final
    out = Transcription(text="x", turn_id="t", turn_revision=2)
    h.before_emit_output(out)
    assert h.should_process_input(mk("progressive", 2)) is False           # revision already finalised
    h.on_session_end()
    assert h.should_process_input(mk("final", 2)) is True
