"""Handler-contract tests (CPU, fake engines) for the parts of the reference's handler surface round 4 lacked:
`on_session_end()` / `cleanup()` on the three handlers (STT/parakeet_tdt_handler.py:636-648, TTS/qwen3_tts_handler.py:1005-1034), the
language ID hook on the Parakeet transcript (:379-403, :588-596), the Qwen3-TTS generation modes (voice clone / custom voice / voice
design, :841-858, :910-1003) and the per-session voice override (:506-553).  The override tests are the reference's own
(tests/test_qwen3_tts_handler_backend.py:462-503), restated for a layout-backed front-end."""

import logging
import threading
import time
from queue import Queue
from threading import Event
from types import SimpleNamespace

import numpy as np
import pytest

from speech_to_speech_amd.boundary import TTSInput, VADAudio
from speech_to_speech_amd.handlers.parakeet import MI355XParakeetSTTHandler
from speech_to_speech_amd.handlers.qwen3_tts import (MI355XQwen3TTSHandler, infer_model_type, load_reference_codes, load_speaker_embedding)
from speech_to_speech_amd.handlers.whisper import MI355XWhisperSTTHandler
from speech_to_speech_amd.qwen3_tts_checkpoint import TableTextFrontend, TTSPromptLayout
from speech_to_speech_amd.session_pool import SessionPool
from speech_to_speech_amd.text import ByteTextFrontend, IdTextCodec

E_, G_ = 8, 4


class FakeSTT:
    def __init__(self, ids):
        self.ids = ids

    def transcribe_batch(self, audios, options):
        return [{"ids": list(self.ids), "language": None, "starts_s": [0.08 * i for i in range(len(self.ids))]} for _ in audios]

    def close(self):
        self.closed = True


class TextCodec:
    """ids -> a fixed sentence (the language hook sees real text)."""

    def __init__(self, text):
        self.text = text

    def decode(self, ids):
        return self.text if ids else ""


class FakeTTS:
    max_prompt = 96
    dims = SimpleNamespace(t_hidden=E_, n_groups=G_)

    def __init__(self, frontend):
        self.text_frontend = frontend
        self.prompts = []

    def stream(self, prompts, max_new_tokens, chunk_size=8, ctx_frames=25, min_frames=0):
        self.prompts.extend(p.embeds for p in prompts)
        self.trailing = getattr(self, "trailing", []) + [p.trailing for p in prompts]
        yield [np.full(1100, 4000, np.int16) for _ in prompts]

    def close(self):
        self.closed = True


def _layout_frontend(model_type=None, speakers=None):
    rng = np.random.default_rng(0)
    layout = TTSPromptLayout(text_bos=256, text_eos=257, codec_prefix=[1], codec_bos=2, speaker_ids=speakers or {}, language_ids={"english": 5},
                             model_type=model_type)
    return TableTextFrontend(rng.standard_normal((258, E_)).astype(np.float32), layout, None, None,
                             codec_table=rng.standard_normal((16, E_)).astype(np.float32),
                             group_tables=[rng.standard_normal((6, E_)).astype(np.float32) for _ in range(G_ - 1)])


@pytest.fixture
def pool():
    p = SessionPool("fake:1", max_batch=4, batch_window_ms=1.0, tts_lanes=1)
    yield p
    p.shutdown()


def _tts(pool, frontend, **attrs):
    pool._engines["tts"] = FakeTTS(frontend)
    h = object.__new__(MI355XQwen3TTSHandler)
    h.stop_event, h.queue_in, h.queue_out, h._times = Event(), Queue(), Queue(), []
    h.should_listen, h.cancel_scope, h.speculative_turns = Event(), None, None
    h.streaming_chunk_size, h.max_new_tokens, h.blocksize, h.pool = 8, 360, 512, pool
    h.speaker = h.language = h.instruct = h.ref_audio = h.ref_spk = h.ref_rvq = h.ref_text = None
    h.xvec_only, h.model_name = False, "synthetic:test-tiny"
    for k, v in attrs.items():
        setattr(h, k, v)
    return h


# ---- Qwen3-TTS: modes ---------------------------------------------------------------------------------------------------------------
def test_infer_model_type_follows_the_reference_name_rules():
    assert infer_model_type("Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice") == "custom_voice"
    assert infer_model_type("Qwen/Qwen3-TTS-12Hz-1.7B-VoiceDesign") == "voice_design"
    assert infer_model_type("Qwen/Qwen3-TTS-12Hz-1.7B-Base") == "base"
    assert infer_model_type("something-else") == "base"


def test_custom_voice_prompt_carries_speaker_language_and_instruct_rows(pool):
    fe = _layout_frontend("custom_voice", {"aiden": 7, "vivian": 8})
    h = _tts(pool, fe, speaker="Vivian", language="en", instruct="whisper")
    out = list(h.process(TTSInput(text="Hi.")))
    assert len(out) >= 2 and all(b.dtype == np.int16 and len(b) == 512 for b in out)
    rows = pool.engine("tts").prompts[0]
    want = fe.embed("Hi.", 96, speaker="vivian", language="english", instruct="whisper")
    assert np.array_equal(rows, want)
    # control rows: prefix, language, speaker; then the instruct bytes in front of the text
    assert np.allclose(rows[2], fe.codec_table[8]) and np.allclose(rows[1], fe.codec_table[5])
    assert len(rows) == 3 + len("whisper") + (2 + len("Hi.")) + 1


def test_voice_design_uses_instruct_and_no_speaker_row(pool):
    fe = _layout_frontend("voice_design")
    h = _tts(pool, fe, instruct="a calm low voice")
    assert list(h.process(TTSInput(text="Hello there.")))
    assert np.array_equal(pool.engine("tts").prompts[0], fe.embed("Hello there.", 96, instruct="a calm low voice"))


def test_streaming_text_mode_prefills_up_to_the_first_text_token_and_trails_the_rest(pool):
    """Reference arguments `qwen3_tts_non_streaming_mode = False` (:103-108): the prefill ends with the first text token (then the rows that
    follow the text block: codec_bos), the remaining text rows + text_eos are handed to the engine as trailing rows, one per frame.
    Prefill text rows + trailing rows = the text block of the non-streaming prompt, in order, nothing lost or duplicated."""
    fe = _layout_frontend("custom_voice", {"vivian": 8})
    text = "Hello there."
    h = _tts(pool, fe, speaker="vivian", language="en", instruct="calm", non_streaming_mode=False)
    assert list(h.process(TTSInput(text=text)))
    eng = pool.engine("tts")
    rows, trail = eng.prompts[0], eng.trailing[0]
    full = fe.embed(text, 96, speaker="vivian", language="english", instruct="calm")
    n_head, n_ins = 3, len("calm")
    k = n_head + n_ins + 2                                  # control rows, instruct, text_bos, FIRST text token
    assert trail.dtype == np.float32 and len(trail) == len(text) - 1 + 1          # the rest of the text + text_eos
    assert len(rows) == k + 1 and len(rows) + len(trail) == len(full)
    assert np.array_equal(rows[:k], full[:k]) and np.array_equal(rows[k], full[-1])          # ... then the codec_bos row
    assert np.array_equal(trail, full[k:-1])               # text rows carry no codec side: the trailing rows are the prompt's own
    # the default (None / True) still pre-fills the whole text and hands over no trailing rows
    h2 = _tts(pool, fe, speaker="vivian", language="en", non_streaming_mode=True)
    assert list(h2.process(TTSInput(text=text)))
    assert eng.trailing[-1] is None and np.array_equal(eng.prompts[-1], fe.embed(text, 96, speaker="vivian", language="english"))
    # byte frontend (synthetic models): same split; a one-token text trails only the eos row
    bfe = ByteTextFrontend(np.arange(258 * 4, dtype=np.float32).reshape(258, 4))
    pre, tr = bfe.embed("abc", 32, streaming=True)
    assert np.array_equal(np.concatenate([pre, tr]), bfe.embed("abc", 32)) and len(pre) == 2 and len(tr) == 3
    pre, tr = bfe.embed("a", 32, streaming=True)
    assert len(pre) == 2 and len(tr) == 1 and np.array_equal(tr[0], bfe.embed("a", 32)[-1])


def test_base_model_without_reference_yields_no_audio_and_says_why(pool, caplog):
    h = _tts(pool, _layout_frontend("base"))
    with caplog.at_level(logging.ERROR):
        assert list(h.process(TTSInput(text="Hello."))) == []       # reference policy: generation failure -> no audio, thread keeps running
    assert "requires a voice-clone reference" in caplog.text


def test_voice_clone_uses_precomputed_references_without_audio(pool, tmp_path):
    """Reference tests/test_qwen3_tts_handler_backend.py:806-843: ref_spk / ref_rvq are enough, ref_audio stays None."""
    fe = _layout_frontend("base")
    spk = np.arange(E_, dtype=np.float32) / 10
    codes = np.array([[3, 1, 2, 0], [4, 5, 0, 1], [9, 2, 2, 2]], dtype=np.int32)
    spk.tofile(tmp_path / "voice.spk")
    codes.tofile(tmp_path / "voice.rvq")
    h = _tts(pool, fe, ref_spk=tmp_path / "voice.spk", ref_rvq=tmp_path / "voice.rvq", ref_text="Reference text.", language="English")
    assert list(h.process(TTSInput(text="Hello there.")))
    rows = pool.engine("tts").prompts[0]
    ref_rows = fe.code_rows(codes)
    assert np.allclose(ref_rows[1], fe.codec_table[4] + fe.group_tables[0][5] + fe.group_tables[1][0] + fe.group_tables[2][1])
    assert np.array_equal(rows, fe.embed("Hello there.", 96, language="english", speaker_row=spk, ref_text="Reference text.", ref_rows=ref_rows))
    assert np.allclose(rows[2], spk) and np.allclose(rows[-3:], ref_rows)      # speaker row after prefix + language; frames at the end
    # x-vector only: no in-context rows, no reference transcript
    h.xvec_only = True
    pool.engine("tts").prompts.clear()
    assert list(h.process(TTSInput(text="Hello there.")))
    assert np.array_equal(pool.engine("tts").prompts[0], fe.embed("Hello there.", 96, language="english", speaker_row=spk))


def test_voice_clone_from_raw_audio_is_refused_not_approximated(pool, caplog):
    h = _tts(pool, _layout_frontend("base"), ref_audio="TTS/ref_audio.wav")
    with caplog.at_level(logging.ERROR):
        assert list(h.process(TTSInput(text="Hello."))) == []
    assert "speaker encoder" in caplog.text


def test_reference_loaders_validate_sizes(tmp_path):
    np.zeros(5, np.float32).tofile(tmp_path / "bad.spk")
    with pytest.raises(ValueError, match="expected the talker width"):
        load_speaker_embedding(tmp_path / "bad.spk", 8)
    np.save(tmp_path / "c.npy", np.zeros((4, 7), np.int64))        # [groups, frames] is accepted
    assert load_reference_codes(tmp_path / "c.npy", 4).shape == (7, 4)
    np.zeros(7, np.int32).tofile(tmp_path / "odd.rvq")
    with pytest.raises(ValueError, match="whole number"):
        load_reference_codes(tmp_path / "odd.rvq", 4)


def test_synthetic_frontend_voice_modes():
    fe = ByteTextFrontend(np.arange(258 * 4, dtype=np.float32).reshape(258, 4))
    plain = fe.embed("ab", 32)
    assert plain.shape == (4, 4)
    full = fe.embed("ab", 32, instruct="x", speaker_row=np.ones(4), ref_text="r", ref_rows=np.zeros((2, 4)))
    assert full.shape == (1 + 3 + (2 + 4) + 2, 4) and np.array_equal(full[0], np.ones(4))
    with pytest.raises(ValueError):
        fe.embed("abcdef", 8, instruct="xyz")


# ---- Qwen3-TTS: session voice override (reference tests :462-503) and lifecycle ------------------------------------------------------
def _cfg(voice):
    return SimpleNamespace(session=SimpleNamespace(audio=SimpleNamespace(output=SimpleNamespace(voice=voice))))


def test_apply_session_voice_override_warns_for_non_file_for_base_model(pool, caplog):
    h = _tts(pool, _layout_frontend("base"), ref_audio="TTS/ref_audio.wav")
    with caplog.at_level("WARNING"):
        h._apply_session_voice_override("base", runtime_config=_cfg("alloy"))
    assert h.ref_audio == "TTS/ref_audio.wav" and h.speaker is None
    assert "Ignoring Qwen3-TTS session voice override" in caplog.text


def test_apply_session_voice_override_ignores_unsupported_custom_voice_speaker(pool, caplog):
    h = _tts(pool, _layout_frontend("custom_voice", {"aiden": 7, "vivian": 8}), speaker="Aiden")
    with caplog.at_level("WARNING"):
        h._apply_session_voice_override("custom_voice", runtime_config=_cfg("cedar"))
    assert h.ref_audio is None and h.speaker == "Aiden"
    assert "not a supported CustomVoice speaker" in caplog.text and "cedar" in caplog.text


def test_apply_session_voice_override_accepts_supported_custom_voice_speaker(pool):
    h = _tts(pool, _layout_frontend("custom_voice", {"aiden": 7, "vivian": 8}), speaker="Aiden", ref_audio="TTS/ref_audio.wav")
    h._apply_session_voice_override("custom_voice", runtime_config=_cfg("Vivian"))
    assert h.ref_audio is None and h.speaker == "vivian"


def test_response_voice_wins_and_is_used_by_the_next_prompt_then_session_end_restores(pool):
    fe = _layout_frontend("custom_voice", {"aiden": 7, "vivian": 8})
    h = _tts(pool, fe, speaker="aiden")
    h._initial_speaker, h._initial_ref_audio, h._initial_ref_spk, h._initial_ref_rvq = "aiden", None, None, None
    resp = SimpleNamespace(audio=SimpleNamespace(output=SimpleNamespace(voice="VIVIAN")))
    assert list(h.process(TTSInput(text="Hi.", response=resp, runtime_config=_cfg("aiden"))))
    assert h.speaker == "vivian" and np.array_equal(pool.engine("tts").prompts[0], fe.embed("Hi.", 96, speaker="vivian"))
    h.on_session_end()
    assert h.speaker == "aiden" and h.ref_spk is None
    pool.engine("tts").prompts.clear()
    assert list(h.process(TTSInput(text="Hi.")))
    assert np.array_equal(pool.engine("tts").prompts[0], fe.embed("Hi.", 96, speaker="aiden"))


def test_cleanup_releases_the_device_pool_with_its_last_handler():
    pool = SessionPool.for_device("fake:cleanup", max_batch=2, batch_window_ms=1.0, tts_lanes=1)
    eng = FakeTTS(ByteTextFrontend(np.zeros((258, E_), np.float32)))
    pool._engines["tts"] = eng
    a, b = object.__new__(MI355XQwen3TTSHandler), object.__new__(MI355XWhisperSTTHandler)
    for h in (a, b):
        h.pool, h._owns_pool = pool, True
        pool.attach(h)
    a.cleanup()
    assert a.pool is None and SessionPool._pools.get("fake:cleanup") is pool and not getattr(eng, "closed", False)
    b.cleanup()
    assert "fake:cleanup" not in SessionPool._pools and eng.closed
    b.cleanup()                                            # idempotent
    # a pool somebody CONSTRUCTED belongs to that caller: its handlers' cleanup never shuts it down
    p2 = SessionPool("fake:given", max_batch=2, batch_window_ms=1.0, tts_lanes=1)
    c = object.__new__(MI355XParakeetSTTHandler)
    c.pool, c._owns_pool = p2, False
    p2.attach(c)
    c.cleanup()
    assert not p2._stopping
    p2.shutdown()


def test_every_user_of_a_device_pool_is_counted_and_a_failed_setup_lets_go():
    """(a) a handler that was GIVEN a `for_device` pool, and a direct `for_device` user (server, bench), keep the engines alive
    after the last self-served handler cleaned up; (b) a setup() that raises after attaching does not leave a stale attachment."""
    from speech_to_speech_amd.session_pool import release_pool_on_failure

    pool = SessionPool.for_device("fake:users", max_batch=2, batch_window_ms=1.0, tts_lanes=1)
    eng = FakeTTS(ByteTextFrontend(np.zeros((258, E_), np.float32)))
    pool._engines["tts"] = eng
    server = object()
    pool.attach(server)
    own, given = object.__new__(MI355XWhisperSTTHandler), object.__new__(MI355XQwen3TTSHandler)
    own.pool = given.pool = pool
    pool.attach(own)
    pool.attach(given)
    own.cleanup()
    assert SessionPool._pools.get("fake:users") is pool and not getattr(eng, "closed", False)
    given.cleanup()
    assert SessionPool._pools.get("fake:users") is pool and not getattr(eng, "closed", False)      # the server still uses it

    class Boom:
        pool = None

        @release_pool_on_failure
        def setup(self):
            self.pool = pool
            pool.attach(self)
            raise RuntimeError("bad checkpoint")

    b = Boom()
    with pytest.raises(RuntimeError):
        b.setup()
    assert b.pool is None and id(b) not in pool._attached
    assert pool.detach(server) and "fake:users" not in SessionPool._pools and eng.closed


# ---- Parakeet: language ID hook + lifecycle ----------------------------------------------------------------------------------------
def _parakeet(pool, text, language="auto", detector=None):
    pool._engines["parakeet"] = FakeSTT([1, 2, 3])
    h = object.__new__(MI355XParakeetSTTHandler)
    h.stop_event, h.queue_in, h.queue_out, h._times = Event(), Queue(), Queue(), []
    h.start_language, h.last_language, h.tokenizer, h.pool = language, "en" if language in (None, "auto") else language, TextCodec(text), pool
    h.language_detector, h.processing_final, h._windows = detector, False, {}
    return h


def test_parakeet_language_id_on_the_transcript(pool):
    seen = []

    def det(text):
        seen.append(text)
        return "de"

    h = _parakeet(pool, "Das ist ein ziemlich langer deutscher Satz.", detector=det)
    out = list(h.process(VADAudio(audio=np.zeros(16000, np.float32), mode="final", turn_id="t", turn_revision=1)))
    assert out[0].language_code == "de" and h.last_language == "de" and seen == ["Das ist ein ziemlich langer deutscher Satz."]
    # short utterances keep the previous language and never reach the detector (reference :392-394)
    h.tokenizer = TextCodec("Ja.")
    assert list(h.process(VADAudio(audio=np.zeros(16000, np.float32), mode="final")))[0].language_code == "de" and len(seen) == 1
    # an unsupported code falls back to the last language (:348-352)
    h.tokenizer, h.language_detector = TextCodec("x" * 40), lambda t: "ja"
    assert list(h.process(VADAudio(audio=np.zeros(16000, np.float32), mode="final")))[0].language_code == "de"
    h.on_session_end()
    assert h.last_language == "en" and h._windows == {} and h.processing_final is False


def test_parakeet_fixed_language_is_reported_without_detection(pool):
    h = _parakeet(pool, "Das ist ein ziemlich langer deutscher Satz.", language="fr", detector=lambda t: 1 / 0)
    assert list(h.process(VADAudio(audio=np.zeros(16000, np.float32), mode="final")))[0].language_code == "fr"


def test_parakeet_skips_progressive_items_while_a_final_is_in_flight(pool):
    h = _parakeet(pool, "hello")
    h.processing_final = True
    assert list(h.process(VADAudio(audio=np.zeros(16000, np.float32), mode="progressive", turn_id="a", turn_revision=0))) == []


def test_whisper_session_end_resets_the_carried_language(pool):
    h = object.__new__(MI355XWhisperSTTHandler)
    h.start_language, h.last_language = "auto", "de"
    h.on_session_end()
    assert h.last_language is None
    h.start_language, h.last_language = "fr", "de"
    h.on_session_end()
    assert h.last_language == "fr"


# ---- pool: cancelled requests get a terminal item -------------------------------------------------------------------------
def test_cancelled_request_dropped_from_a_backlog_terminates_its_stream():
    from speech_to_speech_amd.session_pool import _TTSRequest

    live, dead = _TTSRequest(prompt=None, max_new_tokens=8), _TTSRequest(prompt=None, max_new_tokens=8)
    dead.cancelled.set()
    assert SessionPool._drop_cancelled([live, dead]) == [live]
    assert dead.chunks.get_nowait() is None and live.chunks.empty()


# ---- pool: live partials behind the finals, per-unit skip rule, worker routing ---------------------------
class _GatedSTT:
    """Blocks its first batch until released; records what every batch held."""

    max_batch = 4

    def __init__(self):
        self.batches, self.gate, self.entered = [], threading.Event(), threading.Event()

    def transcribe_batch(self, audios, options):
        self.batches.append([int(a[0]) for a in audios])
        self.entered.set()
        self.gate.wait(10)
        return [{"ids": [int(a[0])], "language": None, "starts_s": [0.0]} for a in audios]

    def replica(self):
        return self

    def close(self):
        pass


def _tagged(tag):
    a = np.zeros(1600, np.float32)
    a[0] = tag
    return a


def test_pool_serves_finals_before_queued_live_partials_and_drops_none():
    p = SessionPool("fake:prio", max_batch=4, batch_window_ms=1.0, tts_lanes=1)
    eng = _GatedSTT()
    p._engines["parakeet"] = eng
    out = {}
    try:
        def call(tag, live):
            out[tag] = p.transcribe("parakeet", _tagged(tag), live=live)["ids"]

        first = threading.Thread(target=call, args=(1, False))
        first.start()
        assert eng.entered.wait(5)                       # batch [1] is on the engine: everything below queues up behind it
        ths = [threading.Thread(target=call, args=(10 + k, True)) for k in range(6)]         # six live partials first ...
        [t.start() for t in ths]
        while p.stt_backlog() < 6:
            time.sleep(0.002)
        fin = [threading.Thread(target=call, args=(2 + k, False)) for k in range(2)]         # ... then two finals
        [t.start() for t in fin]
        while p.stt_backlog() < 8:
            time.sleep(0.002)
        eng.gate.set()
        [t.join(10) for t in [first] + ths + fin]
        assert eng.batches[0] == [1]
        assert set(eng.batches[1][:2]) == {2, 3} and len(eng.batches[1]) == 4          # the finals lead the next batch, partials fill its free rows
        assert sorted(sum(eng.batches, [])) == [1, 2, 3, 10, 11, 12, 13, 14, 15]        # every partial was answered: the pool sheds nothing
        assert all(out[t] == [t] for t in out) and len(out) == 9
    finally:
        eng.gate.set()
        p.shutdown()


def test_extra_stt_workers_never_see_what_only_the_first_worker_serves():
    """builds, host calls and requests for engines without replica() used to be taken and re-appended by the extra workers."""
    p = SessionPool("fake:route", max_batch=4, batch_window_ms=1.0, tts_lanes=1, stt_workers=2)
    try:
        p._engines["norep"] = FakeSTT([7])                                # no replica(): first worker only
        req_q = p._stt_route(SimpleNamespace(model="norep", live=False))
        assert req_q is p._stt_q0
        p._engines["rep"] = _GatedSTT()
        p._engines["rep"].gate.set()
        assert p._stt_route(SimpleNamespace(model="rep", live=False)) is p._stt_q and p._stt_route(SimpleNamespace(model="rep", live=True)) is p._stt_pq
        assert p.transcribe("norep", _tagged(5))["ids"] == [7]
        assert p.transcribe("rep", _tagged(5))["ids"] == [5]
        built = []
        p.ensure_engine("late", lambda: built.append(threading.current_thread().name) or FakeSTT([1]))
        assert built and built[0].endswith("fake:route")                  # built on the first worker's thread
    finally:
        p.shutdown()


def test_a_pool_grows_to_the_stt_workers_a_handler_asks_for():
    """`ensure_stt_workers` (MI355XParakeetSTTHandler(stt_workers = 3): live partials): the pool starts the extra workers after it exists, builds
    their replicas of the engines that are already there, never shrinks; while the first worker is stuck in a batch the others serve the queue."""
    p = SessionPool("fake:grow", max_batch=4, batch_window_ms=1.0, tts_lanes=1)
    try:
        reps = []

        class Rep(_GatedSTT):
            def replica(self):
                r = _GatedSTT()
                r.gate.set()
                reps.append(r)
                return r

        eng = Rep()
        p._engines["rep"] = eng
        assert p.stt_workers == 1 and p._stt_route(SimpleNamespace(model="rep", live=True)) is p._stt_pq
        p.ensure_stt_workers(3)
        p.ensure_stt_workers(2)
        assert p.stt_workers == 3 and len(p._stt_extra) == 2 and len(reps) == 2 and all(t.is_alive() for t in p._stt_extra)
        # requests until the FIRST worker holds one (its engine blocks on the gate); whatever the other workers took is answered at once
        for k in range(1, 200):
            threading.Thread(target=lambda k=k: p.transcribe("rep", _tagged(k)), daemon=True).start()
            if eng.entered.wait(0.05):
                break
        assert eng.entered.is_set()
        t = time.time()
        assert [p.transcribe("rep", _tagged(k), live=True)["ids"] for k in (1002, 1003, 1004)] == [[1002], [1003], [1004]]
        assert time.time() - t < 5.0 and not eng.gate.is_set()           # served by the replicas while the first worker is still stuck
    finally:
        eng.gate.set()
        p.shutdown()


def test_parakeet_partial_is_skipped_only_while_this_units_own_partial_runs():
    """Reference STT/parakeet_tdt_handler.py:264-295: `with self._compute_lock_context(timeout=0.01)` - the lock is the handler's own (:129).
    A deep POOL backlog (other units' work) does not skip a partial; a held lock of this unit does."""
    p = SessionPool("fake:unit", max_batch=2, batch_window_ms=1.0, tts_lanes=1)
    try:
        h = _parakeet(p, "hello there")
        h.incremental_partials = False
        for k in range(12):                               # a backlog far beyond max_batch, made of OTHER units' partials
            threading.Thread(target=lambda k=k: p.transcribe("parakeet", _tagged(k), live=True), daemon=True).start()
        va = VADAudio(audio=np.zeros(16000, np.float32), mode="progressive", turn_id="t", turn_revision=0)
        got = list(h.process(va))
        assert len(got) == 1 and got[0].text == "hello there"
        h._compute_lock.acquire()
        try:
            va2 = VADAudio(audio=np.zeros(24000, np.float32), mode="progressive", turn_id="t", turn_revision=0)
            assert list(h.process(va2)) == []
        finally:
            h._compute_lock.release()
        assert len(list(h.process(va2))) == 1
    finally:
        p.shutdown()
