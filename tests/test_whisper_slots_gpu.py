"""Continuous batching of the Whisper token loop (s2s_whisper_slots_*, speech-to-speech_amd/stt_continuous.py) vs the oracle.

First run on a GPU in round 5 (profiles/r05_whisper_slots.txt): every session bit-exact; the open-loop A/B against the batch worker
is in the same file (no gain at 1 100 sessions: the STT worker is encoder-bound), so SessionPool(stt_slots=0) stays the default.

float32 engine: a session's greedy ids and detected language are BIT-EXACT against the oracle's greedy decode of that session alone,
whatever else shares the loop: sessions that join while others are mid-sentence, different token budgets, a slot that is reused, the
encoder pass of the next group running on a second stream while the loop steps.
"""

import numpy as np
import pytest
import torch

from oracle import logmel as OL
from oracle import whisper as OW
from speech_to_speech_amd.stt_continuous import ContinuousSTT
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import WhisperDims, whisper_state_dict
from speech_to_speech_amd.text import WhisperVocab
from speech_to_speech_amd.whisper import WhisperEngine

pytestmark = [pytest.mark.gpu]
VOC = WhisperVocab(sot=300, eot=319, transcribe=305, notimestamps=306, lang_to_id={"en": 301, "de": 302, "fr": 303, "es": 304},
                   suppress=(1, 2, 7, 300, 301, 302, 303, 304, 305, 306), begin_suppress=(220, 319))


class _Job:
    def __init__(self, audio, **options):
        self.audio, self.options = audio, {"vocab": VOC, **options}


def _oracle(dims, sd, audio, language, max_new):
    feats = torch.from_numpy(OL.whisper_logmel(audio, dims.n_mels).astype(np.float32))[None]
    with torch.no_grad():
        enc = OW.encode(sd, feats, dims.n_heads, dims.enc_layers)
        ids, filled = OW.greedy_decode(sd, enc, dims.n_heads, dims.dec_layers, [VOC.prompt(language)], VOC.eot, max_new, VOC.suppress, VOC.begin_suppress,
                                       list(VOC.lang_to_id.values()))
    return ids[0], filled[0]


def test_sessions_in_a_running_token_loop_get_their_solo_ids():
    dims = WhisperDims.test_tiny()
    sd = whisper_state_dict(dims, seed=3, gain=2.0, pos_std=2.0)
    eng = WhisperEngine(dims, sd, dtype=torch.float32, max_batch=8)
    side = torch.cuda.Stream()

    def make_event():
        ev = torch.cuda.Event()
        ev.record()
        return ev

    cs = ContinuousSTT(eng, steps_per_round=3, max_group=3, enc_stream=lambda: torch.cuda.stream(side), make_event=make_event,
                       wait_event=lambda ev: torch.cuda.current_stream().wait_event(ev))
    spec = [(48000, "auto", 24), (30500, "en", 6), (64000, "auto", 24), (20000, "de", 12), (160000, "auto", 24), (41000, "en", 24), (8000, "auto", 3)]
    jobs = [_Job(synth_audio(n, 200 + i), language=lang, max_new_tokens=m) for i, (n, lang, m) in enumerate(spec)]
    want = [_oracle(dims, sd, j.audio, j.options["language"], j.options["max_new_tokens"]) for j in jobs]
    got, backlog, rounds = {}, list(jobs), 0
    while backlog or not cs.idle:
        if backlog and cs.can_encode():
            backlog = cs.encode(backlog[:3]) + backlog[3:]          # groups of <= 3 join while earlier sessions are mid-sentence
            assert not cs.rejected
        for j, res in cs.step():
            got[jobs.index(j)] = res
        rounds += 1
        assert rounds < 400
    assert cs.stats["groups"] >= 3 and cs.free == [(0, 8)]
    for i, (ids, filled) in enumerate(want):
        lang = VOC.id_to_lang(filled[1])
        print(f"session {i}: {len(ids)} ids, language {lang}; loop -> {got[i]['language']}, {len(got[i]['ids'])} ids")
        assert got[i]["ids"] == ids[: spec[i][2]] and got[i]["language"] == lang
    # the library checks the scheduler: an idle slot cannot be released twice ...
    with pytest.raises(Exception, match="double release"):
        eng.slots_release(2, 1)
    # ... the batch API refuses to run from the moment a slot is RESERVED by an encoder pass (its cross K|V would be overwritten) ...
    assert cs.can_encode() and cs.encode([jobs[0]]) == []
    with pytest.raises(Exception, match="one mode at a time"):
        eng.transcribe_ids([jobs[1].audio], VOC.prompt("en"), VOC.eot, 4)
    # ... a second encoder pass cannot take a reserved slot, a live slot cannot be admitted again ...
    with pytest.raises(Exception, match="reserved by an earlier encoder pass"):
        eng.slots_encode([jobs[2].audio], 0)
    cs.step()
    assert 0 in cs.rows
    with pytest.raises(Exception, match="double admit"):
        eng.slots_admit(0, [VOC.prompt("en")], [4])
    with pytest.raises(Exception, match="live in the token loop"):
        eng.slots_encode([jobs[2].audio], 0)
    # ... and the session that went through all of that still gets its solo ids
    again = {}
    while not cs.idle:
        for j, res in cs.step():
            again[jobs.index(j)] = res
    assert again[0]["ids"] == want[0][0][: spec[0][2]]
    ids, _ = eng.transcribe_ids([jobs[1].audio], VOC.prompt("en"), VOC.eot, 6, suppress=VOC.suppress, begin_suppress=VOC.begin_suppress)
    assert ids[0] == want[1][0]
    eng.close()
