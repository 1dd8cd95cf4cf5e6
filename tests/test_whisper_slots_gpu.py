"""Continuous batching of the Whisper token loop (s2s_whisper_slots_*, speech-to-speech_amd/stt_continuous.py) vs the oracle.

NOT YET RUN ON A GPU: written at the end of round 4 after the round's GPU budget was spent; the feature is off by default
(SessionPool(stt_slots=0)).  Set S2S_WHISPER_SLOTS=1 to run it (first thing in round 5).

float32 engine: a session's greedy ids and detected language are BIT-EXACT against the oracle's greedy decode of that session alone,
whatever else shares the loop: sessions that join while others are mid-sentence, different token budgets, a slot that is reused, the
encoder pass of the next group running on a second stream while the loop steps.
"""

import os

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

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("S2S_WHISPER_SLOTS") != "1", reason="round-5 work in progress: not yet run on a GPU (S2S_WHISPER_SLOTS=1)")]
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
    # the batch API refuses to run while slots are live, and works again once the loop is empty
    assert cs.can_encode() and cs.encode([jobs[0]]) == []
    with pytest.raises(Exception, match="one mode at a time"):
        eng.transcribe_ids([jobs[1].audio], VOC.prompt("en"), VOC.eot, 4)
    while not cs.idle:
        cs.step()
    ids, _ = eng.transcribe_ids([jobs[1].audio], VOC.prompt("en"), VOC.eot, 6, suppress=VOC.suppress, begin_suppress=VOC.begin_suppress)
    assert ids[0] == want[1][0]
    eng.close()
