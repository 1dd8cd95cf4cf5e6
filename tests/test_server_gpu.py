"""OpenAI-compatible server mode on the real engines: what the reference's `--stt openai` client posts
(STT/openai_compatible_handler.py:57-80, 115-160) is answered by the MI355X session pool - the transcript is the text of the oracle's
greedy ids for that audio, `/v1/audio/speech` streams exactly the int16 samples a direct engine call produces, a bad request is
refused without touching the sessions batched next to it."""

import threading

import numpy as np
import pytest
import torch

from oracle import logmel as OL
from oracle import whisper as OW
from speech_to_speech_amd.session_pool import SessionPool
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import Qwen3TTSDims, WhisperDims, qwen3_tts_state_dict, whisper_state_dict
from speech_to_speech_amd.text import ByteTextFrontend, IdTextCodec, WhisperVocab
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt
from speech_to_speech_amd.whisper import WhisperEngine

pytestmark = pytest.mark.gpu
VOC = WhisperVocab(sot=300, eot=319, transcribe=305, notimestamps=306, lang_to_id={"en": 301, "de": 302, "fr": 303, "es": 304},
                   suppress=(1, 2, 7, 300, 301, 302, 303, 304, 305, 306), begin_suppress=(220, 319))


def test_server_routes_match_direct_engine_calls():
    from starlette.testclient import TestClient

    from speech_to_speech_amd.server import create_app, float_pcm16_wav

    wd = WhisperDims.test_tiny()
    wsd = whisper_state_dict(wd, seed=3, gain=2.0, pos_std=2.0)
    td = Qwen3TTSDims.test_tiny()
    tsd = qwen3_tts_state_dict(td, seed=5)
    g = torch.Generator().manual_seed(7)
    front = ByteTextFrontend(torch.randn(258, td.t_hidden, generator=g).numpy())
    pad = torch.randn(td.t_hidden, generator=g) * 0.5
    pool = SessionPool("cuda:0", max_batch=4, batch_window_ms=20.0, tts_chunk_size=4, tts_lanes=1)
    pool.ensure_engine("whisper", lambda: WhisperEngine(wd, wsd, dtype=torch.float32, max_batch=4))
    pool.ensure_engine("tts", lambda: TTSEngine(td, tsd, dtype=torch.float32, max_batch=4, max_prompt=32, max_frames=16, max_codec_frames=8, pad_embed=pad))
    prompt_of = lambda text: TTSPrompt(front.embed(text, 32))      # noqa: E731
    app = create_app(pool, "whisper", tokenizer=IdTextCodec(), stt_options={"vocab": VOC, "max_new_tokens": 10}, tts_prompt=prompt_of, tts_max_new_tokens=12)
    client = TestClient(app)

    # ---- /v1/audio/transcriptions: three clients at once (they are batched), one of them with an unknown language
    lens = [32000, 20500, 41000]
    pcm16 = [np.clip(synth_audio(n, 300 + i) * 32768, -32768, 32767).astype(np.int16) for i, n in enumerate(lens)]
    audios = [p.astype(np.float32) / 32768.0 for p in pcm16]          # what the server decodes from the WAV
    feats = torch.from_numpy(np.stack([OL.whisper_logmel(a, wd.n_mels) for a in audios]).astype(np.float32))
    with torch.no_grad():
        enc = OW.encode(wsd, feats, wd.n_heads, wd.enc_layers)
        ids, _ = OW.greedy_decode(wsd, enc, wd.n_heads, wd.dec_layers, [VOC.prompt("de"), VOC.prompt("en"), VOC.prompt("en")], VOC.eot, 10, VOC.suppress, VOC.begin_suppress)
    want = [" ".join(f"t{j}" for j in row) for row in ids]
    got: dict = {}

    def post(i, lang):
        got[i] = client.post("/v1/audio/transcriptions", data={"model": "openai/whisper-large-v3", "language": lang, "response_format": "json"},
                             files={"file": ("audio.wav", float_pcm16_wav(pcm16[i % 3]), "audio/wav")})

    ths = [threading.Thread(target=post, args=(i, lang)) for i, lang in enumerate(["de", "en", "en", "xx"])]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert got[3].status_code == 400 and "unsupported language" in got[3].text            # refused before it can fail the batch it would have joined
    for i in range(3):
        assert got[i].status_code == 200, got[i].text
        assert got[i].json() == {"text": want[i], "language": ["de", "en", "en"][i]}
    assert pool.stats["stt_requests"] == 3

    # ---- /v1/audio/speech: streamed PCM == the engine's own stream of that prompt; wav == the same samples in a RIFF container
    text = "Hello there."
    r = client.post("/v1/audio/speech", json={"input": text, "response_format": "pcm", "max_new_tokens": 12})
    assert r.status_code == 200
    pcm = np.frombuffer(r.content, dtype="<i2")
    r2 = client.post("/v1/audio/speech", json={"input": text, "response_format": "wav", "max_new_tokens": 12})
    assert r2.status_code == 200 and r2.content[:4] == b"RIFF" and np.array_equal(np.frombuffer(r2.content[44:], dtype="<i2"), pcm)
    assert client.post("/v1/audio/speech", json={"input": text, "response_format": "mp3"}).status_code == 400
    assert client.post("/v1/audio/speech", json={"input": "x" * 400}).status_code == 400          # longer than the engine's prompt budget
    health = client.get("/health").json()
    assert health["status"] == "ok" and health["pool"]["tts_requests"] == 2
    pool.shutdown(close_engines=False)
    eng = pool.engine("tts")
    direct = np.concatenate([c[0] for c in eng.stream([prompt_of(text)], [12], chunk_size=4) if c[0] is not None] or [np.zeros(0, np.int16)])
    print(f"/v1/audio/speech: {len(pcm)} samples served, {len(direct)} from the direct engine stream")
    assert len(pcm) > 0 and pcm.shape == direct.shape          # (live eos on random weights: the reply is short)
    d = np.abs(pcm.astype(int) - direct.astype(int))
    assert d.max() <= 2, f"served PCM differs from the direct engine stream (max {d.max()})"
    eng.close()
    pool.engine("whisper").close()
