"""CPU tests of host-side pieces of the path that need no GPU: output shaping (TTS/qwen3_tts_handler.py:716-742 semantics),
the 24->16 kHz filter design, checkpoint naming, the synthetic text front-ends."""

import numpy as np
import pytest

from speech_to_speech_amd.checkpoints import PARAKEET_DIMS, TTS_DIMS, WHISPER_DIMS, parse_model_name
from speech_to_speech_amd.text import ByteTextFrontend, IdTextCodec
from speech_to_speech_amd.tts import AudioBlocker, resample_fir_2_3


def _reference_blocks(chunks, blocksize=512):
    """The reference's _stream loop restated literally for whole-utterance input: trim the leading silence of the FIRST
    chunk that contains speech (threshold 327, 640 samples of pre-roll), then emit 512-sample blocks, zero-pad the tail."""
    out, found, buf = [], False, np.array([], dtype=np.int16)
    for c in chunks:
        if not found:
            above = np.abs(c) > int(32768 * 0.01)
            if not above.any():
                continue
            c = c[max(0, int(np.argmax(above)) - 640):]
            found = True
        buf = np.concatenate([buf, c])
        while len(buf) >= blocksize:
            out.append(buf[:blocksize])
            buf = buf[blocksize:]
    if len(buf):
        out.append(np.pad(buf, (0, blocksize - len(buf))))
    return out


@pytest.mark.parametrize("lead_silence,chunk_len", [(0, 10240), (3000, 10240), (12000, 10240), (700, 333)])
def test_audio_blocker_trims_silence_and_reblocks(lead_silence, chunk_len):
    rng = np.random.default_rng(lead_silence + chunk_len)
    speech = (rng.standard_normal(25000) * 4000).astype(np.int16)
    speech[np.abs(speech) <= 327] = 1000
    x = np.concatenate([np.zeros(lead_silence, np.int16), speech])
    chunks = [x[i:i + chunk_len] for i in range(0, len(x), chunk_len)]
    blk = AudioBlocker()
    got = [b.copy() for c in chunks for b in blk.push(c)] + [b.copy() for b in blk.flush()]
    want = _reference_blocks(chunks)
    assert len(got) == len(want) and all(g.shape == (512,) and g.dtype == np.int16 for g in got)
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    # pre-roll: at most 640 samples of the silence survive
    flat = np.concatenate(got)
    first = int(np.argmax(np.abs(flat) > 327))
    assert first <= 640


def test_audio_blocker_all_silence_emits_nothing():
    blk = AudioBlocker()
    assert list(blk.push(np.zeros(5000, np.int16))) == [] and list(blk.flush()) == []


def test_resample_filter_is_scipys_design():
    from scipy import signal

    h = resample_fir_2_3()
    # scipy.signal.resample_poly(up=2, down=3): firwin(2*10*3+1, 1/3, window=("kaiser", 5.0)) * up
    want = (signal.firwin(61, 1.0 / 3.0, window=("kaiser", 5.0)) * 2.0)
    assert h.shape == (61,) and np.allclose(h, want.astype(h.dtype), rtol=0, atol=1e-7)
    assert abs(float(h.sum()) - 2.0) < 1e-5


def test_model_names():
    assert parse_model_name("synthetic:large-v3") == ("synthetic", "large-v3", 0)
    assert parse_model_name("synthetic:1.7b:7") == ("synthetic", "1.7b", 7)
    assert set(WHISPER_DIMS) >= {"tiny.en", "large-v3", "distil-large-v3"} and "tdt-0.6b" in PARAKEET_DIMS and "1.7b" in TTS_DIMS
    d = WHISPER_DIMS["large-v3"]()
    assert (d.d_model, d.n_heads, d.enc_layers, d.dec_layers, d.ffn, d.n_mels, d.vocab) == (1280, 20, 32, 32, 5120, 128, 51866)
    d = WHISPER_DIMS["distil-large-v3"]()
    assert (d.enc_layers, d.dec_layers) == (32, 2)
    with pytest.raises(FileNotFoundError, match="no network"):
        parse_model_name("openai/whisper-large-v3")


def test_text_frontends_round_trip():
    codec = IdTextCodec()
    assert codec.encode(codec.decode([5, 17, 300])) == [5, 17, 300]
    table = np.arange(258 * 4, dtype=np.float32).reshape(258, 4)
    fe = ByteTextFrontend(table)
    e = fe.embed("hi", 16)
    assert e.shape[1] == 4 and 2 <= e.shape[0] <= 16
    assert fe.embed("x" * 100, 8).shape[0] == 8


def test_whisper_vocab_from_generation_config_and_synthetic_layouts(tmp_path):
    """suppress_tokens / begin_suppress_tokens / lang_to_id / task ids come from the checkpoint's generation_config.json (what the
    reference's generate() applies, generation_whisper.py:1774-1812); synthetic vocabularies are laid out inside the model's size."""
    import json

    from speech_to_speech_amd.text import WhisperVocab

    gen = {"suppress_tokens": [1, 2, 7, 50359], "begin_suppress_tokens": [220, 50257], "decoder_start_token_id": 50258, "eos_token_id": 50257,
           "no_timestamps_token_id": 50364, "lang_to_id": {"<|en|>": 50259, "<|de|>": 50261}, "task_to_id": {"transcribe": 50360, "translate": 50359},
           "is_multilingual": True}
    (tmp_path / "generation_config.json").write_text(json.dumps(gen))
    v = WhisperVocab.from_checkpoint(str(tmp_path)).validate(51866)
    assert v.suppress == (1, 2, 7, 50359) and v.begin_suppress == (220, 50257) and v.lang_to_id == {"en": 50259, "de": 50261}
    assert v.prompt("de") == [50258, 50261, 50360, 50364] and v.prompt("auto")[1] == -1 and v.prompt("en", "translate")[2] == 50359
    with pytest.raises(ValueError):
        v.prompt("xx")
    with pytest.raises(ValueError):
        v.validate(320)                                      # large-v3 ids do not fit a 320-token model
    t = WhisperVocab.synthetic(320).validate(320)
    assert (t.sot, t.eot, t.transcribe, t.notimestamps) == (300, 319, 305, 306) and t.lang_to_id["en"] == 301
    en = WhisperVocab.synthetic(51864).validate(51864)       # English-only (tiny.en): <sot><notimestamps>, no language token
    assert en.prompt("en") == [50257, 50362] and not en.multilingual
    assert WhisperVocab.synthetic(51866).prompt("fr") == [50258, 50265, 50360, 50364]
    # a generation config that lacks the ids falls back to the tokenizer's special tokens
    (tmp_path / "generation_config.json").write_text(json.dumps({"suppress_tokens": [5]}))
    (tmp_path / "tokenizer.json").write_text(json.dumps({"added_tokens": [{"id": 50257, "content": "<|endoftext|>"}, {"id": 50258, "content": "<|startoftranscript|>"},
                                                                            {"id": 50259, "content": "<|en|>"}, {"id": 50360, "content": "<|transcribe|>"},
                                                                            {"id": 50364, "content": "<|notimestamps|>"}]}))
    (tmp_path / "config.json").write_text(json.dumps({"vocab_size": 51866}))
    f = WhisperVocab.from_checkpoint(str(tmp_path))
    assert (f.sot, f.eot, f.transcribe, f.notimestamps, f.suppress, f.lang_to_id) == (50258, 50257, 50360, 50364, (5,), {"en": 50259})
