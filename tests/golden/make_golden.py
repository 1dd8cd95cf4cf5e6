"""Generates the committed golden fixtures under tests/golden/ from the third-party model
code the reference's hot path executes (transformers, present in this container).

Run from the repo root:  python tests/golden/make_golden.py [section ...]
The fixtures pin oracle/ (CPU restatement) to the real implementation; the GPU parity
tests then compare the HIP path with oracle/ on the same seeded inputs.
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from speech_to_speech_amd.synth import synth_audio  # noqa: E402


def make_logmel():
    """Whisper + Parakeet log-mel of two seeded utterances (one ragged length)."""
    import torch
    from transformers import WhisperFeatureExtractor
    from transformers.audio_utils import mel_filter_bank
    from transformers.feature_extraction_sequence_utils import SequenceFeatureExtractor
    from transformers.models.parakeet.feature_extraction_parakeet import ParakeetFeatureExtractor

    out = {}
    lens = [32000, 23877]
    for i, n in enumerate(lens):
        x = synth_audio(n, session_idx=100 + i)
        for nm in ((80,) if i == 0 else (128,)):
            feats = WhisperFeatureExtractor(feature_size=nm)(x, sampling_rate=16000, return_tensors="np").input_features[0]
            keep = n // 160 + 8      # frames that see signal; the rest is the constant floor
            out[f"whisper{nm}_{i}"] = feats[:, :keep].astype(np.float32)
            out[f"whisper{nm}_{i}_floor"] = np.float32(feats[0, -1])
            out[f"whisper{nm}_{i}_sum"] = np.float64(feats.astype(np.float64).sum())
        # ParakeetFeatureExtractor.__init__ needs librosa only for the mel filters; the file's own
        # alternative (feature_extraction_parakeet.py:84-94) is mel_filter_bank -> use that.
        fe = object.__new__(ParakeetFeatureExtractor)
        SequenceFeatureExtractor.__init__(fe, feature_size=128, sampling_rate=16000, padding_value=0.0)
        fe.hop_length, fe.n_fft, fe.win_length, fe.preemphasis = 160, 512, 400, 0.97
        mf = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=0.0,
                             max_frequency=8000.0, sampling_rate=16000, norm="slaney", mel_scale="slaney")
        fe.mel_filters = torch.from_numpy(mf.T.copy()).to(torch.float32)
        r = fe(x, sampling_rate=16000, return_tensors="pt")
        out[f"parakeet128_{i}"] = r.input_features[0].numpy().astype(np.float32)
        out[f"parakeet128_{i}_valid"] = np.int32(int(r.attention_mask.sum()))
    out["lens"] = np.asarray(lens, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "logmel.npz"), **out)
    print("logmel.npz written")


def make_smart_turn():
    """Smart-Turn input features exactly as reference VAD/smart_turn.py:130-143 builds them: last 8 s / left padding, then
    WhisperFeatureExtractor(chunk_length=8)(..., padding="max_length", max_length=128000, truncation=True, do_normalize=True)."""
    from transformers import WhisperFeatureExtractor

    from oracle.logmel import smart_turn_prepare_audio

    fe = WhisperFeatureExtractor(chunk_length=8)
    out = {}
    lens = [40000, 128000 + 7777, 16000]
    for i, n in enumerate(lens):
        x = synth_audio(n, session_idx=200 + i)
        if i == 2:
            x = x * 0.01 + 0.05          # quiet, with a DC offset: the normalisation matters
        a = smart_turn_prepare_audio(x)
        f = fe(a, sampling_rate=16000, return_tensors="np", padding="max_length", max_length=128000, truncation=True, do_normalize=True).input_features[0]
        assert f.shape == (80, 800)
        out[f"feat_{i}"] = f[:, ::4].astype(np.float32)              # every 4th frame (size) ...
        out[f"tail_{i}"] = f[:, -64:].astype(np.float32)             # ... plus the dense tail, where the speech ends
        out[f"sum_{i}"] = np.float64(f.astype(np.float64).sum())
    out["lens"] = np.asarray(lens, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "smart_turn.npz"), **out)
    print("smart_turn.npz written")


WHISPER_TEST = dict(eos=319, sot=300, lang_ids=[301, 302, 303, 304], transcribe=305, notimestamps=306,
                    suppress=[1, 2, 7, 300, 301, 302, 303, 304, 305, 306], begin_suppress=[220, 319], max_new=12,
                    seed=3, gain=2.0, pos_std=2.0)


def make_whisper():
    """Tiny seeded Whisper through transformers' WhisperForConditionalGeneration (CPU float32):
    encoder output rows, language-detection ids and greedy token ids for two utterances."""
    import torch
    from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration

    from speech_to_speech_amd.synth_weights import WhisperDims, whisper_state_dict

    dims = WhisperDims.test_tiny()
    t = WHISPER_TEST
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=dims.n_mels, d_model=dims.d_model,
                        encoder_layers=dims.enc_layers, decoder_layers=dims.dec_layers,
                        encoder_attention_heads=dims.n_heads, decoder_attention_heads=dims.n_heads,
                        encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn, max_source_positions=1500,
                        max_target_positions=dims.max_target, pad_token_id=t["eos"], bos_token_id=t["eos"],
                        eos_token_id=t["eos"], decoder_start_token_id=t["sot"], suppress_tokens=t["suppress"],
                        begin_suppress_tokens=t["begin_suppress"])
    torch.manual_seed(0)
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = whisper_state_dict(dims, seed=3, gain=2.0, pos_std=2.0)
    missing, unexpected = model.load_state_dict({**sd, "proj_out.weight": sd["model.decoder.embed_tokens.weight"]}, strict=False)
    assert not unexpected and all("proj_out" in m for m in missing), (missing, unexpected)
    gc = model.generation_config
    gc.lang_to_id = {f"<|l{i}|>": v for i, v in enumerate(t["lang_ids"])}
    gc.task_to_id = {"transcribe": t["transcribe"]}
    gc.no_timestamps_token_id = t["notimestamps"]
    gc.is_multilingual = True
    gc.suppress_tokens = t["suppress"]
    gc.begin_suppress_tokens = t["begin_suppress"]

    lens = [48000, 30500]
    audios = [synth_audio(n, 200 + i) for i, n in enumerate(lens)]
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels)
    feats = torch.from_numpy(fe(audios, sampling_rate=16000, return_tensors="np").input_features)
    out = {"lens": np.asarray(lens, np.int32)}
    with torch.no_grad():
        enc_o = model.get_encoder()(feats)
        enc = enc_o.last_hidden_state
        out["enc_rows"] = enc[:, ::60].numpy().astype(np.float32)          # [2, 25, d]
        out["enc_abs_mean"] = np.float64(enc.abs().mean().item())
        lang = model.detect_language(encoder_outputs=enc_o, generation_config=gc)
        out["lang"] = lang.numpy().astype(np.int32)
        ids = model.generate(feats, generation_config=gc, max_new_tokens=t["max_new"], num_beams=1, do_sample=False,
                             return_timestamps=False, task="transcribe", language="<|l1|>")
        out["ids_forced_l1"] = ids.numpy().astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "whisper_tiny.npz"), **out)
    print("whisper_tiny.npz written", out["lang"], out["ids_forced_l1"])


def make_parakeet():
    """Tiny seeded Parakeet-TDT through transformers' ParakeetForTDT (CPU float32): encoder rows, greedy TDT tokens + durations."""
    import torch
    from transformers import ParakeetForTDT, ParakeetTDTConfig
    from transformers.models.parakeet.configuration_parakeet import ParakeetEncoderConfig

    from oracle import logmel as OL   # features via the (already pinned) oracle: the HF extractor needs librosa
    from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

    out = {}
    dims = ParakeetDims.test_tiny()
    for tag, seed, gain in (("a", 12, 1.5), ("b", 11, 1.5), ("stuck", 13, 2.0)):
        sd = parakeet_state_dict(dims, seed=seed, gain=gain)
        enc = ParakeetEncoderConfig(hidden_size=dims.d_model, num_hidden_layers=dims.layers, num_attention_heads=dims.n_heads,
                                    intermediate_size=dims.ffn, num_mel_bins=dims.n_mels, subsampling_conv_channels=dims.sub_channels)
        cfg = ParakeetTDTConfig(vocab_size=dims.vocab, decoder_hidden_size=dims.dec_hidden, num_decoder_layers=dims.dec_layers,
                                blank_token_id=dims.blank, encoder_config=enc, pad_token_id=2)
        m = ParakeetForTDT(cfg).eval()
        sdl = dict(sd)
        for l in range(dims.layers):
            sdl[f"encoder.layers.{l}.conv.norm.num_batches_tracked"] = torch.tensor(0)
        m.load_state_dict(sdl, strict=True)
        lens = [48000, 30500]
        audios = [synth_audio(n, 300 + i) for i, n in enumerate(lens)]
        tmax = max(lens) // 160 + 1
        feats = torch.from_numpy(np.stack([OL.parakeet_logmel(a, 128, t_total=tmax)[0] for a in audios]).astype(np.float32))
        flen = torch.tensor([n // 160 for n in lens])
        amask = torch.arange(tmax)[None, :] < flen[:, None]
        m.generation_config.decoder_start_token_id = dims.blank
        with torch.no_grad():
            eo = m.get_audio_features(input_features=feats, attention_mask=amask)
            g = m.generate(input_features=feats, attention_mask=amask, return_dict_in_generate=True)
        out[f"{tag}_enc"] = eo.last_hidden_state.numpy()[:, ::3].astype(np.float32)
        out[f"{tag}_enc_len"] = eo.attention_mask.sum(-1).numpy().astype(np.int32)
        out[f"{tag}_seq"] = g.sequences.numpy().astype(np.int32)
        out[f"{tag}_dur"] = g.durations.numpy().astype(np.int32)
    out["lens"] = np.asarray(lens, np.int32)
    np.savez_compressed(os.path.join(HERE, "parakeet_tiny.npz"), **out)
    print("parakeet_tiny.npz written", {k: v.shape for k, v in out.items()})


def make_qwen3_tts():
    """Sibling modules of Qwen3-TTS in transformers (CPU float32) on the tiny seeded checkpoint:
    Code2Wav waveform, code-predictor greedy codes, dense Qwen3 decoder stack hidden states."""
    import torch
    from transformers import Qwen3Config, Qwen3Model
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import (Qwen3OmniMoeCode2WavConfig,
                                                                                Qwen3OmniMoeTalkerCodePredictorConfig)
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import (
        Qwen3OmniMoeCode2Wav, Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration)

    from speech_to_speech_amd.synth_weights import Qwen3TTSDims, qwen3_tts_state_dict

    dims = Qwen3TTSDims.test_tiny()
    sd = qwen3_tts_state_dict(dims, seed=5)
    out = {}
    g = torch.Generator().manual_seed(1)
    # code2wav
    c = Qwen3OmniMoeCode2WavConfig(codebook_size=dims.codebook, hidden_size=dims.c_hidden, num_attention_heads=dims.c_heads,
                                   num_key_value_heads=dims.c_heads, intermediate_size=dims.c_ffn, num_hidden_layers=dims.c_layers,
                                   num_quantizers=dims.n_groups, upsample_rates=list(dims.upsample_rates),
                                   upsampling_ratios=list(dims.upsampling_ratios), decoder_dim=dims.c_decoder_dim,
                                   sliding_window=dims.c_window)
    m = Qwen3OmniMoeCode2Wav(c).eval()
    m.load_state_dict({k[len("code2wav."):]: v for k, v in sd.items() if k.startswith("code2wav.")}, strict=True)
    codes = torch.randint(0, dims.codebook, (2, dims.n_groups, 20), generator=g)
    with torch.no_grad():
        out["c2w_codes"] = codes.numpy().astype(np.int32)
        out["c2w_wav"] = m(codes).numpy().astype(np.float32)
    # code predictor
    pc = Qwen3OmniMoeTalkerCodePredictorConfig(vocab_size=dims.codebook, hidden_size=dims.p_hidden, intermediate_size=dims.p_ffn,
                                               num_hidden_layers=dims.p_layers, num_attention_heads=dims.p_heads,
                                               num_key_value_heads=dims.p_kv_heads, head_dim=dims.p_head_dim,
                                               num_code_groups=dims.n_groups, rms_norm_eps=dims.eps)
    pm = Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration(pc).eval()
    pm.load_state_dict({k[len("code_predictor."):]: v for k, v in sd.items() if k.startswith("code_predictor.")}, strict=True)
    ph, e0 = torch.randn(3, dims.p_hidden, generator=g), torch.randn(3, dims.p_hidden, generator=g)
    with torch.no_grad():
        r = pm.generate(inputs_embeds=torch.stack([ph, e0], 1), max_new_tokens=dims.n_groups - 1, do_sample=False,
                        return_dict_in_generate=True)
    out["cp_hidden"], out["cp_e0"], out["cp_codes"] = ph.numpy(), e0.numpy(), r.sequences.numpy().astype(np.int32)
    # talker backbone = dense Qwen3 decoder
    qc = Qwen3Config(vocab_size=dims.codec_vocab, hidden_size=dims.t_hidden, intermediate_size=dims.t_ffn, num_hidden_layers=dims.t_layers,
                     num_attention_heads=dims.t_heads, num_key_value_heads=dims.t_kv_heads, head_dim=dims.t_head_dim, rms_norm_eps=dims.eps,
                     rope_parameters={"rope_theta": dims.t_theta, "rope_type": "default"}, attention_bias=False, tie_word_embeddings=False)
    qm = Qwen3Model(qc).eval()
    sub = {k[len("talker.model."):]: v for k, v in sd.items() if k.startswith("talker.model.") and "codec_embedding" not in k}
    sub["embed_tokens.weight"] = sd["talker.model.codec_embedding.weight"]
    qm.load_state_dict(sub, strict=True)
    x = torch.randn(2, 9, dims.t_hidden, generator=g)
    with torch.no_grad():
        out["talker_in"] = x.numpy()
        out["talker_hidden"] = qm(inputs_embeds=x).last_hidden_state.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "qwen3_tts_tiny.npz"), **out)
    print("qwen3_tts_tiny.npz written", {k: v.shape for k, v in out.items()}, float(np.abs(out["c2w_wav"]).mean()))


SECTIONS = {"logmel": make_logmel, "smart_turn": make_smart_turn, "whisper": make_whisper, "parakeet": make_parakeet, "qwen3_tts": make_qwen3_tts}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    for nme in names:
        SECTIONS[nme]()
