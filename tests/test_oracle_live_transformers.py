"""Live cross-check of oracle/whisper.py against transformers' WhisperForConditionalGeneration on a configuration and on
inputs that are NOT in the committed fixtures (distil-style: more encoder than decoder layers, 80 mels, odd head count,
a 0.3 s clip next to a 4 s one, different seed) - guards against an oracle that only fits tests/golden/.  transformers is
part of the image (it is the third-party package the reference's Whisper path executes); skipped where it is missing."""

import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")

from oracle import logmel as OL  # noqa: E402
from oracle import whisper as OW  # noqa: E402
from speech_to_speech_amd.synth import synth_audio  # noqa: E402
from speech_to_speech_amd.synth_weights import WhisperDims, whisper_state_dict  # noqa: E402

T = dict(eos=119, sot=100, lang_ids=[101, 102, 103], transcribe=104, notimestamps=105,
         suppress=[0, 3, 100, 101, 102, 103, 104, 105], begin_suppress=[7, 119], max_new=10)


@pytest.fixture(scope="module")
def live():
    from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration

    dims = WhisperDims(n_mels=80, d_model=96, n_heads=3, enc_layers=3, dec_layers=1, ffn=160, vocab=120, max_target=40)
    cfg = WhisperConfig(vocab_size=dims.vocab, num_mel_bins=dims.n_mels, d_model=dims.d_model, encoder_layers=dims.enc_layers,
                        decoder_layers=dims.dec_layers, encoder_attention_heads=dims.n_heads, decoder_attention_heads=dims.n_heads,
                        encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn, max_source_positions=1500,
                        max_target_positions=dims.max_target, pad_token_id=T["eos"], bos_token_id=T["eos"], eos_token_id=T["eos"],
                        decoder_start_token_id=T["sot"], suppress_tokens=T["suppress"], begin_suppress_tokens=T["begin_suppress"])
    torch.manual_seed(0)
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = whisper_state_dict(dims, seed=11, gain=2.5, pos_std=1.5)
    missing, unexpected = model.load_state_dict({**sd, "proj_out.weight": sd["model.decoder.embed_tokens.weight"]}, strict=False)
    assert not unexpected and all("proj_out" in m for m in missing)
    gc = model.generation_config
    gc.lang_to_id = {f"<|l{i}|>": v for i, v in enumerate(T["lang_ids"])}
    gc.task_to_id = {"transcribe": T["transcribe"]}
    gc.no_timestamps_token_id = T["notimestamps"]
    gc.is_multilingual = True
    gc.suppress_tokens, gc.begin_suppress_tokens = T["suppress"], T["begin_suppress"]
    audios = [synth_audio(4800, 901), synth_audio(64000, 902)]          # 0.3 s and 4 s
    hf_feats = torch.from_numpy(WhisperFeatureExtractor(feature_size=dims.n_mels)(audios, sampling_rate=16000, return_tensors="np").input_features)
    return dims, sd, model, gc, audios, hf_feats


def test_logmel_and_encoder_match(live):
    dims, sd, model, gc, audios, hf_feats = live
    feats = torch.from_numpy(np.stack([OL.whisper_logmel(a, dims.n_mels) for a in audios]).astype(np.float32))
    assert float((feats - hf_feats).abs().max()) < 2e-4
    with torch.no_grad():
        enc = OW.encode(sd, feats, dims.n_heads, dims.enc_layers)
        ref = model.get_encoder()(hf_feats).last_hidden_state
    assert enc.shape == ref.shape == (2, 1500, dims.d_model)
    assert float((enc - ref).abs().max()) < 5e-4 * max(1.0, float(ref.abs().max()))


def test_language_detection_and_greedy_ids_match(live):
    dims, sd, model, gc, audios, hf_feats = live
    with torch.no_grad():
        enc_o = model.get_encoder()(hf_feats)
        lang = model.detect_language(encoder_outputs=enc_o, generation_config=gc).tolist()
        enc = OW.encode(sd, hf_feats, dims.n_heads, dims.enc_layers)
        _, prompts = OW.greedy_decode(sd, enc, dims.n_heads, dims.dec_layers, [[T["sot"], -1, T["transcribe"], T["notimestamps"]]] * 2,
                                      T["eos"], 1, T["suppress"], T["begin_suppress"], T["lang_ids"])
        assert [p[1] for p in prompts] == lang
        ref = model.generate(hf_feats, generation_config=gc, max_new_tokens=T["max_new"], num_beams=1, do_sample=False,
                             return_timestamps=False, task="transcribe", language="<|l2|>")
        out, _ = OW.greedy_decode(sd, enc, dims.n_heads, dims.dec_layers, [[T["sot"], 103, T["transcribe"], T["notimestamps"]]] * 2,
                                  T["eos"], T["max_new"], T["suppress"], T["begin_suppress"])
    for b in range(2):
        ids = [int(t) for t in ref[b] if int(t) != T["eos"]]
        ids = ids[-len(out[b]):] if len(ids) > len(out[b]) else ids
        assert out[b] == ids, (b, out[b], ref[b].tolist())


def test_parakeet_unseen_lengths_match_transformers():
    """oracle/parakeet.py vs ParakeetForTDT on a seed and on lengths outside the fixtures: a 0.5 s clip (7 encoder frames)
    next to a 6.4 s one, i.e. heavy padding inside the batch (masked convolutions, masked attention, per-session stop)."""
    from transformers import ParakeetForTDT, ParakeetTDTConfig
    from transformers.models.parakeet.configuration_parakeet import ParakeetEncoderConfig

    from oracle import parakeet as OP
    from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

    dims = ParakeetDims.test_tiny()
    sd = parakeet_state_dict(dims, seed=21, gain=1.5)
    enc_cfg = ParakeetEncoderConfig(hidden_size=dims.d_model, num_hidden_layers=dims.layers, num_attention_heads=dims.n_heads,
                                    intermediate_size=dims.ffn, num_mel_bins=dims.n_mels, subsampling_conv_channels=dims.sub_channels)
    cfg = ParakeetTDTConfig(vocab_size=dims.vocab, decoder_hidden_size=dims.dec_hidden, num_decoder_layers=dims.dec_layers,
                            blank_token_id=dims.blank, encoder_config=enc_cfg, pad_token_id=2)
    m = ParakeetForTDT(cfg).eval()
    sdl = dict(sd)
    for l in range(dims.layers):
        sdl[f"encoder.layers.{l}.conv.norm.num_batches_tracked"] = torch.tensor(0)
    m.load_state_dict(sdl, strict=True)
    m.generation_config.decoder_start_token_id = dims.blank
    lens = [8000, 102400]
    audios = [synth_audio(n, 950 + i) for i, n in enumerate(lens)]
    tmax = max(lens) // 160 + 1
    feats = torch.from_numpy(np.stack([OL.parakeet_logmel(a, 128, t_total=tmax)[0] for a in audios]).astype(np.float32))
    flen = torch.tensor([n // 160 for n in lens])
    amask = torch.arange(tmax)[None, :] < flen[:, None]
    with torch.no_grad():
        eo = m.get_audio_features(input_features=feats, attention_mask=amask)
        g = m.generate(input_features=feats, attention_mask=amask, return_dict_in_generate=True)
        enc, elen = OP.encode(sd, feats, flen, dims.n_heads, dims.layers)
        toks, durs, frames = OP.tdt_greedy(sd, enc, elen, dims.vocab, dims.blank)
    assert elen.tolist() == eo.attention_mask.sum(-1).tolist()
    for b in range(2):
        n = int(elen[b])
        np.testing.assert_allclose(enc[b, :n].numpy(), eo.last_hidden_state[b, :n].numpy(), atol=3e-4, rtol=0)
        steps = len(toks[b])
        assert g.sequences[b][1 : 1 + steps].tolist() == toks[b]
        assert g.durations[b][1 : 1 + steps].tolist() == durs[b]


def test_qwen3_tts_siblings_unseen_dims():
    """oracle/qwen3_tts.py vs the sibling modules on dimensions outside the fixture: three decoder blocks (x5, x3, x2),
    GQA 6:2, window 4 < sequence length (the sliding window actually cuts), 6 code groups, a single-frame utterance."""
    from transformers import Qwen3Config, Qwen3Model
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import (Qwen3OmniMoeCode2WavConfig,
                                                                                Qwen3OmniMoeTalkerCodePredictorConfig)
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import (
        Qwen3OmniMoeCode2Wav, Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration)

    from oracle import qwen3_tts as OT
    from speech_to_speech_amd.synth_weights import Qwen3TTSDims, qwen3_tts_state_dict

    dims = Qwen3TTSDims(t_hidden=96, t_layers=2, t_heads=6, t_kv_heads=2, t_head_dim=16, t_ffn=160, codec_vocab=90, codec_eos=88,
                        p_hidden=96, p_layers=1, p_heads=4, p_kv_heads=1, p_head_dim=16, p_ffn=96, n_groups=6, codebook=80,   # predictor width ==
                        # talker width: the sibling has no input projection (that one is a Qwen3-TTS assumption)
                        c_hidden=48, c_layers=2, c_heads=3, c_ffn=64, c_window=4, c_decoder_dim=64, upsample_rates=(5, 3, 2),
                        upsampling_ratios=(2, 2))
    sd = qwen3_tts_state_dict(dims, seed=9)
    g = torch.Generator().manual_seed(4)
    c = Qwen3OmniMoeCode2WavConfig(codebook_size=dims.codebook, hidden_size=dims.c_hidden, num_attention_heads=dims.c_heads,
                                   num_key_value_heads=dims.c_heads, intermediate_size=dims.c_ffn, num_hidden_layers=dims.c_layers,
                                   num_quantizers=dims.n_groups, upsample_rates=list(dims.upsample_rates),
                                   upsampling_ratios=list(dims.upsampling_ratios), decoder_dim=dims.c_decoder_dim,
                                   sliding_window=dims.c_window)
    m = Qwen3OmniMoeCode2Wav(c).eval()
    m.load_state_dict({k[len("code2wav."):]: v for k, v in sd.items() if k.startswith("code2wav.")}, strict=True)
    for frames in (1, 11):
        codes = torch.randint(0, dims.codebook, (2, dims.n_groups, frames), generator=g)
        with torch.no_grad():
            ref = m(codes)
            got = OT.codec_decode(sd, dims.codec_cfg(), codes)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) < 2e-4, frames

    pc = Qwen3OmniMoeTalkerCodePredictorConfig(vocab_size=dims.codebook, hidden_size=dims.p_hidden, intermediate_size=dims.p_ffn,
                                               num_hidden_layers=dims.p_layers, num_attention_heads=dims.p_heads,
                                               num_key_value_heads=dims.p_kv_heads, head_dim=dims.p_head_dim,
                                               num_code_groups=dims.n_groups, rms_norm_eps=dims.eps)
    pm = Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration(pc).eval()
    pm.load_state_dict({k[len("code_predictor."):]: v for k, v in sd.items() if k.startswith("code_predictor.")}, strict=True)
    ph, e0 = torch.randn(4, dims.p_hidden, generator=g), torch.randn(4, dims.p_hidden, generator=g)
    with torch.no_grad():
        r = pm.generate(inputs_embeds=torch.stack([ph, e0], 1), max_new_tokens=dims.n_groups - 1, do_sample=False, return_dict_in_generate=True)
        codes, _ = OT.code_predictor_greedy(sd, dims.predictor_cfg(), ph, e0, dims.n_groups)
    assert codes.tolist() == r.sequences.tolist()

    qc = Qwen3Config(vocab_size=dims.codec_vocab, hidden_size=dims.t_hidden, intermediate_size=dims.t_ffn, num_hidden_layers=dims.t_layers,
                     num_attention_heads=dims.t_heads, num_key_value_heads=dims.t_kv_heads, head_dim=dims.t_head_dim, rms_norm_eps=dims.eps,
                     rope_parameters={"rope_theta": dims.t_theta, "rope_type": "default"}, attention_bias=False, tie_word_embeddings=False)
    qm = Qwen3Model(qc).eval()
    sub = {k[len("talker.model."):]: v for k, v in sd.items() if k.startswith("talker.model.") and "codec_embedding" not in k}
    sub["embed_tokens.weight"] = sd["talker.model.codec_embedding.weight"]
    qm.load_state_dict(sub, strict=True)
    x = torch.randn(3, 7, dims.t_hidden, generator=g)
    with torch.no_grad():
        ref = qm(inputs_embeds=x).last_hidden_state
        h = OT.rms(OT.DecoderStack(sd, "talker.model.", dims.talker_cfg()).forward(x), sd["talker.model.norm.weight"], dims.eps)
    assert float((h - ref).abs().max()) < 3e-4
