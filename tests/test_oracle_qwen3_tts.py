"""Pins oracle/qwen3_tts.py to the sibling modules in transformers (fixture tests/golden/qwen3_tts_tiny.npz):
Qwen3OmniMoeCode2Wav, Qwen3OmniMoeTalkerCodePredictorModelForConditionalGeneration.generate (greedy) and the dense
Qwen3 decoder (Qwen3Model).  Against the real Qwen3-TTS package the parity is unpinned (oracle/__init__.py)."""

import os

import numpy as np
import pytest
import torch

from oracle import qwen3_tts as OT
from speech_to_speech_amd.synth_weights import Qwen3TTSDims, qwen3_tts_state_dict


@pytest.fixture(scope="module")
def setup(golden_dir):
    dims = Qwen3TTSDims.test_tiny()
    return np.load(os.path.join(golden_dir, "qwen3_tts_tiny.npz")), dims, qwen3_tts_state_dict(dims, seed=5)


def test_codec_decoder_matches_code2wav(setup):
    gold, dims, sd = setup
    with torch.no_grad():
        wav = OT.codec_decode(sd, dims.codec_cfg(), torch.from_numpy(gold["c2w_codes"]).long())
    assert wav.shape == gold["c2w_wav"].shape == (2, 1, (20 - 1) * 0 + gold["c2w_wav"].shape[-1])
    np.testing.assert_allclose(wav.numpy(), gold["c2w_wav"], atol=1e-4, rtol=0)
    assert np.abs(gold["c2w_wav"]).max() < 1.0     # not saturated: the comparison is meaningful


def test_code_predictor_greedy_matches_sibling_generate(setup):
    gold, dims, sd = setup
    with torch.no_grad():
        codes, _ = OT.code_predictor_greedy(sd, dims.predictor_cfg(), torch.from_numpy(gold["cp_hidden"]), torch.from_numpy(gold["cp_e0"]), dims.n_groups)
    assert codes.tolist() == gold["cp_codes"].tolist()


def test_talker_stack_matches_dense_qwen3(setup):
    gold, dims, sd = setup
    with torch.no_grad():
        h = OT.DecoderStack(sd, "talker.model.", dims.talker_cfg()).forward(torch.from_numpy(gold["talker_in"]))
        h = OT.rms(h, sd["talker.model.norm.weight"], dims.eps)
    np.testing.assert_allclose(h.numpy(), gold["talker_hidden"], atol=2e-4, rtol=0)


def test_incremental_decode_equals_full_forward(setup):
    gold, dims, sd = setup
    x = torch.from_numpy(gold["talker_in"])
    with torch.no_grad():
        full = OT.DecoderStack(sd, "talker.model.", dims.talker_cfg()).forward(x)
        st = OT.DecoderStack(sd, "talker.model.", dims.talker_cfg())
        a = st.forward(x[:, :5])
        rest = [st.forward(x[:, i : i + 1]) for i in range(5, x.shape[1])]
    np.testing.assert_allclose(torch.cat([a] + rest, 1).numpy(), full.numpy(), atol=1e-5)


def test_frame_loop_runs_and_stops_on_eos(setup):
    gold, dims, sd = setup
    g = torch.Generator().manual_seed(3)
    prompt = torch.randn(2, 6, dims.t_hidden, generator=g)
    pad = torch.randn(dims.t_hidden, generator=g)
    with torch.no_grad():
        codes, n = OT.talker_generate(sd, dims.talker_cfg(), dims.predictor_cfg(), prompt, dims.n_groups, 12, dims.codec_eos,
                                      suppress=range(dims.codebook, dims.codec_vocab - 2), pad_embed=pad, min_frames=2)
    assert codes.shape[0] == 2 and codes.shape[2] == dims.n_groups and int(n.max()) == codes.shape[1]
    valid = codes[codes >= 0]
    assert int(valid.max()) < dims.codebook
