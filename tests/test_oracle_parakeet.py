"""Pins oracle/parakeet.py to transformers' ParakeetForTDT (fixture tests/golden/parakeet_tiny.npz)."""

import os

import numpy as np
import pytest
import torch

from oracle import logmel as OL
from oracle import parakeet as OP
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict


@pytest.mark.parametrize("tag,seed,gain", [("a", 12, 1.5), ("b", 11, 1.5), ("stuck", 13, 2.0)])
def test_encoder_and_tdt_greedy_match_transformers(golden_dir, tag, seed, gain):
    gold = np.load(os.path.join(golden_dir, "parakeet_tiny.npz"))
    dims = ParakeetDims.test_tiny()
    sd = parakeet_state_dict(dims, seed=seed, gain=gain)
    lens = gold["lens"].tolist()
    audios = [synth_audio(n, 300 + i) for i, n in enumerate(lens)]
    tmax = max(lens) // 160 + 1
    feats = torch.from_numpy(np.stack([OL.parakeet_logmel(a, 128, t_total=tmax)[0] for a in audios]).astype(np.float32))
    flen = torch.tensor([n // 160 for n in lens])
    with torch.no_grad():
        enc, elen = OP.encode(sd, feats, flen, dims.n_heads, dims.layers)
        toks, durs, frames = OP.tdt_greedy(sd, enc, elen, dims.vocab, dims.blank)
    assert elen.tolist() == gold[f"{tag}_enc_len"].tolist()
    for b in range(2):
        n = int(elen[b])
        ref = gold[f"{tag}_enc"][b]
        got = enc[b, ::3].numpy()
        k = (n + 2) // 3
        np.testing.assert_allclose(got[:k], ref[:k], atol=2e-4, rtol=0)
        steps = len(toks[b])
        assert gold[f"{tag}_seq"][b][1 : 1 + steps].tolist() == toks[b]
        assert gold[f"{tag}_dur"][b][1 : 1 + steps].tolist() == durs[b]
        if tag != "stuck":
            assert frames[b][-1] + durs[b][-1] >= n     # walked off the end of the encoder output
