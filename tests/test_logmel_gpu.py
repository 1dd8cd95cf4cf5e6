"""GPU parity: HIP log-mel front-ends (through the C ABI) vs the float64 oracle."""

import numpy as np
import pytest
import torch

from oracle import logmel as O
from speech_to_speech_amd.frontend import LogMelFrontend, pack_pcm
from speech_to_speech_amd.synth import synth_audio

pytestmark = pytest.mark.gpu

# Tolerance: outputs are (log10(mel)+4)/4 (Whisper) or per-feature z-scores (Parakeet); the HIP path
# does the DFT and mel projection as f32 fmaf chains, the oracle in float64.
WHISPER_ATOL = 2e-4
PARAKEET_ATOL = 2e-3


def _report(name, got, ref):
    err = np.abs(got - ref)
    print(f"{name}: max_abs_err={err.max():.3e} mean_abs_err={err.mean():.3e}")
    return err.max()


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_logmel_matches_oracle(n_mels):
    dev = torch.device("cuda:0")
    lens = [80000, 160000, 31337, 1, 480000, 481234, 0]
    audios = [synth_audio(n, i) for i, n in enumerate(lens)]
    fe = LogMelFrontend("whisper", n_mels, max_batch=8, device=dev)
    pcm, ns = pack_pcm(audios, dev, max_len=480000)
    out = fe.whisper(pcm, ns).cpu().numpy()
    assert out.shape == (len(lens), n_mels, 3000)
    worst = 0.0
    for i, a in enumerate(audios):
        ref = O.whisper_logmel(a, n_mels)
        worst = max(worst, _report(f"whisper{n_mels}[n={lens[i]}]", out[i], ref))
    assert worst < WHISPER_ATOL
    # bf16 output is the same values rounded
    out_bf = fe.whisper(pcm, ns, out_dtype=torch.bfloat16).float().cpu().numpy()
    assert np.abs(out_bf - out).max() < 8e-3
    fe.close()


def test_parakeet_logmel_matches_oracle():
    dev = torch.device("cuda:0")
    lens = [160000, 80000, 31337, 480000, 1600]
    audios = [synth_audio(n, 10 + i) for i, n in enumerate(lens)]
    fe = LogMelFrontend("parakeet", 128, max_batch=8, device=dev)
    pcm, ns = pack_pcm(audios, dev)
    t_max = max(lens) // 160 + 1
    out, nf = fe.parakeet(pcm, ns, t_max)
    out = out.cpu().numpy()
    nf = nf.cpu().numpy()
    worst = 0.0
    for i, a in enumerate(audios):
        ref, n_valid = O.parakeet_logmel(a, 128, t_total=t_max)
        assert nf[i] == n_valid
        assert np.all(out[i, n_valid:] == 0)
        worst = max(worst, _report(f"parakeet[n={lens[i]}]", out[i], ref))
    assert worst < PARAKEET_ATOL
    fe.close()
