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


def test_smart_turn_features_match_oracle_and_golden(golden_dir):
    """The Smart-Turn feature path (reference VAD/smart_turn.py:130-143) on the device: last 8 s / left padding on the host, waveform
    normalisation + 800-frame Whisper log-mel in the kernels; vs the float64 oracle and vs the WhisperFeatureExtractor fixture."""
    import os

    from speech_to_speech_amd.smart_turn import MI355XSmartTurnAnalyzer, SmartTurnFeatures

    gold = np.load(os.path.join(golden_dir, "smart_turn.npz"))
    lens = [int(n) for n in gold["lens"]] + [0, 1, 127999, 300000]
    audios = []
    for i, n in enumerate(lens):
        x = synth_audio(n, session_idx=200 + i)
        audios.append(x * 0.01 + 0.05 if i == 2 else x)
    st = SmartTurnFeatures(max_batch=4, device="cuda:0")              # 7 utterances through batches of 4
    out = st(audios).cpu().numpy()
    assert out.shape == (len(lens), 80, 800)
    worst = 0.0
    for i, a in enumerate(audios):
        ref = O.whisper_logmel_window(O.smart_turn_prepare_audio(a), 80, 128000, normalize=True)
        worst = max(worst, _report(f"smart_turn[n={lens[i]}]", out[i], ref))
    assert worst < 3e-4
    for i in range(3):
        assert np.abs(out[i][:, ::4] - gold[f"feat_{i}"]).max() < 3e-4 and np.abs(out[i][:, -64:] - gold[f"tail_{i}"]).max() < 3e-4
    st.close()
    # the un-normalised short window agrees with the 30 s path on the frames both cover, when the clip's maximum is in the window
    fe = LogMelFrontend("whisper", 128, max_batch=2, device="cuda:0")
    a = synth_audio(100000, 3)
    pcm, ns = pack_pcm([a], torch.device("cuda:0"))
    short = fe.whisper_window(pcm, ns, 128000).cpu().numpy()[0]
    np.testing.assert_allclose(short, O.whisper_logmel_window(a, 128, 128000), atol=WHISPER_ATOL)
    fe.close()
    # analyzer: the reference's predict() contract with an injected host classifier
    seen = []

    def classifier(f):
        seen.append(f.shape)
        return f.mean(axis=(1, 2)) * 0 + np.linspace(0.2, 0.9, f.shape[0])

    an = MI355XSmartTurnAnalyzer(classifier, threshold=0.5, max_batch=4, warmup=True)
    res = an.predict_batch(audios[:3])
    assert [r.complete for r in res] == [False, True, True] and seen[-1] == (3, 80, 800) and seen[0] == (1, 80, 800)
    one = an.predict(audios[0])
    assert one.complete is False and abs(one.probability - 0.2) < 1e-9 and one.inference_ms > 0
    an.features.close()

    # the reference's own predict() test (tests/test_smart_turn.py:208-236) with its fake ONNX session behind the adapter
    class FakeSession:
        def __init__(self):
            self.feeds = []

        def run(self, _outputs, feeds):
            self.feeds.append(feeds)
            return [np.array([[0.75]], dtype=np.float32)]

    sess = FakeSession()
    an2 = MI355XSmartTurnAnalyzer(lambda f: sess.run(None, {"input_features": f})[0], threshold=0.5, max_batch=2, warmup=False)
    result = an2.predict(np.ones(100, dtype=np.float32))
    assert result.complete is True and result.probability == 0.75
    feats = sess.feeds[0]["input_features"]
    assert feats.shape == (1, 80, 800) and feats.dtype == np.float32
    an2.features.close()


def test_smart_turn_tick_batcher_one_launch_same_features():
    """B sessions whose `predict()` calls fall into one tick are served by ONE `s2s_logmel_whisper_window` launch (the classifier sees one
    [B, 80, 800] batch) and get exactly the features B single calls produce."""
    import threading

    from speech_to_speech_amd.smart_turn import MI355XSmartTurnAnalyzer, SmartTurnFeatures

    rng = np.random.default_rng(5)
    lens = [16000, 128000, 40000, 200000, 9000]
    audios = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    st = SmartTurnFeatures(max_batch=8, device="cuda:0")
    single = [st([a]).cpu().numpy()[0] for a in audios]
    seen = []

    def classifier(f):
        seen.append(np.array(f))
        return f.reshape(f.shape[0], -1).mean(1)

    an = MI355XSmartTurnAnalyzer(classifier, threshold=0.5, max_batch=8, warmup=False, tick_ms=500.0, features=st)      # (a long tick: five thread starts fit into it on a busy box too)
    res = {}
    ths = [threading.Thread(target=lambda i=i: res.__setitem__(i, an.predict(audios[i]))) for i in range(5)]
    [t.start() for t in ths]
    [t.join(timeout=30) for t in ths]
    assert len(res) == 5 and len(seen) == 1 and seen[0].shape == (5, 80, 800), [x.shape for x in seen]
    want = sorted(float(s.mean()) for s in single)
    assert np.allclose(sorted(r.probability for r in res.values()), want, rtol=0, atol=1e-7)
    for row in seen[0]:
        assert any(np.array_equal(row, s) for s in single)            # bit-identical to the B = 1 launches
    an.close()
