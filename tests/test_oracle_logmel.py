"""Pins oracle/logmel.py to the third-party extractors (fixtures made by tests/golden/make_golden.py
from transformers' WhisperFeatureExtractor / ParakeetFeatureExtractor) and checks the product's host-side
filter bank against the oracle's."""

import os

import numpy as np
import pytest

from oracle import logmel as O
from speech_to_speech_amd.synth import synth_audio


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "logmel.npz"))


@pytest.mark.parametrize("i,nm", [(0, 80), (1, 128)])
def test_whisper_oracle_matches_transformers(gold, i, nm):
    n = int(gold["lens"][i])
    x = synth_audio(n, session_idx=100 + i)
    ref = gold[f"whisper{nm}_{i}"]
    got = O.whisper_logmel(x, nm)
    assert got.shape == (nm, 3000)
    np.testing.assert_allclose(got[:, : ref.shape[1]], ref, atol=2e-5, rtol=0)
    # everything after the signal is the constant floor value
    np.testing.assert_allclose(got[:, ref.shape[1] :], float(gold[f"whisper{nm}_{i}_floor"]), atol=2e-5)
    assert abs(got.sum() - float(gold[f"whisper{nm}_{i}_sum"])) < 0.5


@pytest.mark.parametrize("i", [0, 1])
def test_parakeet_oracle_matches_transformers(gold, i):
    n = int(gold["lens"][i])
    x = synth_audio(n, session_idx=100 + i)
    ref = gold[f"parakeet128_{i}"]
    got, n_valid = O.parakeet_logmel(x, 128)
    assert n_valid == int(gold[f"parakeet128_{i}_valid"]) == n // 160
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=5e-4, rtol=0)
    assert np.all(got[n_valid:] == 0)


def test_whisper_oracle_truncates_and_pads():
    long = synth_audio(480000 + 1234, 3)
    a = O.whisper_logmel(long, 80)
    b = O.whisper_logmel(long[:480000], 80)
    np.testing.assert_array_equal(a, b)
    empty = O.whisper_logmel(np.zeros(0, np.float32), 80)
    assert np.allclose(empty, empty[0, 0])   # silence: one constant


def test_product_filter_bank_equals_oracle():
    from speech_to_speech_amd.frontend import slaney_mel_filters

    for nb, nm, fmax in [(201, 80, 8000.0), (201, 128, 8000.0), (257, 128, None)]:
        a = slaney_mel_filters(nb, nm, fmax=fmax)
        b = O.mel_filter_bank_slaney(nb, nm, fmax=fmax)
        assert a.dtype == np.float32 and a.shape == (nb, nm)
        np.testing.assert_allclose(a, b, atol=1e-9)


def _smart_turn_input(i, n):
    x = synth_audio(n, session_idx=200 + i)
    return x * 0.01 + 0.05 if i == 2 else x


@pytest.mark.parametrize("i", [0, 1, 2])
def test_smart_turn_oracle_matches_whisper_feature_extractor(golden_dir, i):
    """oracle.whisper_logmel_window(8 s, normalised) == WhisperFeatureExtractor(chunk_length=8)(do_normalize=True) on the audio
    the reference's SmartTurnAnalyzer._prepare_audio hands it (fixture: tests/golden/make_golden.py smart_turn)."""
    gold = np.load(os.path.join(golden_dir, "smart_turn.npz"))
    n = int(gold["lens"][i])
    a = O.smart_turn_prepare_audio(_smart_turn_input(i, n))
    assert a.shape == (128000,)
    got = O.whisper_logmel_window(a, 80, 128000, normalize=True)
    assert got.shape == (80, 800)
    np.testing.assert_allclose(got[:, ::4], gold[f"feat_{i}"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(got[:, -64:], gold[f"tail_{i}"], atol=3e-5, rtol=0)
    assert abs(got.sum() - float(gold[f"sum_{i}"])) < 0.2


def test_windowed_oracle_equals_the_30s_oracle_at_full_window():
    x = synth_audio(30000, 5)
    np.testing.assert_array_equal(O.whisper_logmel_window(x, 80), O.whisper_logmel(x, 80))


def test_smart_turn_prepare_audio_keeps_the_last_eight_seconds():
    from speech_to_speech_amd.smart_turn import prepare_audio

    long = synth_audio(128000 + 999, 1)
    short = synth_audio(5000, 2)
    for a in (long, short):
        np.testing.assert_array_equal(prepare_audio(a), O.smart_turn_prepare_audio(a))
    assert np.array_equal(prepare_audio(long), long[-128000:]) and np.all(prepare_audio(short)[:-5000] == 0)
    with pytest.raises(ValueError):
        prepare_audio(np.zeros((2, 10), np.float32))
    # other rates are resampled on the host exactly as the reference does (scipy.signal.resample_poly)
    assert prepare_audio(np.zeros(48000, np.float32), 48000).shape == (128000,)


def test_smart_turn_prepare_audio_known_answers_of_the_reference_tests():
    """The reference's own cases (tests/test_smart_turn.py:131-148) on the adapter's `prepare_audio`."""
    from speech_to_speech_amd.smart_turn import MAX_AUDIO_SECONDS, MODEL_SAMPLE_RATE, prepare_audio

    max_samples = MAX_AUDIO_SECONDS * MODEL_SAMPLE_RATE
    short_audio = np.arange(100, dtype=np.float32)
    prepared_short = prepare_audio(short_audio, MODEL_SAMPLE_RATE)
    assert prepared_short.shape == (max_samples,)
    np.testing.assert_array_equal(prepared_short[-100:], short_audio)
    assert np.count_nonzero(prepared_short[:-100]) == 0
    long_audio = np.arange(max_samples + 10, dtype=np.float32)
    np.testing.assert_array_equal(prepare_audio(long_audio, MODEL_SAMPLE_RATE), long_audio[-max_samples:])
    prepared = prepare_audio(np.ones(8000, dtype=np.float32), sample_rate=8000)
    assert prepared.shape == (max_samples,) and np.count_nonzero(prepared[-MODEL_SAMPLE_RATE:]) > 0
    with pytest.raises(ValueError):
        prepare_audio(np.ones(10, np.float32), 0)
