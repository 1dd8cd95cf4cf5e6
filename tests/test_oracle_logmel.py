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
