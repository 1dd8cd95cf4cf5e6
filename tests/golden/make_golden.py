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


SECTIONS = {"logmel": make_logmel}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    for nme in names:
        SECTIONS[nme]()
