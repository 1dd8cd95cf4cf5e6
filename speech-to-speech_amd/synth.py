"""Deterministic synthetic inputs shared by the tests and bench.py.

Audio: ``rng = np.random.default_rng(1234 + session_idx)``; 0.1-sigma white noise through a
one-pole low-pass plus a 220 Hz tone at 0.05, clipped to [-1, 1], mono float32 @ 16 kHz.
"""

from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000


def synth_audio(n_samples: int, session_idx: int = 0) -> np.ndarray:
    rng = np.random.default_rng(1234 + session_idx)
    noise = 0.1 * rng.standard_normal(n_samples)
    # one-pole low-pass y[n] = a*y[n-1] + (1-a)*x[n], a = 0.7, evaluated with a cumulative form
    a = 0.7
    y = np.empty(n_samples, dtype=np.float64)
    acc = 0.0
    # vectorised IIR via scipy when available, plain loop fallback kept for tiny inputs
    try:
        from scipy.signal import lfilter

        y = lfilter([1.0 - a], [1.0, -a], noise)
    except Exception:  # pragma: no cover
        for i in range(n_samples):
            acc = a * acc + (1.0 - a) * noise[i]
            y[i] = acc
    t = np.arange(n_samples) / SAMPLE_RATE
    y = y * 2.0 + 0.05 * np.sin(2.0 * np.pi * 220.0 * t)
    return np.clip(y, -1.0, 1.0).astype(np.float32)
