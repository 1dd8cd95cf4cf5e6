"""CPU oracle of the API-edge resampler - TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product).

Restates the reference's own function, src/speech_to_speech/api/openai_realtime/utils.py:5-12 (``resample(audio_int16, from_rate, to_rate)``):
int16 bytes -> float32 / 32768 -> scipy.signal.resample_poly(up = to / gcd, down = from / gcd) -> clip(x * 32768) -> int16 bytes."""

import numpy as np


def reference_resample(audio_int16: bytes, from_rate: int, to_rate: int) -> bytes:
    from scipy.signal import resample_poly

    if from_rate == to_rate:
        return audio_int16
    samples = np.frombuffer(audio_int16, dtype=np.int16).astype(np.float32) / 32768.0
    g = np.gcd(to_rate, from_rate)
    resampled = resample_poly(samples, up=to_rate // g, down=from_rate // g)
    return np.clip(resampled * 32768, -32768, 32767).astype(np.int16).tobytes()
