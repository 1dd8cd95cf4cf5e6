"""Oracle: log-mel front-ends in numpy float64 (test infrastructure, see oracle/__init__.py).

Restates
  * transformers/models/whisper/feature_extraction_whisper.py:105-131
    (``_np_extract_fbank_features``: the path WhisperSTTHandler.prepare_model_inputs,
    reference STT/whisper_stt_handler.py:83-87, takes on CPU) together with
    transformers/audio_utils.py ``spectrogram`` (centre/reflect padding, periodic Hann,
    power=2, mel projection, log10 with 1e-10 floor) and ``mel_filter_bank``
    (slaney scale + slaney norm);
  * transformers/models/parakeet/feature_extraction_parakeet.py:101-127 (STFT with a
    symmetric Hann(400) centred in n_fft=512, constant padding, power, mel, log(x+2^-24))
    and :252-278 (pre-emphasis, valid-frame count, per-feature mean / unbiased-std
    normalisation, masking).
"""

from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
HOP = 160
WHISPER_N_FFT = 400
WHISPER_N_SAMPLES = 480000
WHISPER_N_FRAMES = 3000
PARAKEET_N_FFT = 512
PARAKEET_WIN = 400


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    mel = f / f_sp
    log_t = f >= min_log_hz
    mel = np.where(log_t, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)
    return mel


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    f = f_sp * m
    log_t = m >= min_log_mel
    return np.where(log_t, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filter_bank_slaney(n_bins: int, n_mels: int, sr: int = SAMPLE_RATE, fmin: float = 0.0,
                           fmax: float | None = None) -> np.ndarray:
    """[n_bins, n_mels] float64 triangular filters, slaney scale, slaney (area) norm.
    Follows transformers/audio_utils.py mel_filter_bank(norm="slaney", mel_scale="slaney")."""
    fmax = sr / 2.0 if fmax is None else fmax
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0.0, sr / 2.0, n_bins)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2 : n_mels + 2] - hz_pts[:n_mels])
    return fb * enorm[None, :]


def _frames(sig: np.ndarray, n_frames: int, frame_len: int) -> np.ndarray:
    idx = np.arange(frame_len)[None, :] + HOP * np.arange(n_frames)[:, None]
    return sig[idx]


def whisper_logmel(audio: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """float64 [n_mels, 3000] Whisper input features of one utterance (padded/truncated to 30 s)."""
    x = np.zeros(WHISPER_N_SAMPLES, dtype=np.float64)
    n = min(len(audio), WHISPER_N_SAMPLES)
    x[:n] = np.asarray(audio[:n], dtype=np.float64)
    xp = np.pad(x, (WHISPER_N_FFT // 2, WHISPER_N_FFT // 2), mode="reflect")
    window = np.hanning(WHISPER_N_FFT + 1)[:-1]          # periodic Hann
    fr = _frames(xp, WHISPER_N_FRAMES + 1, WHISPER_N_FFT) * window[None, :]
    spec = np.fft.rfft(fr, n=WHISPER_N_FFT, axis=1)
    power = np.abs(spec) ** 2                               # [3001, 201]
    filt = mel_filter_bank_slaney(WHISPER_N_FFT // 2 + 1, n_mels, fmax=8000.0)
    mel = np.maximum(1e-10, power @ filt).T                 # [n_mels, 3001]
    log_spec = np.log10(mel)[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def parakeet_logmel(audio: np.ndarray, n_mels: int = 128, t_total: int | None = None):
    """(features float64 [t_total, n_mels] zero beyond valid frames, n_valid).
    t_total defaults to len(audio)//160 + 1 (what torch.stft(center=True) yields)."""
    x = np.asarray(audio, dtype=np.float32)
    n = len(x)
    # pre-emphasis in float32 exactly as the reference extractor does it
    y = np.concatenate([x[:1], x[1:] - np.float32(0.97) * x[:-1]]).astype(np.float64)
    n_frames = n // HOP + 1
    if t_total is None:
        t_total = n_frames
    yp = np.pad(y, (PARAKEET_N_FFT // 2, PARAKEET_N_FFT // 2 + HOP * max(0, t_total - n_frames)))
    window = np.zeros(PARAKEET_N_FFT)
    off = (PARAKEET_N_FFT - PARAKEET_WIN) // 2
    window[off : off + PARAKEET_WIN] = np.hanning(PARAKEET_WIN)   # symmetric Hann
    fr = _frames(yp, t_total, PARAKEET_N_FFT) * window[None, :]
    power = np.abs(np.fft.rfft(fr, n=PARAKEET_N_FFT, axis=1)) ** 2
    filt = mel_filter_bank_slaney(PARAKEET_N_FFT // 2 + 1, n_mels)
    mel = np.log(power @ filt + 2.0 ** -24)                 # [t_total, n_mels]
    n_valid = n // HOP
    valid = mel[:n_valid]
    mean = valid.sum(0) / n_valid
    var = ((valid - mean) ** 2).sum(0) / (n_valid - 1)
    out = np.zeros_like(mel)
    out[:n_valid] = (valid - mean) / (np.sqrt(var) + 1e-5)
    return out, n_valid


def whisper_logmel_window(audio: np.ndarray, n_mels: int = 80, window_samples: int = WHISPER_N_SAMPLES, normalize: bool = False) -> np.ndarray:
    """float64 [n_mels, window_samples // 160]: WhisperFeatureExtractor(chunk_length=window_samples / 16000) features, optionally
    with do_normalize=True (feature_extraction_whisper.py: zero_mean_unit_var_norm over the valid samples - population variance,
    + 1e-7 under the root, padding left at 0 - applied BEFORE the STFT).  The Smart-Turn feature path of the reference
    (VAD/smart_turn.py:130-143) is window_samples = 128000, normalize = True, n_mels = 80."""
    n_frames = window_samples // HOP
    x = np.zeros(window_samples, dtype=np.float64)
    n = min(len(audio), window_samples)
    v = np.asarray(audio[:n], dtype=np.float32)
    if normalize and n > 0:
        v = ((v - v.mean()) / np.sqrt(v.var() + 1e-7)).astype(np.float32)      # the extractor normalises in float32
    x[:n] = v.astype(np.float64)
    xp = np.pad(x, (WHISPER_N_FFT // 2, WHISPER_N_FFT // 2), mode="reflect")
    window = np.hanning(WHISPER_N_FFT + 1)[:-1]
    fr = _frames(xp, n_frames + 1, WHISPER_N_FFT) * window[None, :]
    power = np.abs(np.fft.rfft(fr, n=WHISPER_N_FFT, axis=1)) ** 2
    filt = mel_filter_bank_slaney(WHISPER_N_FFT // 2 + 1, n_mels, fmax=8000.0)
    log_spec = np.log10(np.maximum(1e-10, power @ filt).T)[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def smart_turn_prepare_audio(audio: np.ndarray, max_samples: int = 128000) -> np.ndarray:
    """reference VAD/smart_turn.py:103-128 at 16 kHz: keep the LAST 8 s, left-pad shorter audio with zeros."""
    a = np.asarray(audio, dtype=np.float32)
    if a.size > max_samples:
        return a[-max_samples:]
    return np.pad(a, (max_samples - a.size, 0), mode="constant")
