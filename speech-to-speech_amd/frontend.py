"""Host side of the log-mel front-ends (device work is in csrc/logmel.hip).

Mirrors what the reference obtains from ``processor(audio, sampling_rate=16000)``
(STT/whisper_stt_handler.py:83-87) and from nano_parakeet's internal feature
extraction (STT/parakeet_tdt_handler.py:621-623): float32 mono 16 kHz PCM in,
model input features out - but batched over sessions and resident on the GPU.
"""

from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import native

SAMPLE_RATE = 16000
HOP = 160
WHISPER_FRAMES = 3000
WHISPER_SAMPLES = 480000


def slaney_mel_filters(n_bins: int, n_mels: int, sr: int = SAMPLE_RATE, fmin: float = 0.0,
                       fmax: float | None = None) -> np.ndarray:
    """float32 [n_bins, n_mels] triangular mel filters (slaney mel scale, area-normalised).

    Same construction as the extractor configs the reference relies on
    (WhisperFeatureExtractor: 0-8000 Hz; Parakeet: 0-sr/2), computed in float64.
    """
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    hz = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    freqs = np.linspace(0.0, sr / 2.0, n_bins)
    diff = np.diff(hz)
    slopes = hz[None, :] - freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    fb *= (2.0 / (hz[2 : n_mels + 2] - hz[:n_mels]))[None, :]
    return np.ascontiguousarray(fb, dtype=np.float32)


def pack_pcm(audios: Sequence[np.ndarray], device: torch.device, max_len: int | None = None):
    """Ragged float32 utterances -> (pcm [B, stride] on device, n_samples int32 [B] on device)."""
    lens = [min(len(a), max_len) if max_len else len(a) for a in audios]
    stride = max(8, (max(lens) + 7) // 8 * 8) if lens else 8
    host = torch.zeros((len(audios), stride), dtype=torch.float32, pin_memory=torch.cuda.is_available())
    for i, (a, n) in enumerate(zip(audios, lens)):
        host[i, :n] = torch.from_numpy(np.ascontiguousarray(a[:n], dtype=np.float32))
    pcm = host.to(device, non_blocking=True)
    n_samples = torch.tensor(lens, dtype=torch.int32).to(device, non_blocking=True)
    return pcm, n_samples


class LogMelFrontend:
    """Owns one native front-end handle (DFT basis, mel weights, scratch) on one GPU."""

    def __init__(self, kind: str, n_mels: int, max_batch: int = 64, device: str | torch.device = "cuda:0"):
        if kind not in ("whisper", "parakeet"):
            raise ValueError(f"unknown front-end kind {kind!r}")
        self.kind = kind
        self.n_mels = int(n_mels)
        self.max_batch = int(max_batch)
        self.device = torch.device(device)
        self._lib = native.load()
        if kind == "whisper":
            filt = slaney_mel_filters(201, n_mels, fmax=8000.0)
            code = native.FRONTEND_WHISPER
        else:
            filt = slaney_mel_filters(257, n_mels)
            code = native.FRONTEND_PARAKEET
        self.mel_filters = filt
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_frontend_create(code, self.n_mels, filt.ctypes.data, self.max_batch, C.byref(handle)),
                "s2s_frontend_create",
            )
        self._handle = handle

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.s2s_frontend_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def whisper(self, pcm: torch.Tensor, n_samples: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """pcm [B, stride] f32 + n_samples [B] i32 (both on the GPU) -> [B, n_mels, 3000]."""
        assert self.kind == "whisper"
        B = pcm.shape[0]
        out = torch.empty((B, self.n_mels, WHISPER_FRAMES), dtype=out_dtype, device=self.device)
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_logmel_whisper(self._handle, native.stream_handle(), native.ptr(pcm), pcm.stride(0),
                                             native.ptr(n_samples), B, native.ptr(out), native.dtype_code(out_dtype)),
                "s2s_logmel_whisper",
            )
        return out

    def whisper_window(self, pcm: torch.Tensor, n_samples: torch.Tensor, window_samples: int, normalize: bool = False,
                       out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """Features of a shorter window (Smart-Turn: 8 s, normalised waveform) -> [B, n_mels, window_samples // 160]."""
        assert self.kind == "whisper"
        B = pcm.shape[0]
        out = torch.empty((B, self.n_mels, int(window_samples) // HOP), dtype=out_dtype, device=self.device)
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_logmel_whisper_window(self._handle, native.stream_handle(), native.ptr(pcm), pcm.stride(0), native.ptr(n_samples), B,
                                                    int(window_samples), int(bool(normalize)), native.ptr(out), native.dtype_code(out_dtype)),
                "s2s_logmel_whisper_window",
            )
        return out

    def parakeet(self, pcm: torch.Tensor, n_samples: torch.Tensor, t_max: int,
                 out_dtype: torch.dtype = torch.float32, out: torch.Tensor | None = None, n_frames: torch.Tensor | None = None):
        """-> (features [B, t_max, n_mels] zero beyond valid frames, n_frames int32 [B]); `out` / `n_frames`: write into these rows of a
        larger batch instead of new tensors."""
        assert self.kind == "parakeet"
        B = pcm.shape[0]
        if out is None:
            out = torch.empty((B, t_max, self.n_mels), dtype=out_dtype, device=self.device)
        if n_frames is None:
            n_frames = torch.empty((B,), dtype=torch.int32, device=self.device)
        assert out.is_contiguous() and out.shape == (B, t_max, self.n_mels) and out.dtype == out_dtype and n_frames.is_contiguous()
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_logmel_parakeet(self._handle, native.stream_handle(), native.ptr(pcm), pcm.stride(0),
                                              native.ptr(n_samples), B, int(t_max), native.ptr(out),
                                              native.dtype_code(out_dtype), native.ptr(n_frames)),
                "s2s_logmel_parakeet",
            )
        return out, n_frames

    def parakeet_cached_rows(self, pcm_rows: torch.Tensor, cache_rows: torch.Tensor, n_samples: torch.Tensor, first_frame: torch.Tensor,
                             new_pcm: torch.Tensor | None, new_off: torch.Tensor | None, old_n: torch.Tensor | None, t_max: int, out: torch.Tensor,
                             n_frames: torch.Tensor) -> None:
        """The growing windows of a whole tick (s2s_logmel_parakeet_cached_rows): pcm_rows / cache_rows are int64 device tensors of B device
        pointers, new_pcm the flat block of all rows' new samples (row b: new_off[b] .. new_off[b + 1], appended at old_n[b])."""
        assert self.kind == "parakeet" and out.is_contiguous() and pcm_rows.dtype == torch.int64 and cache_rows.dtype == torch.int64
        B = int(pcm_rows.numel())
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_logmel_parakeet_cached_rows(self._handle, native.stream_handle(), B, int(t_max), native.ptr(pcm_rows), native.ptr(cache_rows),
                                                          native.ptr(n_samples), native.ptr(first_frame), native.ptr(new_pcm) if new_pcm is not None else None,
                                                          native.ptr(new_off) if new_off is not None else None, native.ptr(old_n) if old_n is not None else None,
                                                          native.ptr(out), native.dtype_code(out.dtype), native.ptr(n_frames)),
                "s2s_logmel_parakeet_cached_rows",
            )

    def parakeet_cached(self, pcm_row: torch.Tensor, n_samples: torch.Tensor, first_frame: torch.Tensor, t_max: int, cache: torch.Tensor,
                        out_row: torch.Tensor, n_frames_row: torch.Tensor) -> None:
        """One growing window (s2s_logmel_parakeet_cached, B = 1): pcm_row [1][cap] holds the window's samples so far, cache [cap_frames][n_mels]
        float32 its un-normalised log-mel frames; frames from first_frame on are computed, the normalised features go to out_row [1][t_max][n_mels]."""
        assert self.kind == "parakeet" and cache.dtype == torch.float32 and cache.is_contiguous() and out_row.is_contiguous()
        with torch.cuda.device(self.device):
            native.check(
                self._lib.s2s_logmel_parakeet_cached(self._handle, native.stream_handle(), native.ptr(pcm_row), pcm_row.stride(0), native.ptr(n_samples),
                                                     native.ptr(first_frame), 1, int(t_max), native.ptr(cache), native.ptr(out_row),
                                                     native.dtype_code(out_row.dtype), native.ptr(n_frames_row)),
                "s2s_logmel_parakeet_cached",
            )
