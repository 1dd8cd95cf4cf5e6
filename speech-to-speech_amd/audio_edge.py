"""API-edge audio ops on the GPU.  The reference resamples every chunk of every session on the asyncio
thread with ``scipy.signal.resample_poly`` (api/openai_realtime/utils.py:5-12: int16 -> float32 / 32768 -> resample_poly(up, down)
-> clip(y * 32768).astype(int16)), re-chunks client audio into 512-sample VAD blocks (handlers/audio.py:122-149) and base64-encodes
every output chunk (handlers/audio.py:293-330).  ``EdgeResampler`` does the resampling for MANY sessions' chunks in one launch,
bit-identically (same prototype filter, same tap order and separately rounded multiply / add as scipy's upfirdn in float32);
``PcmChunker`` and ``encode_audio_chunks`` restate the host-side bookkeeping for batches."""

from __future__ import annotations

import base64
from math import gcd
from typing import Sequence

import numpy as np

PIPELINE_SAMPLE_RATE = 16000
CHUNK_SAMPLES = 512


def resample_fir(up: int, down: int, dtype=np.float32) -> np.ndarray:
    """The prototype low-pass scipy.signal.resample_poly(x, up, down) designs: firwin(2*half_len+1, 1/max(up,down), kaiser 5.0) * up."""
    from scipy.signal import firwin

    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(dtype)      # scipy casts to the signal's dtype FIRST ...
    h *= dtype(up)                                                                        # ... and applies the gain in that precision
    return h


class EdgeResampler:
    """Batched int16 resampling between a client rate and the pipeline rate on one GPU."""

    def __init__(self, device: str = "cuda:0"):
        import torch

        from . import native

        self.device = torch.device(device)
        self._lib = native.load()
        self._native = native
        self._firs: dict[tuple[int, int], "torch.Tensor"] = {}

    def _fir(self, up: int, down: int):
        import torch

        key = (up, down)
        if key not in self._firs:
            self._firs[key] = torch.from_numpy(resample_fir(up, down)).to(self.device)
        return self._firs[key]

    def resample_batch(self, chunks: Sequence[bytes | np.ndarray], from_rate: int, to_rate: int) -> list[bytes]:
        """One launch for all sessions' chunks: int16 PCM (bytes or arrays, ragged lengths) at from_rate -> int16 PCM bytes at to_rate."""
        import torch

        arrays = [np.frombuffer(c, dtype=np.int16) if isinstance(c, (bytes, bytearray, memoryview)) else np.asarray(c, dtype=np.int16) for c in chunks]
        if from_rate == to_rate or not arrays:
            return [a.tobytes() for a in arrays]
        g = gcd(from_rate, to_rate)
        up, down = to_rate // g, from_rate // g
        lens = [len(a) for a in arrays]
        n_max = max(lens)
        if n_max == 0:
            return [b"" for _ in arrays]
        host = np.zeros((len(arrays), n_max), dtype=np.int16)
        for i, a in enumerate(arrays):
            host[i, : len(a)] = a
        n_out = [-(-n * up // down) for n in lens]
        n_out_max = max(n_out)
        with torch.cuda.device(self.device):
            x = torch.from_numpy(host).to(self.device)
            nin = torch.tensor(lens, dtype=torch.int32).to(self.device)
            y = torch.empty((len(arrays), n_out_max), dtype=torch.int16, device=self.device)
            fir = self._fir(up, down)
            nat = self._native
            nat.check(self._lib.s2s_resample_poly_int16(nat.stream_handle(), nat.ptr(x), 1, n_max, nat.ptr(nin), up, down, nat.ptr(fir), fir.numel(),
                                                        nat.ptr(y), None, n_out_max, n_out_max, len(arrays)), "s2s_resample_poly_int16")
            out = y.cpu().numpy()
        return [out[i, : n_out[i]].tobytes() for i in range(len(arrays))]


class PcmChunker:
    """Per-connection remainder bookkeeping of handlers/audio.py:122-149 ``append_pcm``: 16 kHz int16 bytes -> 512-sample chunks."""

    def __init__(self, chunk_samples: int = CHUNK_SAMPLES):
        self.chunk_bytes = 2 * chunk_samples
        self.remainder = b""
        self.has_data = False

    def append(self, pcm16k: bytes) -> list[bytes]:
        data = self.remainder + pcm16k
        n = len(data) // self.chunk_bytes * self.chunk_bytes
        chunks = [data[i : i + self.chunk_bytes] for i in range(0, n, self.chunk_bytes)]
        self.remainder = data[n:]
        if chunks:
            self.has_data = True
        return chunks


def encode_audio_chunks(chunks: Sequence[bytes]) -> list[str]:
    """base64 payloads of ``response.output_audio.delta`` events (handlers/audio.py:318-319), one per session chunk."""
    return [base64.b64encode(c).decode("ascii") for c in chunks]
