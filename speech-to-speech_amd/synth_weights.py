"""Deterministic synthetic checkpoints (no network, no weights on disk - SURVEY.md 8d).

Parameter names and shapes are the checkpoint's own (transformers naming), values come from
``np.random.default_rng`` seeded by (seed, crc32(name)) so that any consumer - the MI355X
engine, the oracle, or the third-party model used to make golden fixtures - regenerates the
identical tensors without shipping them.
"""

from __future__ import annotations

import zlib
from dataclasses import dataclass

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _normal(seed, name, shape, std):
    return torch.from_numpy((_rng(seed, name).standard_normal(shape) * std).astype(np.float32))


@dataclass
class WhisperDims:
    n_mels: int = 80
    d_model: int = 384
    n_heads: int = 6
    enc_layers: int = 4
    dec_layers: int = 4
    ffn: int = 1536
    vocab: int = 51864
    max_target: int = 448
    max_source: int = 1500

    @staticmethod
    def tiny_en() -> "WhisperDims":
        return WhisperDims()

    @staticmethod
    def large_v3() -> "WhisperDims":
        return WhisperDims(n_mels=128, d_model=1280, n_heads=20, enc_layers=32, dec_layers=32, ffn=5120, vocab=51866)

    @staticmethod
    def distil_large_v3() -> "WhisperDims":
        return WhisperDims(n_mels=128, d_model=1280, n_heads=20, enc_layers=32, dec_layers=2, ffn=5120, vocab=51866)

    @staticmethod
    def test_tiny() -> "WhisperDims":
        """Small enough for CPU fixtures, still D=64 heads so the MFMA attention path is exercised."""
        return WhisperDims(n_mels=80, d_model=128, n_heads=2, enc_layers=2, dec_layers=2, ffn=256, vocab=320)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Whisper's fixed encoder position table (modeling_whisper.py:55-56 semantics)."""
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2))
    t = np.arange(length)[:, None] * inv[None, :]
    return torch.from_numpy(np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32))


def whisper_state_dict(dims: WhisperDims, seed: int = 0, gain: float = 1.0, pos_std: float = 0.1) -> dict[str, torch.Tensor]:
    """float32 state dict with transformers' WhisperForConditionalGeneration names."""
    d, f = dims.d_model, dims.ffn
    sd: dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = _normal(seed, name + ".weight", (out_f, in_f), gain / np.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = _normal(seed, name + ".bias", (out_f,), 0.05)

    def ln(name):
        sd[name + ".weight"] = 1.0 + _normal(seed, name + ".weight", (d,), 0.1)
        sd[name + ".bias"] = _normal(seed, name + ".bias", (d,), 0.05)

    e = "model.encoder."
    sd[e + "conv1.weight"] = _normal(seed, e + "conv1.weight", (d, dims.n_mels, 3), gain / np.sqrt(3 * dims.n_mels))
    sd[e + "conv1.bias"] = _normal(seed, e + "conv1.bias", (d,), 0.05)
    sd[e + "conv2.weight"] = _normal(seed, e + "conv2.weight", (d, d, 3), gain / np.sqrt(3 * d))
    sd[e + "conv2.bias"] = _normal(seed, e + "conv2.bias", (d,), 0.05)
    sd[e + "embed_positions.weight"] = sinusoids(dims.max_source, d)
    for l in range(dims.enc_layers):
        p = f"{e}layers.{l}."
        ln(p + "self_attn_layer_norm")
        lin(p + "self_attn.q_proj", d, d)
        lin(p + "self_attn.k_proj", d, d, bias=False)
        lin(p + "self_attn.v_proj", d, d)
        lin(p + "self_attn.out_proj", d, d)
        ln(p + "final_layer_norm")
        lin(p + "fc1", f, d)
        lin(p + "fc2", d, f)
    ln(e + "layer_norm")
    dd = "model.decoder."
    sd[dd + "embed_tokens.weight"] = _normal(seed, dd + "embed_tokens.weight", (dims.vocab, d), 0.5)
    sd[dd + "embed_positions.weight"] = _normal(seed, dd + "embed_positions.weight", (dims.max_target, d), pos_std)
    for l in range(dims.dec_layers):
        p = f"{dd}layers.{l}."
        ln(p + "self_attn_layer_norm")
        for a in ("self_attn", "encoder_attn"):
            lin(p + a + ".q_proj", d, d)
            lin(p + a + ".k_proj", d, d, bias=False)
            lin(p + a + ".v_proj", d, d)
            lin(p + a + ".out_proj", d, d)
        ln(p + "encoder_attn_layer_norm")
        ln(p + "final_layer_norm")
        lin(p + "fc1", f, d)
        lin(p + "fc2", d, f)
    ln(dd + "layer_norm")
    return sd
