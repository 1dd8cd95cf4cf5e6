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


@dataclass
class ParakeetDims:
    n_mels: int = 128
    d_model: int = 1024
    n_heads: int = 8
    layers: int = 24
    ffn: int = 4096
    sub_channels: int = 256
    conv_kernel: int = 9
    vocab: int = 8193          # incl. blank (last id)
    dec_hidden: int = 640
    dec_layers: int = 2
    n_durations: int = 5
    scale_input: bool = True

    @property
    def blank(self) -> int:
        return self.vocab - 1

    @staticmethod
    def tdt_0_6b() -> "ParakeetDims":
        return ParakeetDims()

    @staticmethod
    def test_tiny() -> "ParakeetDims":
        return ParakeetDims(d_model=256, n_heads=2, layers=2, ffn=512, sub_channels=32, vocab=65, dec_hidden=64)


def parakeet_state_dict(dims: ParakeetDims, seed: int = 0, gain: float = 1.0) -> dict[str, torch.Tensor]:
    """float32 state dict with transformers' ParakeetForTDT parameter names."""
    d, f, C = dims.d_model, dims.ffn, dims.sub_channels
    D = d // dims.n_heads
    sd: dict[str, torch.Tensor] = {}

    def t(name, shape, std):
        sd[name] = _normal(seed, name, shape, std)

    def lin(name, o, i, bias=True, g=gain):
        t(name + ".weight", (o, i), g / np.sqrt(i))
        if bias:
            t(name + ".bias", (o,), 0.05)

    def ln(name, n=d):
        sd[name + ".weight"] = 1.0 + _normal(seed, name + ".weight", (n,), 0.1)
        t(name + ".bias", (n,), 0.05)

    s = "encoder.subsampling."
    t(s + "layers.0.weight", (C, 1, 3, 3), gain / 3.0); t(s + "layers.0.bias", (C,), 0.05)
    for i in (2, 5):
        t(s + f"layers.{i}.weight", (C, 1, 3, 3), gain / 3.0); t(s + f"layers.{i}.bias", (C,), 0.05)
        t(s + f"layers.{i + 1}.weight", (C, C, 1, 1), gain / np.sqrt(C)); t(s + f"layers.{i + 1}.bias", (C,), 0.05)
    lin(s + "linear", d, C * (dims.n_mels // 8))
    for l in range(dims.layers):
        p = f"encoder.layers.{l}."
        for ff in ("feed_forward1", "feed_forward2"):
            lin(p + ff + ".linear1", f, d); lin(p + ff + ".linear2", d, f)
        a = p + "self_attn."
        t(a + "bias_u", (dims.n_heads, D), 0.2); t(a + "bias_v", (dims.n_heads, D), 0.2)
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(a + n, d, d)
        lin(a + "relative_k_proj", d, d, bias=False)
        c = p + "conv."
        t(c + "pointwise_conv1.weight", (2 * d, d, 1), gain / np.sqrt(d)); t(c + "pointwise_conv1.bias", (2 * d,), 0.05)
        t(c + "depthwise_conv.weight", (d, 1, dims.conv_kernel), gain / 3.0); t(c + "depthwise_conv.bias", (d,), 0.05)
        sd[c + "norm.weight"] = 1.0 + _normal(seed, c + "norm.weight", (d,), 0.1)
        t(c + "norm.bias", (d,), 0.05)
        t(c + "norm.running_mean", (d,), 0.1)
        sd[c + "norm.running_var"] = 1.0 + 0.3 * torch.rand(d, generator=torch.Generator().manual_seed(seed * 1000 + l))
        t(c + "pointwise_conv2.weight", (d, d, 1), gain / np.sqrt(d)); t(c + "pointwise_conv2.bias", (d,), 0.05)
        for n in ("norm_feed_forward1", "norm_self_att", "norm_conv", "norm_feed_forward2", "norm_out"):
            ln(p + n)
    Hd = dims.dec_hidden
    lin("encoder_projector", Hd, d)
    t("decoder.embedding.weight", (dims.vocab, Hd), 1.0)
    for l in range(dims.dec_layers):
        t(f"decoder.lstm.weight_ih_l{l}", (4 * Hd, Hd), 1.5 / np.sqrt(Hd)); t(f"decoder.lstm.weight_hh_l{l}", (4 * Hd, Hd), 1.5 / np.sqrt(Hd))
        t(f"decoder.lstm.bias_ih_l{l}", (4 * Hd,), 0.1); t(f"decoder.lstm.bias_hh_l{l}", (4 * Hd,), 0.1)
    lin("decoder.decoder_projector", Hd, Hd, g=2.0)
    lin("joint.head", dims.vocab + dims.n_durations, Hd, g=2.0)
    # keep the duration logits below every token logit: transformers' generic greedy loop takes the argmax over the
    # concatenated [tokens | durations] vector, so a checkpoint whose duration logits could win is not decodable there
    sd["joint.head.bias"][dims.vocab:] -= 30.0
    # make the synthetic transducer behave like a trained one: blank is common, zero-duration emissions are rare
    sd["joint.head.bias"][dims.blank] += 3.0
    sd["joint.head.bias"][dims.vocab] -= 2.0
    return sd
