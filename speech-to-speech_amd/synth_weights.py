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


@dataclass
class Qwen3TTSDims:
    """Qwen3-TTS-12Hz dims.  The 1.7B numbers are the SURVEY.md 8a assumptions (no checkpoint/config offline)."""
    # talker (dense Qwen3 decoder)
    t_hidden: int = 2048
    t_layers: int = 28
    t_heads: int = 16
    t_kv_heads: int = 8
    t_head_dim: int = 128
    t_ffn: int = 6144
    t_theta: float = 1000000.0
    codec_vocab: int = 3072         # codebook-0 ids + codec special tokens (eos, pad, bos ...)
    codec_eos: int = 2150
    # code predictor
    p_hidden: int = 1024
    p_layers: int = 5
    p_heads: int = 16
    p_kv_heads: int = 8
    p_head_dim: int = 128
    p_ffn: int = 3072
    p_theta: float = 10000.0
    n_groups: int = 16
    codebook: int = 2048
    eps: float = 1e-6
    # codec decoder
    c_hidden: int = 1024
    c_layers: int = 8
    c_heads: int = 16
    c_ffn: int = 3072
    c_window: int = 72
    c_theta: float = 10000.0
    c_eps: float = 1e-5
    c_decoder_dim: int = 1536
    upsample_rates: tuple = (8, 5, 4, 3)
    upsampling_ratios: tuple = (2, 2)

    @property
    def total_upsample(self) -> int:
        n = 1
        for r in tuple(self.upsample_rates) + tuple(self.upsampling_ratios):
            n *= r
        return n

    def talker_cfg(self):
        return dict(hidden=self.t_hidden, layers=self.t_layers, heads=self.t_heads, kv_heads=self.t_kv_heads, head_dim=self.t_head_dim,
                    ffn=self.t_ffn, eps=self.eps, theta=self.t_theta, qk_norm=True)

    def predictor_cfg(self):
        return dict(hidden=self.p_hidden, layers=self.p_layers, heads=self.p_heads, kv_heads=self.p_kv_heads, head_dim=self.p_head_dim,
                    ffn=self.p_ffn, eps=self.eps, theta=self.p_theta, qk_norm=True)

    def codec_cfg(self):
        return dict(hidden=self.c_hidden, layers=self.c_layers, heads=self.c_heads, ffn=self.c_ffn, eps=self.c_eps, theta=self.c_theta,
                    window=self.c_window, num_quantizers=self.n_groups, codebook_size=self.codebook,
                    upsample_rates=list(self.upsample_rates), upsampling_ratios=list(self.upsampling_ratios))

    @staticmethod
    def tts_1_7b() -> "Qwen3TTSDims":
        return Qwen3TTSDims()

    @staticmethod
    def test_tiny() -> "Qwen3TTSDims":
        return Qwen3TTSDims(t_hidden=128, t_layers=2, t_heads=4, t_kv_heads=2, t_head_dim=32, t_ffn=192, codec_vocab=72, codec_eos=70,
                            p_hidden=128, p_layers=2, p_heads=4, p_kv_heads=2, p_head_dim=32, p_ffn=192, n_groups=4, codebook=64,
                            c_hidden=64, c_layers=2, c_heads=2, c_ffn=96, c_window=6, c_decoder_dim=64, upsample_rates=(4, 3),
                            upsampling_ratios=(2, 2))


def qwen3_tts_state_dict(dims: Qwen3TTSDims, seed: int = 0, gain: float = 1.0) -> dict[str, torch.Tensor]:
    """float32 state dict: talker.* (transformers Qwen3 names + codec_embedding / codec_head), code_predictor.* and
    code2wav.* (transformers qwen3_omni_moe names)."""
    sd: dict[str, torch.Tensor] = {}

    def t(name, shape, std):
        sd[name] = _normal(seed, name, shape, std)

    def stack(prefix, hidden, layers, heads, kv, hd, ffn, qk_norm, layer_scale):
        for l in range(layers):
            p = f"{prefix}layers.{l}."
            t(p + "self_attn.q_proj.weight", (heads * hd, hidden), gain / np.sqrt(hidden))
            t(p + "self_attn.k_proj.weight", (kv * hd, hidden), gain / np.sqrt(hidden))
            t(p + "self_attn.v_proj.weight", (kv * hd, hidden), gain / np.sqrt(hidden))
            t(p + "self_attn.o_proj.weight", (hidden, heads * hd), gain / np.sqrt(heads * hd))
            if qk_norm:
                sd[p + "self_attn.q_norm.weight"] = 1.0 + _normal(seed, p + "qn", (hd,), 0.1)
                sd[p + "self_attn.k_norm.weight"] = 1.0 + _normal(seed, p + "kn", (hd,), 0.1)
            t(p + "mlp.gate_proj.weight", (ffn, hidden), gain / np.sqrt(hidden))
            t(p + "mlp.up_proj.weight", (ffn, hidden), gain / np.sqrt(hidden))
            t(p + "mlp.down_proj.weight", (hidden, ffn), gain / np.sqrt(ffn))
            sd[p + "input_layernorm.weight"] = 1.0 + _normal(seed, p + "iln", (hidden,), 0.1)
            sd[p + "post_attention_layernorm.weight"] = 1.0 + _normal(seed, p + "pln", (hidden,), 0.1)
            if layer_scale:
                sd[p + "self_attn_layer_scale.scale"] = 0.3 + _normal(seed, p + "als", (hidden,), 0.05)
                sd[p + "mlp_layer_scale.scale"] = 0.3 + _normal(seed, p + "mls", (hidden,), 0.05)

    E = dims.t_hidden
    stack("talker.model.", E, dims.t_layers, dims.t_heads, dims.t_kv_heads, dims.t_head_dim, dims.t_ffn, True, False)
    sd["talker.model.norm.weight"] = 1.0 + _normal(seed, "talker.norm", (E,), 0.1)
    t("talker.model.codec_embedding.weight", (dims.codec_vocab, E), 1.0)
    t("talker.codec_head.weight", (dims.codec_vocab, E), 2.0 / np.sqrt(E))
    P = dims.p_hidden
    stack("code_predictor.model.", P, dims.p_layers, dims.p_heads, dims.p_kv_heads, dims.p_head_dim, dims.p_ffn, True, False)
    sd["code_predictor.model.norm.weight"] = 1.0 + _normal(seed, "cp.norm", (P,), 0.1)
    if P != E:
        t("code_predictor.in_proj.weight", (P, E), gain / np.sqrt(E))
        t("code_predictor.in_proj.bias", (P,), 0.05)
    for g in range(dims.n_groups - 1):
        t(f"code_predictor.model.codec_embedding.{g}.weight", (dims.codebook, E), 1.0)
        t(f"code_predictor.lm_head.{g}.weight", (dims.codebook, P), 2.0 / np.sqrt(P))
    H = dims.c_hidden
    c = "code2wav."
    t(c + "code_embedding.weight", (dims.codebook * dims.n_groups, H), 1.0)
    stack(c + "pre_transformer.", H, dims.c_layers, dims.c_heads, dims.c_heads, H // dims.c_heads, dims.c_ffn, False, True)
    sd[c + "pre_transformer.norm.weight"] = 1.0 + _normal(seed, "c2w.norm", (H,), 0.1)
    for i, r in enumerate(dims.upsampling_ratios):
        q = f"{c}upsample.{i}."
        t(q + "0.conv.weight", (H, H, r), gain / np.sqrt(H)); t(q + "0.conv.bias", (H,), 0.05)
        t(q + "1.dwconv.conv.weight", (H, 1, 7), gain / np.sqrt(7)); t(q + "1.dwconv.conv.bias", (H,), 0.05)
        sd[q + "1.norm.weight"] = 1.0 + _normal(seed, q + "nw", (H,), 0.1); t(q + "1.norm.bias", (H,), 0.05)
        t(q + "1.pwconv1.weight", (4 * H, H), gain / np.sqrt(H)); t(q + "1.pwconv1.bias", (4 * H,), 0.05)
        t(q + "1.pwconv2.weight", (H, 4 * H), gain / np.sqrt(4 * H)); t(q + "1.pwconv2.bias", (H,), 0.05)
        sd[q + "1.gamma"] = 0.3 + _normal(seed, q + "gamma", (H,), 0.05)
    d = c + "decoder."
    Dd = dims.c_decoder_dim
    t(d + "0.conv.weight", (Dd, H, 7), gain / np.sqrt(7 * H)); t(d + "0.conv.bias", (Dd,), 0.05)
    for i, r in enumerate(dims.upsample_rates):
        cin, cout = Dd // 2 ** i, Dd // 2 ** (i + 1)
        q = f"{d}{i + 1}.block."
        t(q + "0.alpha", (cin,), 0.3); t(q + "0.beta", (cin,), 0.3)
        t(q + "1.conv.weight", (cin, cout, 2 * r), gain / np.sqrt(2 * cin)); t(q + "1.conv.bias", (cout,), 0.05)
        for j in range(3):
            u = f"{q}{j + 2}."
            t(u + "act1.alpha", (cout,), 0.3); t(u + "act1.beta", (cout,), 0.3)
            t(u + "conv1.conv.weight", (cout, cout, 7), gain / np.sqrt(7 * cout)); t(u + "conv1.conv.bias", (cout,), 0.05)
            t(u + "act2.alpha", (cout,), 0.3); t(u + "act2.beta", (cout,), 0.3)
            t(u + "conv2.conv.weight", (cout, cout, 1), gain / np.sqrt(cout)); t(u + "conv2.conv.bias", (cout,), 0.05)
    n = len(dims.upsample_rates) + 1
    co = Dd // 2 ** len(dims.upsample_rates)
    t(f"{d}{n}.alpha", (co,), 0.3); t(f"{d}{n}.beta", (co,), 0.3)
    t(f"{d}{n + 1}.conv.weight", (1, co, 7), 0.02 / np.sqrt(7 * co)); t(f"{d}{n + 1}.conv.bias", (1,), 0.01)
    return sd
