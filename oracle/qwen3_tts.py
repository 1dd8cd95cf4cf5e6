"""Oracle: Qwen3-TTS talker / code predictor / codec decoder in torch-CPU float32 (test infrastructure,
see oracle/__init__.py).

The reference runs this arithmetic inside faster-qwen3-tts -> qwentts-cpp-python==0.3.1 (ggml) or qwen_tts
(torch) - reference TTS/qwen3_tts_handler.py:213-249, 910-1003 - none of which is available offline.  The
algorithm is restated from the sibling architecture that IS in the container:
  transformers/models/qwen3_omni_moe/modeling_qwen3_omni_moe.py
    :2250-2330, :2440-2620   code predictor (Qwen3 block with per-head q/k RMSNorm, RoPE, GQA, SwiGLU;
                             one embedding table + one head per residual code group)
    :3100-3177               frame loop: predictor input = [talker last hidden, embed(code0)]; next talker input =
                             sum of all code-group embeddings + trailing text hidden | tts_pad_embed
    :3180-3697               Code2Wav codec decoder (sliding-window transformer, causal transposed convs,
                             ConvNeXt, SnakeBeta residual units, clamp)
  transformers/models/qwen3/modeling_qwen3.py  dense Qwen3 decoder layer (the talker backbone; the sibling's
                             talker uses an MoE MLP, Qwen3-TTS a dense SwiGLU one).
PINNING: every block below is pinned against those sibling modules (tests/golden/qwen3_tts_tiny.npz).  Against the
real Qwen3-TTS checkpoint (prompt layout, speaker table, sampler defaults) parity is UNPINNED - see DESIGN.md.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def rms(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_tables(positions: torch.Tensor, head_dim: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    return emb.cos(), emb.sin()


def _rot(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def apply_rope(x, cos, sin):   # x [B,H,T,D], cos/sin [T,D]
    return x * cos[None, None] + _rot(x) * sin[None, None]


class DecoderStack:
    """Qwen3-style pre-norm decoder stack with KV cache.  cfg keys: hidden, layers, heads, kv_heads, head_dim,
    ffn, eps, theta, qk_norm (bool), layer_scale (bool), window (0 = full causal)."""

    def __init__(self, sd, prefix, cfg):
        self.sd, self.p, self.c = sd, prefix, cfg
        self.k = [None] * cfg["layers"]
        self.v = [None] * cfg["layers"]
        self.pos = 0

    def forward(self, x):   # x [B, T, hidden] -> [B, T, hidden] (before the final norm)
        sd, c = self.sd, self.c
        B, T, _ = x.shape
        H, Hk, D = c["heads"], c["kv_heads"], c["head_dim"]
        cos, sin = rope_tables(torch.arange(self.pos, self.pos + T), D, c["theta"])
        for l in range(c["layers"]):
            p = f"{self.p}layers.{l}."
            h = rms(x, sd[p + "input_layernorm.weight"], c["eps"])
            q = F.linear(h, sd[p + "self_attn.q_proj.weight"]).view(B, T, H, D)
            k = F.linear(h, sd[p + "self_attn.k_proj.weight"]).view(B, T, Hk, D)
            v = F.linear(h, sd[p + "self_attn.v_proj.weight"]).view(B, T, Hk, D)
            if c.get("qk_norm", True):
                q = rms(q, sd[p + "self_attn.q_norm.weight"], c["eps"])
                k = rms(k, sd[p + "self_attn.k_norm.weight"], c["eps"])
            q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            self.k[l] = k if self.k[l] is None else torch.cat([self.k[l], k], 2)
            self.v[l] = v if self.v[l] is None else torch.cat([self.v[l], v], 2)
            kk = self.k[l].repeat_interleave(H // Hk, 1)
            vv = self.v[l].repeat_interleave(H // Hk, 1)
            Tk = kk.shape[2]
            s = (q @ kk.transpose(-1, -2)) * D ** -0.5
            qi = torch.arange(self.pos, self.pos + T)[:, None]
            kj = torch.arange(Tk)[None, :]
            allowed = kj <= qi
            if c.get("window", 0) > 0:
                allowed &= kj > qi - c["window"]
            s = s.masked_fill(~allowed, -math.inf)
            a = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, T, H * D)
            a = F.linear(a, sd[p + "self_attn.o_proj.weight"])
            if c.get("layer_scale", False):
                a = a * sd[p + "self_attn_layer_scale.scale"]
            x = x + a
            h = rms(x, sd[p + "post_attention_layernorm.weight"], c["eps"])
            m = F.linear(F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"]),
                         sd[p + "mlp.down_proj.weight"])
            if c.get("layer_scale", False):
                m = m * sd[p + "mlp_layer_scale.scale"]
            x = x + m
        self.pos += T
        return x


def code_predictor_greedy(sd, cfg, past_hidden, code0_embed, n_groups, forced=None, logits_out=None):
    """past_hidden / code0_embed [B, E] -> residual codes int64 [B, n_groups-1] (greedy) and the embeddings
    (predictor tables) of each generated code [B, n_groups-1, E].  forced [B, n_groups-1] (optional): emit these codes
    instead of the argmax (teacher forcing); logits_out (optional list): receives every group's logits [B, codebook]."""
    p = "code_predictor."
    x = torch.stack([past_hidden, code0_embed], 1)
    if p + "in_proj.weight" in sd:
        x = F.linear(x, sd[p + "in_proj.weight"], sd.get(p + "in_proj.bias"))
    stack = DecoderStack(sd, p + "model.", cfg)
    codes, embs = [], []
    for g in range(n_groups - 1):
        h = stack.forward(x)
        h = rms(h[:, -1], sd[p + "model.norm.weight"], cfg["eps"])
        logits = F.linear(h, sd[p + f"lm_head.{g}.weight"])
        if logits_out is not None:
            logits_out.append(logits.clone())
        c = logits.argmax(-1)
        if forced is not None:
            c = torch.where(forced[:, g] >= 0, forced[:, g], c)
        codes.append(c)
        e = sd[p + f"model.codec_embedding.{g}.weight"][c]
        embs.append(e)
        x = e[:, None]
        if p + "in_proj.weight" in sd:
            x = F.linear(x, sd[p + "in_proj.weight"], sd.get(p + "in_proj.bias"))
    return torch.stack(codes, 1), torch.stack(embs, 1)


def talker_generate(sd, tcfg, pcfg, prompt_embeds, n_groups, max_frames, eos_id, suppress=(), trailing=None, pad_embed=None,
                    min_frames=0, forced=None, logits_out=None):
    """Greedy frame loop.  prompt_embeds [B, P, E].  Returns codes int64 [B, frames, n_groups] (a session's rows after
    its eos frame are filled with -1), n_frames per session.  forced int64 [B, frames, n_groups] (optional, entries < 0 keep
    the greedy pick): teacher forcing; logits_out (optional list): per frame a list of n_groups raw logits tensors
    (group 0: [B, codec_vocab] BEFORE the suppress / eos masks, groups 1..: [B, codebook])."""
    B = prompt_embeds.shape[0]
    stack = DecoderStack(sd, "talker.model.", tcfg)
    norm_w, head = sd["talker.model.norm.weight"], sd["talker.codec_head.weight"]
    emb0 = sd["talker.model.codec_embedding.weight"]
    h = stack.forward(prompt_embeds)[:, -1]
    out = []
    done = torch.zeros(B, dtype=torch.bool)
    nfr = torch.zeros(B, dtype=torch.long)
    for f in range(max_frames):
        hn = rms(h, norm_w, tcfg["eps"])
        logits = F.linear(hn, head)
        frame_logits = [logits.clone()] if logits_out is not None else None
        if len(suppress):
            logits[:, list(suppress)] = -math.inf
        if f < min_frames:
            logits[:, eos_id] = -math.inf
        c0 = logits.argmax(-1)
        if forced is not None:
            c0 = torch.where(forced[:, f, 0] >= 0, forced[:, f, 0], c0)
        done = done | (c0 == eos_id)
        if bool(done.all()):
            break
        e0 = emb0[c0]
        rc, re = code_predictor_greedy(sd, pcfg, hn, e0, n_groups, None if forced is None else forced[:, f, 1:], frame_logits)
        if logits_out is not None:
            logits_out.append(frame_logits)
        codes = torch.cat([c0[:, None], rc], 1)
        codes[done] = -1
        out.append(codes)
        nfr += (~done).long()
        x = e0 + re.sum(1)
        if trailing is not None and f < trailing.shape[1]:
            x = x + trailing[:, f]
        elif pad_embed is not None:
            x = x + pad_embed
        h = stack.forward(x[:, None])[:, -1]
    codes = torch.stack(out, 1) if out else torch.zeros(B, 0, n_groups, dtype=torch.long)
    return codes, nfr


# ---- codec decoder (Code2Wav) --------------------------------------------------------------------------
def _causal_conv(x, w, b, dilation=1, stride=1):   # x [B,C,T]
    k_eff = (w.shape[-1] - 1) * dilation + 1
    pad = k_eff - stride
    length = x.shape[-1]
    n_frames = (length - k_eff + pad) / stride + 1
    extra = (math.ceil(n_frames) - 1) * stride + (k_eff - pad) - length
    return F.conv1d(F.pad(x, (pad, extra)), w, b, stride=stride, dilation=dilation, groups=1 if w.shape[1] == x.shape[1] else x.shape[1])


def _causal_tconv(x, w, b, stride):
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pad = k - stride
    return y[..., pad: y.shape[-1] - pad] if pad > 0 else y


def _snake(x, alpha, beta):
    a, b = torch.exp(alpha)[None, :, None], torch.exp(beta)[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def codec_decode(sd, cfg, codes):
    """codes int64 [B, Q, F] -> waveform [B, 1, samples] in [-1, 1]."""
    p = "code2wav."
    Q, cb = cfg["num_quantizers"], cfg["codebook_size"]
    off = (torch.arange(Q) * cb).view(1, -1, 1)
    h = sd[p + "code_embedding.weight"][codes + off].mean(1)          # [B, F, hidden]
    tc = dict(hidden=cfg["hidden"], layers=cfg["layers"], heads=cfg["heads"], kv_heads=cfg["heads"], head_dim=cfg["hidden"] // cfg["heads"],
              ffn=cfg["ffn"], eps=cfg["eps"], theta=cfg["theta"], qk_norm=False, layer_scale=True, window=cfg["window"])
    h = DecoderStack(sd, p + "pre_transformer.", tc).forward(h)
    h = rms(h, sd[p + "pre_transformer.norm.weight"], cfg["eps"]).permute(0, 2, 1)
    for i, r in enumerate(cfg["upsampling_ratios"]):
        q = f"{p}upsample.{i}."
        h = _causal_tconv(h, sd[q + "0.conv.weight"], sd[q + "0.conv.bias"], r)
        res = h
        y = _causal_conv(h, sd[q + "1.dwconv.conv.weight"], sd[q + "1.dwconv.conv.bias"]).permute(0, 2, 1)
        y = F.layer_norm(y, (y.shape[-1],), sd[q + "1.norm.weight"], sd[q + "1.norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[q + "1.pwconv1.weight"], sd[q + "1.pwconv1.bias"])), sd[q + "1.pwconv2.weight"], sd[q + "1.pwconv2.bias"])
        h = res + (sd[q + "1.gamma"] * y).permute(0, 2, 1)
    d = p + "decoder."
    w = _causal_conv(h, sd[d + "0.conv.weight"], sd[d + "0.conv.bias"])
    for i, r in enumerate(cfg["upsample_rates"]):
        q = f"{d}{i + 1}.block."
        w = _snake(w, sd[q + "0.alpha"], sd[q + "0.beta"])
        w = _causal_tconv(w, sd[q + "1.conv.weight"], sd[q + "1.conv.bias"], r)
        for j, dil in enumerate((1, 3, 9)):
            u = f"{q}{j + 2}."
            res = w
            y = _snake(w, sd[u + "act1.alpha"], sd[u + "act1.beta"])
            y = _causal_conv(y, sd[u + "conv1.conv.weight"], sd[u + "conv1.conv.bias"], dilation=dil)
            y = _snake(y, sd[u + "act2.alpha"], sd[u + "act2.beta"])
            y = _causal_conv(y, sd[u + "conv2.conv.weight"], sd[u + "conv2.conv.bias"])
            w = y + res
    n = len(cfg["upsample_rates"]) + 1
    w = _snake(w, sd[f"{d}{n}.alpha"], sd[f"{d}{n}.beta"])
    w = _causal_conv(w, sd[f"{d}{n + 1}.conv.weight"], sd[f"{d}{n + 1}.conv.bias"])
    return w.clamp(-1, 1)


# ---- host post-processing of the reference handler (TTS/qwen3_tts_handler.py:612-613, 674-749) ----------
def resample_24k_to_16k(x):
    from scipy.signal import resample_poly

    return resample_poly(x, up=2, down=3)


def to_int16(x):
    import numpy as np

    return np.clip(x * 32768, -32768, 32767).astype(np.int16)
