"""Oracle: Whisper encoder + greedy decoder in torch-CPU float32 (test infrastructure, see
oracle/__init__.py).  Written from scratch as plain tensor algebra over a state dict with
the checkpoint's parameter names; pinned to transformers' WhisperForConditionalGeneration
by tests/golden/whisper_tiny.npz (tests/test_oracle_whisper.py).

Restates transformers/models/whisper/modeling_whisper.py:
  encoder  conv1+GELU, conv2(stride 2)+GELU, + embed_positions over all 1500 frames (:618-624),
           pre-LN blocks: self-attention with q scaled by head_dim**-0.5, k_proj without bias
           (:267-330), exact-erf GELU MLP (:371-377), final layer_norm (:642)
  decoder  token + learned position embeddings (:754-758), causal self-attention with KV cache,
           cross-attention over the encoder output, MLP, final layer_norm, tied-embedding logits
and generation_whisper.py: forced prefix (:1455-1608), SuppressTokens / SuppressTokensAtBegin
(:1774-1812), language detection = argmax over language tokens after <|startoftranscript|>
(:1610-1675).  The reference handler calls these with num_beams=1, max_new_tokens=128
(STT/whisper_stt_handler.py:243; arguments_classes/whisper_stt_arguments.py:27-52).
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _heads(x, H):
    B, T, d = x.shape
    return x.view(B, T, H, d // H).transpose(1, 2)


def _attn(q, k, v, H, mask=None):
    B, Tq, d = q.shape
    D = d // H
    qh = _heads(q * D ** -0.5, H)
    kh, vh = _heads(k, H), _heads(v, H)
    s = qh @ kh.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, d)


def encode(sd: dict, feats: torch.Tensor, n_heads: int, n_layers: int) -> torch.Tensor:
    """feats [B, n_mels, 3000] float32 -> encoder output [B, 1500, d]."""
    pre = "model.encoder."
    x = F.gelu(F.conv1d(feats, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[pre + "embed_positions.weight"]
    for l in range(n_layers):
        p = f"{pre}layers.{l}."
        h = _ln(x, sd, p + "self_attn_layer_norm")
        a = _attn(_lin(h, sd, p + "self_attn.q_proj"), _lin(h, sd, p + "self_attn.k_proj"),
                  _lin(h, sd, p + "self_attn.v_proj"), n_heads)
        x = x + _lin(a, sd, p + "self_attn.out_proj")
        h = _ln(x, sd, p + "final_layer_norm")
        x = x + _lin(F.gelu(_lin(h, sd, p + "fc1")), sd, p + "fc2")
    return _ln(x, sd, pre + "layer_norm")


class Decoder:
    """Step-wise decoder with self-attention KV cache (one batch)."""

    def __init__(self, sd: dict, enc_out: torch.Tensor, n_heads: int, n_layers: int):
        self.sd, self.H, self.L = sd, n_heads, n_layers
        pre = "model.decoder."
        self.cross = []
        for l in range(n_layers):
            p = f"{pre}layers.{l}.encoder_attn."
            self.cross.append((_lin(enc_out, sd, p + "k_proj"), _lin(enc_out, sd, p + "v_proj")))
        self.k = [None] * n_layers
        self.v = [None] * n_layers
        self.pos = 0

    def step(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int64 [B] -> logits [B, vocab] for the next position."""
        sd, pre = self.sd, "model.decoder."
        x = sd[pre + "embed_tokens.weight"][ids] + sd[pre + "embed_positions.weight"][self.pos]
        x = x[:, None, :]
        for l in range(self.L):
            p = f"{pre}layers.{l}."
            h = _ln(x, sd, p + "self_attn_layer_norm")
            k, v = _lin(h, sd, p + "self_attn.k_proj"), _lin(h, sd, p + "self_attn.v_proj")
            self.k[l] = k if self.k[l] is None else torch.cat([self.k[l], k], 1)
            self.v[l] = v if self.v[l] is None else torch.cat([self.v[l], v], 1)
            a = _attn(_lin(h, sd, p + "self_attn.q_proj"), self.k[l], self.v[l], self.H)
            x = x + _lin(a, sd, p + "self_attn.out_proj")
            h = _ln(x, sd, p + "encoder_attn_layer_norm")
            a = _attn(_lin(h, sd, p + "encoder_attn.q_proj"), self.cross[l][0], self.cross[l][1], self.H)
            x = x + _lin(a, sd, p + "encoder_attn.out_proj")
            h = _ln(x, sd, p + "final_layer_norm")
            x = x + _lin(F.gelu(_lin(h, sd, p + "fc1")), sd, p + "fc2")
        x = _ln(x, sd, pre + "layer_norm")[:, 0]
        self.pos += 1
        return F.linear(x, sd[pre + "embed_tokens.weight"])


def greedy_decode(sd: dict, enc_out: torch.Tensor, n_heads: int, n_layers: int, prompt: Sequence[Sequence[int]],
                  eos: int, max_new_tokens: int, suppress: Sequence[int] = (), begin_suppress: Sequence[int] = (),
                  lang_ids: Sequence[int] = (), return_logits: bool = False):
    """prompt: per-session forced prefix; -1 entries are filled with the argmax over `lang_ids`.
    Returns (list of generated id lists without eos, filled prompts, [per-step logits])."""
    B = enc_out.shape[0]
    prompt = [list(p) for p in prompt]
    P = len(prompt[0])
    dec = Decoder(sd, enc_out, n_heads, n_layers)
    cur = torch.tensor([p[0] for p in prompt])
    out = [[] for _ in range(B)]
    done = [False] * B
    all_logits = []
    for s in range(P - 1 + max_new_tokens):
        logits = dec.step(cur)
        nxt = []
        if s + 1 < P:
            for b in range(B):
                t = prompt[b][s + 1]
                if t < 0:
                    m = torch.full_like(logits[b], -math.inf)
                    m[list(lang_ids)] = logits[b][list(lang_ids)]
                    t = int(torch.argmax(m))
                    prompt[b][s + 1] = t
                nxt.append(t)
        else:
            lg = logits.clone()
            if len(suppress):
                lg[:, list(suppress)] = -math.inf
            if s == P - 1 and len(begin_suppress):
                lg[:, list(begin_suppress)] = -math.inf
            all_logits.append(lg)
            pick = torch.argmax(lg, dim=-1).tolist()
            for b in range(B):
                t = pick[b]
                if not done[b]:
                    if t == eos:
                        done[b] = True
                    elif len(out[b]) < max_new_tokens:
                        out[b].append(t)
                nxt.append(eos if done[b] else t)
            if all(done):
                break
        cur = torch.tensor(nxt)
    if return_logits:
        return out, prompt, all_logits
    return out, prompt
