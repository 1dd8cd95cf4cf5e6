"""Oracle: Parakeet-TDT (FastConformer encoder + TDT greedy decode) in torch-CPU float32
(test infrastructure, see oracle/__init__.py).  From-scratch tensor algebra over a state dict
with the checkpoint's parameter names; pinned to transformers' ParakeetForTDT by
tests/golden/parakeet_tiny.npz.

The reference obtains this arithmetic from nano_parakeet's ``model.transcribe`` (absent offline;
STT/parakeet_tdt_handler.py:621-623).  Restated from transformers/models/parakeet/:
  modeling_parakeet.py:365-431  dw-striding conv2d x8 subsampling (+ per-conv time masking)
  :68-106    sinusoidal relative position table, positions T-1 .. -(T-1)
  :267-362   rel-pos attention: (q+u).k^T*s + relshift((q+v).R^T)*s, -inf key mask
  :124-193   conv module: pointwise->GLU->mask->depthwise k9->BatchNorm(eval)->SiLU->pointwise
  :434-479   block: x+=.5*FF1(LN) ; x+=MHSA(LN) ; x+=Conv(LN) ; x+=.5*FF2(LN) ; LN_out
  :831-895   LSTM prediction network, state advanced only on non-blank emissions
  :1035-1052 joint = head(relu(enc_proj + dec_proj)), vocab + 5 duration logits
  generation_parakeet.py:271-294  TDT greedy: token/duration argmax, blank&dur0 -> 1, frame += dur,
  :113-122   stop when the frame pointer reaches the valid encoder length.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def rel_pos_table(T: int, d: int) -> torch.Tensor:
    """[2T-1, d] interleaved sin/cos of positions T-1 .. -(T-1) (modeling_parakeet.py:68-106)."""
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)
    fr = pos[:, None] * inv[None, :]
    return torch.stack([fr.sin(), fr.cos()], dim=-1).reshape(2 * T - 1, d)


def subsample(sd, feats: torch.Tensor, lengths: torch.Tensor):
    """feats [B, T, n_mels], lengths [B] valid frames -> ([B, T', d], lengths')."""
    p = "encoder.subsampling."
    x = feats[:, None]
    cur = lengths.clone()

    def mask(x, cur):
        m = torch.arange(x.shape[2])[None, :] < cur[:, None]
        return x * m[:, None, :, None]

    def out_len(n):
        return (n + 2 - 3) // 2 + 1

    x = F.conv2d(x, sd[p + "layers.0.weight"], sd[p + "layers.0.bias"], stride=2, padding=1)
    cur = out_len(cur)
    x = F.relu(mask(x, cur))
    for i in (2, 5):
        C = x.shape[1]
        x = F.conv2d(x, sd[p + f"layers.{i}.weight"], sd[p + f"layers.{i}.bias"], stride=2, padding=1, groups=C)
        cur = out_len(cur)
        x = mask(x, cur)
        x = F.conv2d(x, sd[p + f"layers.{i + 1}.weight"], sd[p + f"layers.{i + 1}.bias"])
        x = F.relu(mask(x, cur))
    B, C, T, Fq = x.shape
    x = x.transpose(1, 2).reshape(B, T, C * Fq)
    return _lin(x, sd, p + "linear"), cur


def encode(sd, feats: torch.Tensor, lengths: torch.Tensor, n_heads: int, n_layers: int, scale_input: bool = True):
    """-> (encoder output [B, T', d], valid lengths [B])."""
    x, lens = subsample(sd, feats, lengths)
    B, T, d = x.shape
    if scale_input:
        x = x * math.sqrt(d)
    pos = rel_pos_table(T, d)
    D = d // n_heads
    s = D ** -0.5
    valid = torch.arange(T)[None, :] < lens[:, None]                  # [B, T]
    amask = (valid[:, :, None] & valid[:, None, :])[:, None]          # [B,1,T,T]
    for l in range(n_layers):
        p = f"encoder.layers.{l}."
        h = _ln(x, sd, p + "norm_feed_forward1")
        x = x + 0.5 * _lin(F.silu(_lin(h, sd, p + "feed_forward1.linear1")), sd, p + "feed_forward1.linear2")
        h = _ln(x, sd, p + "norm_self_att")
        a = p + "self_attn."
        q = _lin(h, sd, a + "q_proj").view(B, T, n_heads, D).transpose(1, 2)
        k = _lin(h, sd, a + "k_proj").view(B, T, n_heads, D).transpose(1, 2)
        v = _lin(h, sd, a + "v_proj").view(B, T, n_heads, D).transpose(1, 2)
        qu = q + sd[a + "bias_u"][None, :, None, :]
        qv = q + sd[a + "bias_v"][None, :, None, :]
        r = F.linear(pos, sd[a + "relative_k_proj.weight"]).view(2 * T - 1, n_heads, D)
        bd = qv @ r.permute(1, 2, 0)[None]                                # [B,H,T,2T-1]
        bd = F.pad(bd, (1, 0)).view(B, n_heads, 2 * T, T)[:, :, 1:].reshape(B, n_heads, T, 2 * T - 1)[..., :T]
        sc = (qu @ k.transpose(-1, -2)) * s + bd * s
        sc = sc.masked_fill(~amask, -math.inf)
        pr = torch.softmax(sc, -1)
        pr = torch.nan_to_num(pr, nan=0.0)          # fully masked (padded) query rows
        o = (pr @ v).transpose(1, 2).reshape(B, T, d)
        x = x + _lin(o, sd, a + "o_proj")
        h = _ln(x, sd, p + "norm_conv")
        c = p + "conv."
        y = F.conv1d(h.transpose(1, 2), sd[c + "pointwise_conv1.weight"], sd[c + "pointwise_conv1.bias"])
        y = F.glu(y, dim=1)
        y = y.masked_fill(~valid[:, None, :], 0.0)
        y = F.conv1d(y, sd[c + "depthwise_conv.weight"], sd[c + "depthwise_conv.bias"], padding=4, groups=d)
        y = F.batch_norm(y, sd[c + "norm.running_mean"], sd[c + "norm.running_var"], sd[c + "norm.weight"], sd[c + "norm.bias"], False, 0.0, 1e-5)
        y = F.conv1d(F.silu(y), sd[c + "pointwise_conv2.weight"], sd[c + "pointwise_conv2.bias"])
        x = x + y.transpose(1, 2)
        h = _ln(x, sd, p + "norm_feed_forward2")
        x = x + 0.5 * _lin(F.silu(_lin(h, sd, p + "feed_forward2.linear1")), sd, p + "feed_forward2.linear2")
        x = _ln(x, sd, p + "norm_out")
    return x, lens


def _lstm_step(sd, x, h, c, layers):
    hs, cs = [], []
    for l in range(layers):
        g = F.linear(x, sd[f"decoder.lstm.weight_ih_l{l}"], sd[f"decoder.lstm.bias_ih_l{l}"]) + \
            F.linear(h[l], sd[f"decoder.lstm.weight_hh_l{l}"], sd[f"decoder.lstm.bias_hh_l{l}"])
        i, f, gg, o = g.chunk(4, -1)
        cn = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
        hn = torch.sigmoid(o) * torch.tanh(cn)
        hs.append(hn)
        cs.append(cn)
        x = hn
    return x, hs, cs


def tdt_greedy(sd, enc: torch.Tensor, lens: torch.Tensor, vocab: int, blank: int, n_lstm: int = 2, max_steps: int | None = None,
               return_logits: bool = False):
    """-> per session (tokens incl. blanks, durations, frame index at which each step was taken)."""
    B, T, _ = enc.shape
    encp = _lin(enc, sd, "encoder_projector")
    Hd = encp.shape[-1]
    h = [torch.zeros(B, Hd) for _ in range(n_lstm)]
    c = [torch.zeros(B, Hd) for _ in range(n_lstm)]
    out, h, c = _lstm_step(sd, sd["decoder.embedding.weight"][torch.full((B,), blank)], h, c, n_lstm)
    dec = _lin(out, sd, "decoder.decoder_projector")
    frame = torch.zeros(B, dtype=torch.long)
    finished = torch.zeros(B, dtype=torch.bool)
    toks = [[] for _ in range(B)]
    durs = [[] for _ in range(B)]
    frames = [[] for _ in range(B)]
    logits_all = []
    max_steps = max_steps or 10 * T - 1   # generate(): max_length = max_symbols_per_step * T' incl. the start token
    for _ in range(max_steps):
        f_idx = frame.clamp(max=T - 1)
        joint = F.relu(encp[torch.arange(B), f_idx] + dec)
        logits = _lin(joint, sd, "joint.head")
        logits_all.append(logits)
        tok = logits[:, :vocab].argmax(-1)
        dur = logits[:, vocab:].argmax(-1)
        is_blank = tok == blank
        dur = torch.where(is_blank & (dur == 0), torch.ones_like(dur), dur)
        for b in range(B):
            if not finished[b]:
                toks[b].append(int(tok[b]))
                durs[b].append(int(dur[b]))
                frames[b].append(int(frame[b]))
        frame = frame + dur
        finished = finished | (frame >= lens)
        if bool(finished.all()):
            break
        # prediction network advances only on non-blank emissions (state of blank rows is kept)
        out, hn, cn = _lstm_step(sd, sd["decoder.embedding.weight"][tok], h, c, n_lstm)
        decn = _lin(out, sd, "decoder.decoder_projector")
        keep = is_blank[:, None]
        h = [torch.where(keep, a, b_) for a, b_ in zip(h, hn)]
        c = [torch.where(keep, a, b_) for a, b_ in zip(c, cn)]
        dec = torch.where(keep, dec, decn)
    if return_logits:
        return toks, durs, frames, logits_all
    return toks, durs, frames
