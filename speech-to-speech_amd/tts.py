"""Host side of the Qwen3-TTS engine (device graph: csrc/tts.cpp, kernels csrc/tts.hip).

WEIGHT LAYOUT handed to ``s2s_tts_set_tensor`` (``W`` = model dtype; norms / biases / tables float32):

  decoder stacks, prefix in {talker, pred, codec}, layer L:
    <p>.L.ln1 [hid]   <p>.L.qkv.w [(H+2Hk)*D][hid] W = q|k|v   <p>.L.qnorm / .knorm [D] (talker, pred)
    <p>.L.o.w [hid][H*D] W   <p>.L.ln2 [hid]   <p>.L.gu.w [2*ffn][hid] W = gate|up   <p>.L.down.w [hid][ffn] W
    (codec: the LayerScale vectors are folded into the rows of o.w and down.w)
    <p>.rope.cos / .rope.sin  [max_pos][D/2]
  talker.norm  talker.emb [codec_vocab][E] W  talker.head.w [codec_vocab][E] W  talker.pad_embed [E] W
  talker.eos_mask uint8 [codec_vocab]
  pred.norm  pred.in_proj.{w [P][E] W, b} (only when P != E)  pred.emb.g [codebook][E] W  pred.head.g.w [codebook][P] W
  codec.emb [codebook*G][H] W   codec.norm [H]
  codec.up.i: tconv.w [r*H][H] W (row j*H+o = W[:, o, j])  tconv.b [r*H]  dw.w [H][7]  dw.b  ln.{g,b}
              pw1.{w [4H][H] W, b}  pw2.{w [H][4H] W, b} with gamma folded in
  codec.dec.in.{w [Dd][7*H] W tap-major, b}
  codec.dec.i: snake.{a = exp(alpha), ib = 1/(exp(beta)+1e-9)}  tconv.w [r*Co][2*C] W (row j*Co+o = [W[:,o,j+r] | W[:,o,j]])
               tconv.b [r*Co]   ruJ.{s1.a, s1.ib, c1.w [Co][7*Co] W tap-major, c1.b, s2.a, s2.ib, c2.w [Co][Co] W, c2.b}
  codec.dec.out.{a, ib, w [1][7*co] W, b [1]}

Prompt layout: the real Qwen3-TTS chat template / speaker table / language tags live in the unavailable qwen_tts
package (SURVEY.md 8a'); prompts are therefore assembled from explicit (text-side, codec-side) embedding rows behind
``TTSPrompt`` and exercised with synthetic prompts.  ``non_streaming_mode=True`` semantics: the whole text is prefilled,
every generated frame adds ``tts_pad_embed``; ``False`` (streaming text): ``TTSPrompt.trailing`` rows replace the pad row frame by frame.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterator, Mapping, Sequence

import numpy as np
import torch

from . import native
from .synth_weights import Qwen3TTSDims

PIPELINE_SR = 16000
CODEC_SR = 24000
BLOCKSIZE = 512


def rope_cos_sin(max_pos: int, head_dim: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()


def _stack(out, sd, src, dst, layers, dtype, layer_scale=False, qk_norm=True):
    W = lambda t: t.to(dtype).contiguous()            # noqa: E731
    F32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
    for l in range(layers):
        p, q = f"{src}layers.{l}.", f"{dst}{l}."
        out[q + "ln1"] = F32(sd[p + "input_layernorm.weight"])
        out[q + "qkv.w"] = W(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0))
        if qk_norm:
            out[q + "qnorm"] = F32(sd[p + "self_attn.q_norm.weight"])
            out[q + "knorm"] = F32(sd[p + "self_attn.k_norm.weight"])
        ow, dw = sd[p + "self_attn.o_proj.weight"], sd[p + "mlp.down_proj.weight"]
        if layer_scale:
            ow = ow * sd[p + "self_attn_layer_scale.scale"][:, None]
            dw = dw * sd[p + "mlp_layer_scale.scale"][:, None]
        out[q + "o.w"] = W(ow)
        out[q + "ln2"] = F32(sd[p + "post_attention_layernorm.weight"])
        out[q + "gu.w"] = W(torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0))
        out[q + "down.w"] = W(dw)


def convert_tts_state_dict(sd: Mapping[str, torch.Tensor], dims: Qwen3TTSDims, dtype: torch.dtype, max_pos: int,
                           max_codec_frames: int, pad_embed: torch.Tensor) -> dict[str, torch.Tensor]:
    out: dict[str, torch.Tensor] = {}
    W = lambda t: t.to(dtype).contiguous()            # noqa: E731
    F32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
    _stack(out, sd, "talker.model.", "talker.", dims.t_layers, dtype)
    _stack(out, sd, "code_predictor.model.", "pred.", dims.p_layers, dtype)
    _stack(out, sd, "code2wav.pre_transformer.", "codec.", dims.c_layers, dtype, layer_scale=True, qk_norm=False)
    for pre, n, hd, th in (("talker.", max_pos, dims.t_head_dim, dims.t_theta), ("pred.", dims.n_groups + 1, dims.p_head_dim, dims.p_theta),
                           ("codec.", max(max_codec_frames, max_pos), dims.c_hidden // dims.c_heads, dims.c_theta)):
        out[pre + "rope.cos"], out[pre + "rope.sin"] = rope_cos_sin(n, hd, th)
    out["talker.norm"] = F32(sd["talker.model.norm.weight"])
    out["talker.emb"] = W(sd["talker.model.codec_embedding.weight"])
    out["talker.head.w"] = W(sd["talker.codec_head.weight"])
    out["talker.pad_embed"] = W(pad_embed)
    eos = torch.zeros(dims.codec_vocab, dtype=torch.uint8)
    eos[dims.codec_eos] = 1
    out["talker.eos_mask"] = eos
    out["pred.norm"] = F32(sd["code_predictor.model.norm.weight"])
    if dims.p_hidden != dims.t_hidden:
        out["pred.in_proj.w"] = W(sd["code_predictor.in_proj.weight"])
        out["pred.in_proj.b"] = F32(sd["code_predictor.in_proj.bias"])
    for g in range(dims.n_groups - 1):
        out[f"pred.emb.{g}"] = W(sd[f"code_predictor.model.codec_embedding.{g}.weight"])
        out[f"pred.head.{g}.w"] = W(sd[f"code_predictor.lm_head.{g}.weight"])
    c = "code2wav."
    H = dims.c_hidden
    out["codec.emb"] = W(sd[c + "code_embedding.weight"])
    out["codec.norm"] = F32(sd[c + "pre_transformer.norm.weight"])
    for i, r in enumerate(dims.upsampling_ratios):
        s, q = f"{c}upsample.{i}.", f"codec.up.{i}."
        tw = sd[s + "0.conv.weight"]                              # [Cin][Cout][r]
        out[q + "tconv.w"] = W(tw.permute(2, 1, 0).reshape(r * H, H))
        out[q + "tconv.b"] = F32(sd[s + "0.conv.bias"].repeat(r))
        out[q + "dw.w"] = F32(sd[s + "1.dwconv.conv.weight"].reshape(H, 7)); out[q + "dw.b"] = F32(sd[s + "1.dwconv.conv.bias"])
        out[q + "ln.g"] = F32(sd[s + "1.norm.weight"]); out[q + "ln.b"] = F32(sd[s + "1.norm.bias"])
        out[q + "pw1.w"] = W(sd[s + "1.pwconv1.weight"]); out[q + "pw1.b"] = F32(sd[s + "1.pwconv1.bias"])
        gam = sd[s + "1.gamma"]
        out[q + "pw2.w"] = W(sd[s + "1.pwconv2.weight"] * gam[:, None]); out[q + "pw2.b"] = F32(sd[s + "1.pwconv2.bias"] * gam)
    d = c + "decoder."

    def conv_w(w):   # [Cout][Cin][k] -> [Cout][k*Cin] tap-major
        return W(w.permute(0, 2, 1).reshape(w.shape[0], -1))

    def snake(dst, a, b):
        out[dst + ".a"] = F32(torch.exp(sd[a])); out[dst + ".ib"] = F32(1.0 / (torch.exp(sd[b]) + 1e-9))

    out["codec.dec.in.w"] = conv_w(sd[d + "0.conv.weight"]); out["codec.dec.in.b"] = F32(sd[d + "0.conv.bias"])
    for i, r in enumerate(dims.upsample_rates):
        s, q = f"{d}{i + 1}.block.", f"codec.dec.{i}."
        snake(q + "snake", s + "0.alpha", s + "0.beta")
        tw = sd[s + "1.conv.weight"]                              # [Cin][Cout][2r]
        cin, cout = tw.shape[0], tw.shape[1]
        early, late = tw[:, :, r:], tw[:, :, :r]                  # tap on x[t-1] / on x[t]
        wcat = torch.cat([early.permute(2, 1, 0), late.permute(2, 1, 0)], dim=2)   # [r][Cout][2*Cin]
        out[q + "tconv.w"] = W(wcat.reshape(r * cout, 2 * cin))
        out[q + "tconv.b"] = F32(sd[s + "1.conv.bias"].repeat(r))
        for j in range(3):
            u, v = f"{s}{j + 2}.", f"{q}ru{j}."
            snake(v + "s1", u + "act1.alpha", u + "act1.beta")
            out[v + "c1.w"] = conv_w(sd[u + "conv1.conv.weight"]); out[v + "c1.b"] = F32(sd[u + "conv1.conv.bias"])
            snake(v + "s2", u + "act2.alpha", u + "act2.beta")
            out[v + "c2.w"] = W(sd[u + "conv2.conv.weight"].reshape(cout, cout)); out[v + "c2.b"] = F32(sd[u + "conv2.conv.bias"])
    n = len(dims.upsample_rates) + 1
    snake("codec.dec.out", f"{d}{n}.alpha", f"{d}{n}.beta")
    out["codec.dec.out.w"] = conv_w(sd[f"{d}{n + 1}.conv.weight"]); out["codec.dec.out.b"] = F32(sd[f"{d}{n + 1}.conv.bias"])
    return out


def random_tts_engine_tensors(dims: Qwen3TTSDims, dtype: torch.dtype, device: torch.device, max_pos: int, max_codec_frames: int,
                              seed: int = 0) -> dict[str, torch.Tensor]:
    """Random-init weights generated directly on the GPU in the engine layout (bench: no checkpoints offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out: dict[str, torch.Tensor] = {}

    def w(name, rows, cols, std=None):
        out[name] = (torch.randn((rows, cols), generator=g, device=device, dtype=torch.float32) * (std or cols ** -0.5)).to(dtype)

    def v(name, n, mean=0.0, std=0.02):
        out[name] = mean + std * torch.randn((n,), generator=g, device=device, dtype=torch.float32)

    def stack(pre, hid, layers, H, Hk, D, ffn, qk_norm, out_scale=1.0):
        for l in range(layers):
            q = f"{pre}{l}."
            v(q + "ln1", hid, 1.0, 0.05); w(q + "qkv.w", (H + 2 * Hk) * D, hid)
            if qk_norm:
                v(q + "qnorm", D, 1.0, 0.05); v(q + "knorm", D, 1.0, 0.05)
            w(q + "o.w", hid, H * D, out_scale * (H * D) ** -0.5); v(q + "ln2", hid, 1.0, 0.05)
            w(q + "gu.w", 2 * ffn, hid); w(q + "down.w", hid, ffn, out_scale * ffn ** -0.5)

    E, P, H, G, CB = dims.t_hidden, dims.p_hidden, dims.c_hidden, dims.n_groups, dims.codebook
    stack("talker.", E, dims.t_layers, dims.t_heads, dims.t_kv_heads, dims.t_head_dim, dims.t_ffn, True)
    stack("pred.", P, dims.p_layers, dims.p_heads, dims.p_kv_heads, dims.p_head_dim, dims.p_ffn, True)
    stack("codec.", H, dims.c_layers, dims.c_heads, dims.c_heads, H // dims.c_heads, dims.c_ffn, False, out_scale=0.3)
    for pre, n, hd, th in (("talker.", max_pos, dims.t_head_dim, dims.t_theta), ("pred.", G + 1, dims.p_head_dim, dims.p_theta),
                           ("codec.", max(max_codec_frames, max_pos), H // dims.c_heads, dims.c_theta)):
        c, s_ = rope_cos_sin(n, hd, th)
        out[pre + "rope.cos"], out[pre + "rope.sin"] = c.to(device), s_.to(device)
    v("talker.norm", E, 1.0, 0.05); w("talker.emb", dims.codec_vocab, E, 1.0); w("talker.head.w", dims.codec_vocab, E)
    out["talker.pad_embed"] = (0.5 * torch.randn((E,), generator=g, device=device)).to(dtype)
    eos = torch.zeros(dims.codec_vocab, dtype=torch.uint8, device=device); eos[dims.codec_eos] = 1
    out["talker.eos_mask"] = eos
    v("pred.norm", P, 1.0, 0.05)
    if P != E:
        w("pred.in_proj.w", P, E); v("pred.in_proj.b", P)
    for i in range(G - 1):
        w(f"pred.emb.{i}", CB, E, 1.0); w(f"pred.head.{i}.w", CB, P)
    w("codec.emb", CB * G, H, 1.0); v("codec.norm", H, 1.0, 0.05)
    for i, r in enumerate(dims.upsampling_ratios):
        q = f"codec.up.{i}."
        w(q + "tconv.w", r * H, H); v(q + "tconv.b", r * H)
        out[q + "dw.w"] = torch.randn((H, 7), generator=g, device=device) * 7 ** -0.5; v(q + "dw.b", H)
        v(q + "ln.g", H, 1.0, 0.05); v(q + "ln.b", H)
        w(q + "pw1.w", 4 * H, H); v(q + "pw1.b", 4 * H); w(q + "pw2.w", H, 4 * H, 0.3 * (4 * H) ** -0.5); v(q + "pw2.b", H)
    Dd = dims.c_decoder_dim
    w("codec.dec.in.w", Dd, 7 * H); v("codec.dec.in.b", Dd)
    for i, r in enumerate(dims.upsample_rates):
        cin, cout = Dd >> i, Dd >> (i + 1)
        q = f"codec.dec.{i}."
        v(q + "snake.a", cin, 1.0, 0.2); v(q + "snake.ib", cin, 1.0, 0.2)
        w(q + "tconv.w", r * cout, 2 * cin); v(q + "tconv.b", r * cout)
        for j in range(3):
            u = f"{q}ru{j}."
            v(u + "s1.a", cout, 1.0, 0.2); v(u + "s1.ib", cout, 1.0, 0.2); w(u + "c1.w", cout, 7 * cout, 0.5 * (7 * cout) ** -0.5); v(u + "c1.b", cout)
            v(u + "s2.a", cout, 1.0, 0.2); v(u + "s2.ib", cout, 1.0, 0.2); w(u + "c2.w", cout, cout, 0.5 * cout ** -0.5); v(u + "c2.b", cout)
    co = Dd >> len(dims.upsample_rates)
    v("codec.dec.out.a", co, 1.0, 0.2); v("codec.dec.out.ib", co, 1.0, 0.2); w("codec.dec.out.w", 1, 7 * co, 0.05 * (7 * co) ** -0.5)
    out["codec.dec.out.b"] = torch.zeros(1, device=device)
    return out


def add_fragment_major_weights(tensors: dict, dims: Qwen3TTSDims) -> None:
    """Decode-path copies of the talker / code-predictor projections in the fragment-major layout of csrc/gemm_dec.hip
    ("<name>.f"), with the RMSNorm weight that precedes qkv / gate-up / the predictor heads folded into their columns.
    Made once per GPU: engine replicas share them through ``TTSEngine._weights``.  No-op when they are already there or
    when the widths do not fit the kernel (multiples of 32, >= 256)."""
    if "talker.0.qkv.f" in tensors:
        return
    E, P = dims.t_hidden, dims.p_hidden
    widths = (E, P, dims.t_ffn, dims.p_ffn, dims.t_heads * dims.t_head_dim, dims.p_heads * dims.p_head_dim)
    if any(w % 32 or w < 256 for w in widths):
        return
    stacks = [("talker.", dims.t_layers), ("pred.", dims.p_layers)]
    if all(w % 32 == 0 and w >= 256 for w in (dims.c_hidden, dims.c_ffn)):
        stacks.append(("codec.", dims.c_layers))      # the codec's pre-transformer (B x chunk-frames rows, 128-row slabs)
    for pre, layers in stacks:
        for l in range(layers):
            q = f"{pre}{l}."
            tensors[q + "qkv.f"] = native.to_frag(tensors[q + "qkv.w"], tensors[q + "ln1"])
            tensors[q + "gu.f"] = native.to_frag(tensors[q + "gu.w"], tensors[q + "ln2"])
            tensors[q + "o.f"] = native.to_frag(tensors[q + "o.w"])
            tensors[q + "down.f"] = native.to_frag(tensors[q + "down.w"])
    for g in range(dims.n_groups - 1):
        tensors[f"pred.head.{g}.f"] = native.to_frag(tensors[f"pred.head.{g}.w"], tensors["pred.norm"])
    if P != E:
        tensors["pred.in_proj.f"] = native.to_frag(tensors["pred.in_proj.w"])
    torch.cuda.current_stream().synchronize()      # (a stream, not the device: never collide with another thread's capture)


def resample_fir_2_3(dtype=np.float32) -> np.ndarray:
    """The prototype low-pass scipy.signal.resample_poly(x, 2, 3) designs (firwin(61, 1/3, kaiser 5.0) * up)."""
    from scipy.signal import firwin

    half_len = 10 * 3
    return (firwin(2 * half_len + 1, 1.0 / 3.0, window=("kaiser", 5.0)) * 2.0).astype(dtype)


@dataclass
class TTSPrompt:
    """Prefill rows of one utterance: float32 [P, t_hidden] (text-side + codec-side embeddings already summed).
    `trailing` (optional, float32 [n, t_hidden]): streaming-text mode - the text rows that are NOT pre-filled (the prompt then ends with the
    first text token); the frame loop adds row f to the input of the step that produces frame f + 1 and `pad_embed` afterwards
    (reference `non_streaming_mode = False`; C ABI s2s_tts_set_trailing)."""

    embeds: np.ndarray
    trailing: np.ndarray | None = None


class AudioBlocker:
    """The reference's per-utterance output shaping (Qwen3TTSHandler._stream, TTS/qwen3_tts_handler.py:716-742):
    drop the leading silence (|x| <= 0.01*32768) keeping 40 ms of pre-roll, re-block to 512 samples, pad the tail."""

    def __init__(self, blocksize: int = BLOCKSIZE):
        self.blocksize = blocksize
        self.found_speech = False
        self.leftover = np.array([], dtype=np.int16)

    def push(self, chunk16: np.ndarray) -> Iterator[np.ndarray]:
        if not self.found_speech:
            threshold = int(32768 * 0.01)
            above = np.abs(chunk16) > threshold
            if not np.any(above):
                return
            start_idx = max(0, int(np.argmax(above)) - int(PIPELINE_SR * 0.040))
            chunk16 = chunk16[start_idx:]
            self.found_speech = True
        chunk16 = np.concatenate([self.leftover, chunk16])
        n = (len(chunk16) // self.blocksize) * self.blocksize
        for i in range(0, n, self.blocksize):
            yield chunk16[i : i + self.blocksize]
        self.leftover = chunk16[n:]

    def flush(self) -> Iterator[np.ndarray]:
        if len(self.leftover) > 0:
            yield np.pad(self.leftover, (0, self.blocksize - len(self.leftover)))
            self.leftover = np.array([], dtype=np.int16)


class TTSEngine:
    """One Qwen3-TTS replica on one GPU: batched talker/code-predictor frame loop, chunked codec decode, resample."""

    def __init__(self, dims: Qwen3TTSDims, state_dict: Mapping[str, torch.Tensor] | None, dtype: torch.dtype = torch.bfloat16,
                 max_batch: int = 8, max_prompt: int = 128, max_frames: int = 1536, max_codec_frames: int = 33,
                 pad_embed: torch.Tensor | None = None, device: str | torch.device = "cuda:0",
                 engine_tensors: Mapping[str, torch.Tensor] | None = None):
        self.dims, self.dtype, self.max_batch = dims, dtype, int(max_batch)
        self.device = torch.device(device)
        self.max_prompt, self.max_frames, self.max_codec_frames = int(max_prompt), int(max_frames), int(max_codec_frames)
        self.max_pos = self.max_prompt + self.max_frames
        self._lib = native.load()
        cfg = native.TTSConfigStruct()
        for k in ("t_hidden", "t_layers", "t_heads", "t_kv_heads", "t_head_dim", "t_ffn", "codec_vocab", "codec_eos", "p_hidden", "p_layers",
                  "p_heads", "p_kv_heads", "p_head_dim", "p_ffn", "n_groups", "codebook", "c_hidden", "c_layers", "c_heads", "c_ffn", "c_window"):
            setattr(cfg, k, int(getattr(dims, k)))
        cfg.c_decoder_dim = dims.c_decoder_dim
        cfg.n_dec_blocks = len(dims.upsample_rates)
        for i, r in enumerate(dims.upsample_rates):
            cfg.up_rate[i] = int(r)
        for i, r in enumerate(dims.upsampling_ratios):
            cfg.up_ratio[i] = int(r)
        cfg.total_upsample = dims.total_upsample
        cfg.max_prompt, cfg.max_frames, cfg.max_pos, cfg.max_codec_frames = self.max_prompt, self.max_frames, self.max_pos, self.max_codec_frames
        cfg.eps, cfg.c_eps, cfg.dtype = dims.eps, dims.c_eps, native.dtype_code(dtype)
        h = C.c_void_p()
        native.check(self._lib.s2s_tts_create(C.byref(cfg), self.max_batch, C.byref(h)), "s2s_tts_create")
        self._h = h
        if pad_embed is None:
            pad_embed = torch.zeros(dims.t_hidden)
        self._weights = {}
        with torch.cuda.device(self.device):
            tensors = engine_tensors if engine_tensors is not None else convert_tts_state_dict(
                state_dict, dims, dtype, self.max_pos, self.max_codec_frames, pad_embed)
            tensors = {name: t.to(self.device) for name, t in tensors.items()}
            if dtype == torch.bfloat16:
                add_fragment_major_weights(tensors, dims)
            for name, dt in tensors.items():
                self._weights[name] = dt
                native.check(self._lib.s2s_tts_set_tensor(h, name.encode(), native.ptr(dt), dt.numel()), name)
            native.check(self._lib.s2s_tts_finalize(h), "s2s_tts_finalize")
            self._fir = torch.from_numpy(resample_fir_2_3()).to(self.device)
        self._B = 0
        self._stream_frames = 0

    @classmethod
    def from_random(cls, dims: Qwen3TTSDims, dtype: torch.dtype = torch.bfloat16, max_batch: int = 8, max_prompt: int = 128, max_frames: int = 128,
                    max_codec_frames: int = 33, device: str | torch.device = "cuda:0", seed: int = 0) -> "TTSEngine":
        dev = torch.device(device)
        t = random_tts_engine_tensors(dims, dtype, dev, max_prompt + max_frames, max_codec_frames, seed)
        return cls(dims, None, dtype, max_batch, max_prompt, max_frames, max_codec_frames, None, dev, engine_tensors=t)

    def warmup(self, prefill_len: int = 100, frames: int = 8) -> None:
        """One throw-away utterance (the reference warms its backend with ``model.warmup(prefill_len=100)`` in setup(),
        tests/test_qwen3_tts_handler_backend.py:361-377): loads every kernel and captures the single-session frame graph."""
        n = max(2, min(int(prefill_len), self.max_prompt))
        f = max(1, min(int(frames), self.max_frames, self.max_codec_frames))
        prompt = TTSPrompt(np.zeros((n, self.dims.t_hidden), np.float32))
        for _ in self.stream([prompt], [f], chunk_size=f, min_frames=f):
            pass

    def replica(self, max_batch: int | None = None) -> "TTSEngine":
        """A second engine instance on the same GPU that shares this one's weight tensors (no copy) and owns only its KV
        caches / workspace / decode graphs - one per TTS lane of the session pool (max_batch: its own slot count, e.g. a small nursery)."""
        rep = TTSEngine(self.dims, None, self.dtype, int(max_batch or self.max_batch), self.max_prompt, self.max_frames, self.max_codec_frames, None, self.device,
                        engine_tensors=self._weights)
        for which, kw in getattr(self, "_sampling", {}).items():      # lanes sample like the engine they were cloned from
            rep.set_sampling(which, **kw)
        return rep

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.s2s_tts_workspace_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2s_tts_destroy(self._h)
            self._h = None

    # ---- stages --------------------------------------------------------------------------------------
    def prefill(self, prompts: Sequence[TTSPrompt]) -> None:
        B = len(prompts)
        lens = [p.embeds.shape[0] for p in prompts]
        Tp = max(lens)
        host = torch.zeros((B, Tp, self.dims.t_hidden), dtype=torch.float32)
        for i, p in enumerate(prompts):
            host[i, : lens[i]] = torch.from_numpy(np.ascontiguousarray(p.embeds, dtype=np.float32))
        emb = host.to(self.device)
        ln = torch.tensor(lens, dtype=torch.int32).to(self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_prefill(self._h, native.stream_handle(), native.ptr(emb), native.ptr(ln), B, Tp), "s2s_tts_prefill")
        self._B = B
        self._keep = (emb, ln)
        self._set_trailing(prompts, 0)

    def _set_trailing(self, prompts: Sequence[TTSPrompt], slot0: int) -> None:
        """Streaming-text rows of the prompts just admitted into the slots [slot0, ...): one call for the group (sessions without rows get 0)."""
        rows = [None if p.trailing is None or len(p.trailing) == 0 else np.ascontiguousarray(p.trailing, dtype=np.float32) for p in prompts]
        if all(r is None for r in rows):
            return
        E = self.dims.t_hidden
        Tr = max(r.shape[0] for r in rows if r is not None)
        if Tr > self.max_prompt:
            raise ValueError(f"{Tr} trailing text rows exceed the engine's max_prompt {self.max_prompt}")
        host = torch.zeros((len(prompts), Tr, E), dtype=torch.float32)
        for i, r in enumerate(rows):
            if r is not None:
                host[i, : r.shape[0]] = torch.from_numpy(r)
        dev_rows = host.to(self.device)
        n_rows = torch.tensor([0 if r is None else r.shape[0] for r in rows], dtype=torch.int32).to(self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_set_trailing(self._h, native.stream_handle(), int(slot0), len(prompts), native.ptr(dev_rows), native.ptr(n_rows), Tr),
                         "s2s_tts_set_trailing")
        self._keep_trailing = (dev_rows, n_rows)

    def generate(self, n_frames: int, min_total_frames: int = 0, check_done: bool = True):
        """-> (codes int32 [B][n_frames][G] on the GPU, valid frames so far per session (numpy), all_done)."""
        B, G = self._B, self.dims.n_groups
        codes = torch.empty((B, n_frames, G), dtype=torch.int32, device=self.device)
        nfr = torch.empty((B,), dtype=torch.int32, device=self.device)
        done = C.c_int32(0)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_generate(self._h, native.stream_handle(), n_frames, int(min_total_frames), native.ptr(codes),
                                                    native.ptr(nfr), C.byref(done) if check_done else None), "s2s_tts_generate")
        return codes, nfr, bool(done.value)

    # ---- continuous batching: slots of the frame loop (csrc/tts.cpp s2s_tts_admit / _step / _read_codes / _release) -------------
    def admit(self, prompts: Sequence[TTSPrompt], slot0: int, max_new_tokens: Sequence[int] | int, min_frames: Sequence[int] | int = 0,
              utterance_id0: int = 0) -> None:
        """Prefill `prompts` into the slots [slot0, slot0 + len(prompts)) while the other slots keep their state."""
        if len(prompts) > self.ADMIT_GROUP:      # a burst: groups of ADMIT_GROUP prompts (the prefill GEMMs are chip-filling long before that)
            per = lambda v, a, b: v if isinstance(v, int) else list(v)[a:b]     # noqa: E731
            for a in range(0, len(prompts), self.ADMIT_GROUP):
                b = min(len(prompts), a + self.ADMIT_GROUP)
                self.admit(prompts[a:b], slot0 + a, per(max_new_tokens, a, b), per(min_frames, a, b), utterance_id0 + a)
            return
        n = len(prompts)
        lens = [p.embeds.shape[0] for p in prompts]
        Tp = max(lens)
        # one pinned staging block per engine, allocated on first use (a pinned allocation synchronises the device: never while serving),
        # used as a ring: consecutive admissions - several cohorts at one chunk boundary when the free slots are fragmented - take
        # disjoint regions and do NOT wait for each other's prefill; the stream is synchronised only when the ring wraps
        E = self.dims.t_hidden
        self.prepare_admission()
        need = n * Tp * E
        if self._admit_off + need > self._admit_host.numel() or self._admit_row >= self._admit_ints.shape[0]:
            torch.cuda.current_stream(self.device).synchronize()      # every copy out of the ring has completed
            self._admit_off, self._admit_row = 0, 0
        host = self._admit_host[self._admit_off: self._admit_off + need].view(n, Tp, E)
        self._admit_off += need
        host.zero_()
        for i, p in enumerate(prompts):
            host[i, : lens[i]] = torch.from_numpy(np.ascontiguousarray(p.embeds, dtype=np.float32))
        emb = host.to(self.device, non_blocking=True)
        ints = self._admit_ints[self._admit_row]
        self._admit_row += 1
        ints[0, :n] = torch.tensor(lens, dtype=torch.int32)
        ln = ints[0, :n].to(self.device, non_blocking=True)

        def per(row, v):
            if isinstance(v, int):
                return None
            ints[row, :n] = torch.tensor([int(x) for x in v], dtype=torch.int32)
            return ints[row, :n].to(self.device, non_blocking=True)

        mn, bd = per(1, min_frames), per(2, max_new_tokens)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_admit(self._h, native.stream_handle(), native.ptr(emb), native.ptr(ln), n, Tp, int(slot0), native.ptr(mn), native.ptr(bd),
                                                 int(min_frames) if mn is None else 0, int(max_new_tokens) if bd is None else self.max_frames,
                                                 int(utterance_id0)), "s2s_tts_admit")
        self._B = max(self._B, slot0 + n)
        self._keep_admit = (emb, ln, mn, bd)
        self._set_trailing(prompts, slot0)

    ADMIT_GROUP = 64                 # most prompts per s2s_tts_admit call (bounds the pinned staging ring: 2 x 64 x max_prompt x E floats)

    def prepare_admission(self) -> None:
        """Allocate the pinned staging ring of `admit` (once; the session pool calls it while the lane warms up, not on the first request)."""
        if getattr(self, "_admit_host", None) is None:
            group = min(self.max_batch, self.ADMIT_GROUP)
            self._admit_host = torch.empty((2 * group * self.max_prompt * self.dims.t_hidden,), dtype=torch.float32, pin_memory=True)
            self._admit_ints = torch.zeros((64, 3, group), dtype=torch.int32, pin_memory=True)
            self._admit_off, self._admit_row = 0, 0

    def step(self, rows: int, n_frames: int) -> None:
        """n_frames frames for the slots [0, rows): every live slot advances by its own frame index."""
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_step(self._h, native.stream_handle(), int(rows), int(n_frames)), "s2s_tts_step")

    def read_codes(self, slot0: int, n: int, frame0: int, n_frames: int):
        """-> (codes int32 [n][n_frames][G], valid frames so far int32 [n], valid 24 kHz samples of this chunk int32 [n]) of the slots
        [slot0, slot0 + n), all on the GPU."""
        codes = torch.empty((n, n_frames, self.dims.n_groups), dtype=torch.int32, device=self.device)
        nfr = torch.empty((n,), dtype=torch.int32, device=self.device)
        nvalid = torch.empty((n,), dtype=torch.int32, device=self.device)
        chunk_samples = self.codec_samples(n_frames) if frame0 == 0 else n_frames * self.dims.total_upsample
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_read_codes(self._h, native.stream_handle(), int(slot0), int(n), int(frame0), int(n_frames), native.ptr(codes),
                                                      native.ptr(nfr), int(chunk_samples), native.ptr(nvalid)), "s2s_tts_read_codes")
        return codes, nfr, nvalid

    def read_codes_rows(self, slot0: int, n: int, frame0_rows: torch.Tensor, n_frames: int):
        """`read_codes` for rows of different ages: frame0_rows int32 [n] on the GPU = first frame to read per slot (continuation chunks)."""
        assert frame0_rows.dtype == torch.int32 and frame0_rows.numel() == n and frame0_rows.is_contiguous()
        codes = torch.empty((n, n_frames, self.dims.n_groups), dtype=torch.int32, device=self.device)
        nfr = torch.empty((n,), dtype=torch.int32, device=self.device)
        nvalid = torch.empty((n,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_read_codes_rows(self._h, native.stream_handle(), int(slot0), int(n), native.ptr(frame0_rows), int(n_frames),
                                                           native.ptr(codes), native.ptr(nfr), int(n_frames * self.dims.total_upsample), native.ptr(nvalid)),
                         "s2s_tts_read_codes_rows")
        return codes, nfr, nvalid

    def codec_decode_rows_at(self, codes: torch.Tensor, slot0: int, frames_done_rows: torch.Tensor, max_frames_done: int) -> torch.Tensor:
        """Streaming codec continuation of the rows [slot0, slot0 + B) in ONE pass, each row at its own stream position
        (frames_done_rows int32 [B] on the GPU; rows without a live stream: any value, output ignored)."""
        B, F, _ = codes.shape
        assert codes.dtype == torch.int32 and codes.is_contiguous() and frames_done_rows.numel() == B
        n = F * self.dims.total_upsample
        wav = torch.empty((B, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_codec_decode_rows_at(self._h, native.stream_handle(), native.ptr(codes), int(slot0), B, F, native.ptr(frames_done_rows),
                                                            int(max_frames_done), native.ptr(wav), n), "s2s_codec_decode_rows_at")
        return wav

    def migrate_from(self, src: "TTSEngine", src_slot0: int, dst_slot0: int, n: int, n_pos: int, n_codec_frames: int, x_rows: torch.Tensor | None = None) -> None:
        """Adopt n live sessions from `src`'s slots [src_slot0, ..) into this engine's slots [dst_slot0, ..) (csrc/tts.cpp s2s_tts_migrate);
        the current stream must already wait for `src`'s last launch.  x_rows: `src.read_x` snapshot of those slots (float32 [n][t_hidden])."""
        assert x_rows is None or (x_rows.dtype == torch.float32 and x_rows.is_contiguous() and x_rows.shape == (n, self.dims.t_hidden))
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_migrate(self._h, src._h, native.stream_handle(), int(src_slot0), int(dst_slot0), int(n), int(n_pos),
                                                   int(n_codec_frames), native.ptr(x_rows)), "s2s_tts_migrate")
        self._B = max(self._B, dst_slot0 + n)
        self._keep_mig = x_rows

    def read_x(self, slot0: int, n: int) -> torch.Tensor:
        """Snapshot of the next talker input rows of the slots [slot0, slot0 + n) (taken before the slots are parked)."""
        out = torch.empty((n, self.dims.t_hidden), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_read_x(self._h, native.stream_handle(), int(slot0), int(n), native.ptr(out)), "s2s_tts_read_x")
        return out

    def release(self, slot0: int, n: int) -> None:
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_release(self._h, native.stream_handle(), int(slot0), int(n)), "s2s_tts_release")

    def codec_decode_rows(self, codes: torch.Tensor, slot0: int, frames_done: int) -> torch.Tensor:
        """Streaming codec continuation of the cohort in slots [slot0, slot0 + B): codes int32 [B][F][G], `frames_done` frames already decoded."""
        B, F, _ = codes.shape
        assert codes.dtype == torch.int32 and codes.is_contiguous()
        n = self.codec_samples(F) if frames_done == 0 else F * self.dims.total_upsample
        wav = torch.empty((B, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_codec_decode_rows(self._h, native.stream_handle(), native.ptr(codes), int(slot0), B, F, int(frames_done), native.ptr(wav), n),
                         "s2s_codec_decode_rows")
        return wav

    def generate_forced(self, forced_codes: torch.Tensor, min_total_frames: int = 0, want_logits: bool = True):
        """Teacher-forced frames: forced_codes int32 [B][n][G] (entries < 0 keep the greedy pick).  -> (emitted codes int32
        [B][n][G], logits float32 [B][n][G][V] with V = max(codec_vocab, codebook); group 0 rows hold codec_vocab values)."""
        B, n, G = forced_codes.shape
        assert B == self._B and G == self.dims.n_groups and forced_codes.dtype == torch.int32
        forced = forced_codes.to(self.device).contiguous()
        V = max(self.dims.codec_vocab, self.dims.codebook)
        logits = torch.zeros((B, n, G, V), dtype=torch.float32, device=self.device) if want_logits else None
        codes = torch.empty((B, n, G), dtype=torch.int32, device=self.device)
        nfr = torch.empty((B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_tts_generate_forced(self._h, native.stream_handle(), n, int(min_total_frames), native.ptr(forced), native.ptr(logits), V,
                                                           native.ptr(codes), native.ptr(nfr)), "s2s_tts_generate_forced")
        return codes, logits

    def set_sampling(self, which: str = "talker", temperature: float = 0.0, top_k: int = 0, top_p: float = 1.0, repetition_penalty: float = 1.0,
                     seed: int = 0) -> None:
        """Seeded sampling for the talker (codebook 0) or the code predictor (residual groups); temperature <= 0 = greedy (default)."""
        native.check(self._lib.s2s_tts_set_sampling(self._h, {"talker": 0, "predictor": 1}[which], float(temperature), int(top_k), float(top_p),
                                                    float(repetition_penalty), int(seed)), "s2s_tts_set_sampling")
        if not hasattr(self, "_sampling"):
            self._sampling = {}
        self._sampling[which] = dict(temperature=temperature, top_k=top_k, top_p=top_p, repetition_penalty=repetition_penalty, seed=seed)

    def configure_sampling(self, gen_kwargs: dict | None) -> None:
        """Apply a handler's ``gen_kwargs`` (do_sample / temperature / top_k / top_p / repetition_penalty / seed) to the talker and
        the code predictor; ``do_sample`` false or absent keeps greedy decoding."""
        g = gen_kwargs or {}
        if not g.get("do_sample"):
            return
        for i, which in enumerate(("talker", "predictor")):
            self.set_sampling(which, temperature=float(g.get("temperature", 0.9)), top_k=int(g.get("top_k", 50)), top_p=float(g.get("top_p", 1.0)),
                              repetition_penalty=float(g.get("repetition_penalty", 1.05)) if which == "talker" else 1.0, seed=int(g.get("seed", 0)) + i)

    def codec_samples(self, F: int) -> int:
        return int(self._lib.s2s_codec_samples(self._h, F))

    def codec_decode(self, codes: torch.Tensor, ctx: int = 0) -> torch.Tensor:
        """codes int32 [B][F][G] (first ctx frames = left context) -> float32 [B][n] 24 kHz samples of the new frames."""
        B, F, _ = codes.shape
        assert codes.dtype == torch.int32 and codes.is_contiguous()
        n = self.codec_samples(F) - ctx * self.dims.total_upsample
        wav = torch.empty((B, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_codec_decode(self._h, native.stream_handle(), native.ptr(codes), B, F, ctx, native.ptr(wav), n), "s2s_codec_decode")
        return wav

    def codec_stream_reset(self) -> None:
        native.check(self._lib.s2s_codec_stream_reset(self._h), "s2s_codec_stream_reset")
        self._stream_frames = 0

    def codec_stream_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """Exact causal continuation: codes int32 [B][F_new][G] -> the float32 24 kHz samples a whole-utterance decode adds."""
        B, F, _ = codes.shape
        assert codes.dtype == torch.int32 and codes.is_contiguous()
        n = self.codec_samples(F) if self._stream_frames == 0 else F * self.dims.total_upsample
        wav = torch.empty((B, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_codec_stream_decode(self._h, native.stream_handle(), native.ptr(codes), B, F, native.ptr(wav), n), "s2s_codec_stream_decode")
        self._stream_frames += F
        return wav

    def resample_int16(self, wav24: torch.Tensor, n_valid: torch.Tensor | None = None, want_float: bool = False):
        """Per-row scipy.signal.resample_poly(x, 2, 3) then clip(x*32768).astype(int16) -> int16 [B][ceil(2n/3)]."""
        B, n = wav24.shape
        n_out = (2 * n + 2) // 3
        out16 = torch.empty((B, n_out), dtype=torch.int16, device=self.device)
        outf = torch.empty((B, n_out), dtype=torch.float32, device=self.device) if want_float else None
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_resample_24k16k_int16(native.stream_handle(), native.ptr(wav24), wav24.stride(0), native.ptr(n_valid),
                                                            native.ptr(self._fir), self._fir.numel(), (self._fir.numel() - 1) // 2,
                                                            native.ptr(out16), native.ptr(outf), n_out, n_out, B), "s2s_resample_24k16k_int16")
        return (out16, outf) if want_float else out16

    # ---- utterance-level streaming (what the handler consumes) ------------------------------------------
    def stream(self, prompts: Sequence[TTSPrompt], max_new_tokens: Sequence[int], chunk_size: int = 8, ctx_frames: int = 25,
               min_frames: int = 0) -> Iterator[list[np.ndarray | None]]:
        """Yields, every `chunk_size` frames, one int16 16 kHz numpy chunk per session (None once a session ended)."""
        B = len(prompts)
        self.prefill(prompts)
        self.codec_stream_reset()
        budget = max(max_new_tokens)
        produced = 0
        while produced < budget:
            n = min(chunk_size, budget - produced)
            codes, nfr, all_done = self.generate(n, min_total_frames=min_frames)
            nfr_h = nfr.cpu().numpy()
            if ctx_frames < 0:
                raise ValueError("ctx_frames must be >= 0")
            wav = self.codec_stream_decode(codes)     # exact causal streaming: nothing is recomputed (codes < 0 after a session's end embed as code 0 in the kernel)
            pcm16 = self.resample_int16(wav).cpu().numpy()
            out: list[np.ndarray | None] = []
            for b in range(B):
                valid = int(min(nfr_h[b], max_new_tokens[b])) - produced
                if valid <= 0:
                    out.append(None)
                elif valid >= n:
                    out.append(pcm16[b])
                else:   # the utterance ended inside this chunk: keep only the samples of its valid frames
                    out.append(pcm16[b][: (2 * valid * self.dims.total_upsample + 2) // 3])
            produced += n
            yield out
            if all_done:
                break
