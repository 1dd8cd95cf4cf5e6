"""Host side of the Whisper engine (device graph: csrc/whisper.cpp).

WEIGHT LAYOUT handed to ``s2s_whisper_set_tensor`` (``W`` = model dtype bf16|f32, biases / LayerNorm / the
encoder position table always float32):

  enc.conv1.w [d][3][n_mels] W   (conv1.weight permuted tap-major)     enc.conv1.b [d]
  enc.conv2.w [d][3][d]      W                                         enc.conv2.b [d]
  enc.pos     [1500][d] f32  (fixed sinusoids)                         enc.ln_post.{g,b}
  enc.L.ln1.{g,b}  enc.L.qkv.w [3d][d] W = q|k|v   enc.L.qkv.b [3d] = qb|0|vb   enc.L.o.{w,b}
  enc.L.ln2.{g,b}  enc.L.fc1.{w,b}  enc.L.fc2.{w,b}
  dec.tok [vocab][d] W (also the logits matrix)   dec.pos [448][d] W   dec.ln.{g,b}
  dec.L.ln1  dec.L.qkv.{w,b}  dec.L.o.{w,b}  dec.L.ln2 (encoder_attn_layer_norm)  dec.L.cq.{w,b}
  dec.L.ckv.w [2d][d] = k|v   dec.L.ckv.b = 0|vb   dec.L.co.{w,b}   dec.L.ln3 (final_layer_norm)  dec.L.fc1  dec.L.fc2
"""

from __future__ import annotations

import ctypes as C
from typing import Mapping, Sequence

import numpy as np
import torch

from . import native
from .frontend import LogMelFrontend, pack_pcm
from .synth_weights import WhisperDims


def convert_whisper_state_dict(sd: Mapping[str, torch.Tensor], dims: WhisperDims, dtype: torch.dtype) -> dict[str, torch.Tensor]:
    """transformers-named float32 state dict -> engine tensors (CPU)."""
    out: dict[str, torch.Tensor] = {}
    d = dims.d_model
    W = lambda t: t.to(dtype).contiguous()          # noqa: E731
    F32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
    e = "model.encoder."
    out["enc.conv1.w"] = W(sd[e + "conv1.weight"].permute(0, 2, 1))
    out["enc.conv1.b"] = F32(sd[e + "conv1.bias"])
    out["enc.conv2.w"] = W(sd[e + "conv2.weight"].permute(0, 2, 1))
    out["enc.conv2.b"] = F32(sd[e + "conv2.bias"])
    out["enc.pos"] = F32(sd[e + "embed_positions.weight"])
    out["enc.ln_post.g"] = F32(sd[e + "layer_norm.weight"])
    out["enc.ln_post.b"] = F32(sd[e + "layer_norm.bias"])

    def attn_qkv(prefix):
        w = torch.cat([sd[prefix + "q_proj.weight"], sd[prefix + "k_proj.weight"], sd[prefix + "v_proj.weight"]], 0)
        b = torch.cat([sd[prefix + "q_proj.bias"], torch.zeros(d), sd[prefix + "v_proj.bias"]], 0)
        return W(w), F32(b)

    def ln(dst, src):
        out[dst + ".g"] = F32(sd[src + ".weight"])
        out[dst + ".b"] = F32(sd[src + ".bias"])

    def lin(dst, src):
        out[dst + ".w"] = W(sd[src + ".weight"])
        out[dst + ".b"] = F32(sd[src + ".bias"])

    for l in range(dims.enc_layers):
        p, q = f"{e}layers.{l}.", f"enc.{l}."
        ln(q + "ln1", p + "self_attn_layer_norm")
        out[q + "qkv.w"], out[q + "qkv.b"] = attn_qkv(p + "self_attn.")
        lin(q + "o", p + "self_attn.out_proj")
        ln(q + "ln2", p + "final_layer_norm")
        lin(q + "fc1", p + "fc1")
        lin(q + "fc2", p + "fc2")
    dd = "model.decoder."
    out["dec.tok"] = W(sd[dd + "embed_tokens.weight"])
    out["dec.pos"] = W(sd[dd + "embed_positions.weight"])
    ln("dec.ln", dd + "layer_norm")
    for l in range(dims.dec_layers):
        p, q = f"{dd}layers.{l}.", f"dec.{l}."
        ln(q + "ln1", p + "self_attn_layer_norm")
        out[q + "qkv.w"], out[q + "qkv.b"] = attn_qkv(p + "self_attn.")
        lin(q + "o", p + "self_attn.out_proj")
        ln(q + "ln2", p + "encoder_attn_layer_norm")
        lin(q + "cq", p + "encoder_attn.q_proj")
        out[q + "ckv.w"] = W(torch.cat([sd[p + "encoder_attn.k_proj.weight"], sd[p + "encoder_attn.v_proj.weight"]], 0))
        out[q + "ckv.b"] = F32(torch.cat([torch.zeros(d), sd[p + "encoder_attn.v_proj.bias"]], 0))
        lin(q + "co", p + "encoder_attn.out_proj")
        ln(q + "ln3", p + "final_layer_norm")
        lin(q + "fc1", p + "fc1")
        lin(q + "fc2", p + "fc2")
    return out


def random_engine_tensors(dims: WhisperDims, dtype: torch.dtype, device: torch.device, seed: int = 0) -> dict[str, torch.Tensor]:
    """Random-init weights generated directly on the GPU in the engine layout (bench: no checkpoints offline)."""
    from .synth_weights import sinusoids

    g = torch.Generator(device=device).manual_seed(seed)
    d, f = dims.d_model, dims.ffn
    out: dict[str, torch.Tensor] = {}

    def w(name, rows, cols, std=None):
        out[name] = (torch.randn((rows, cols), generator=g, device=device, dtype=torch.float32) * (std or cols ** -0.5)).to(dtype)

    def b(name, n, std=0.02):
        out[name] = torch.randn((n,), generator=g, device=device, dtype=torch.float32) * std

    def ln(name):
        out[name + ".g"] = 1.0 + 0.05 * torch.randn((d,), generator=g, device=device)
        out[name + ".b"] = 0.02 * torch.randn((d,), generator=g, device=device)

    w("enc.conv1.w", d, 3 * dims.n_mels); b("enc.conv1.b", d)
    w("enc.conv2.w", d, 3 * d); b("enc.conv2.b", d)
    out["enc.pos"] = sinusoids(dims.max_source, d).to(device)
    ln("enc.ln_post")
    for l in range(dims.enc_layers):
        q = f"enc.{l}."
        ln(q + "ln1"); w(q + "qkv.w", 3 * d, d); b(q + "qkv.b", 3 * d); w(q + "o.w", d, d); b(q + "o.b", d)
        ln(q + "ln2"); w(q + "fc1.w", f, d); b(q + "fc1.b", f); w(q + "fc2.w", d, f); b(q + "fc2.b", d)
    w("dec.tok", dims.vocab, d, 0.3); w("dec.pos", dims.max_target, d, 0.3); ln("dec.ln")
    for l in range(dims.dec_layers):
        q = f"dec.{l}."
        ln(q + "ln1"); w(q + "qkv.w", 3 * d, d); b(q + "qkv.b", 3 * d); w(q + "o.w", d, d); b(q + "o.b", d)
        ln(q + "ln2"); w(q + "cq.w", d, d); b(q + "cq.b", d); w(q + "ckv.w", 2 * d, d); b(q + "ckv.b", 2 * d)
        w(q + "co.w", d, d); b(q + "co.b", d)
        ln(q + "ln3"); w(q + "fc1.w", f, d); b(q + "fc1.b", f); w(q + "fc2.w", d, f); b(q + "fc2.b", d)
    return out


class WhisperEngine:
    """One Whisper replica on one GPU: log-mel front-end + encoder + greedy decoder for batches of sessions."""

    def __init__(self, dims: WhisperDims, state_dict: Mapping[str, torch.Tensor] | None, dtype: torch.dtype = torch.bfloat16,
                 max_batch: int = 8, device: str | torch.device = "cuda:0", engine_tensors: Mapping[str, torch.Tensor] | None = None):
        """state_dict: transformers-named float32 tensors (converted here), or engine_tensors: already in the
        engine layout on the device (see random_engine_tensors)."""
        self.dims, self.dtype, self.max_batch = dims, dtype, int(max_batch)
        self.device = torch.device(device)
        self._lib = native.load()
        cfg = native.WhisperConfigStruct(dims.n_mels, dims.d_model, dims.n_heads, dims.enc_layers, dims.dec_layers,
                                         dims.ffn, dims.vocab, dims.max_target, native.dtype_code(dtype))
        h = C.c_void_p()
        native.check(self._lib.s2s_whisper_create(C.byref(cfg), self.max_batch, C.byref(h)), "s2s_whisper_create")
        self._h = h
        self._weights = {}
        with torch.cuda.device(self.device):
            tensors = engine_tensors if engine_tensors is not None else convert_whisper_state_dict(state_dict, dims, dtype)
            tensors = {name: t.to(self.device) for name, t in tensors.items()}
            if dtype == torch.bfloat16 and dims.d_model % 32 == 0 and dims.ffn % 32 == 0 and dims.d_model >= 256 and "dec.tok.f" not in tensors:
                # decode-path copies in the fragment-major layout of csrc/gemm_dec.hip, shared by replicas through _weights
                for l in range(dims.dec_layers):
                    for nm in ("qkv.w", "o.w", "cq.w", "co.w", "fc1.w", "fc2.w"):
                        tensors[f"dec.{l}.{nm}.f"] = native.to_frag(tensors[f"dec.{l}.{nm}"].view(-1, dims.ffn if nm == "fc2.w" else dims.d_model))
                tensors["dec.tok.f"] = native.to_frag(tensors["dec.tok"].view(dims.vocab, dims.d_model))
                # LayerNorm-free decoder blocks (csrc/gemm_dec.hip LayerNorm prologue): the projection behind each LayerNorm with gamma folded
                # into its columns (".w.lf", one bf16 rounding of W * gamma), beta into its bias (".lb" = b + W beta) and the column sums of
                # the folded weights (".cs", what the mean term multiplies)
                d = dims.d_model
                for l in range(dims.dec_layers):
                    for nm, lnn in (("qkv", "ln1"), ("cq", "ln2"), ("fc1", "ln3")):
                        W = tensors[f"dec.{l}.{nm}.w"].view(-1, d)
                        gam, bet = tensors[f"dec.{l}.{lnn}.g"].float(), tensors[f"dec.{l}.{lnn}.b"].float()
                        tensors[f"dec.{l}.{nm}.w.lf"] = native.to_frag(W, gam)
                        tensors[f"dec.{l}.{nm}.lb"] = (tensors[f"dec.{l}.{nm}.b"].float() + W.float() @ bet).contiguous()
                        tensors[f"dec.{l}.{nm}.cs"] = (W.float() * gam).to(torch.bfloat16).float().sum(1).contiguous()
                torch.cuda.current_stream().synchronize()      # (a stream, not the device: never collide with another thread's capture)
            for name, dt in tensors.items():
                self._weights[name] = dt   # keep alive: the engine holds raw pointers
                native.check(self._lib.s2s_whisper_set_tensor(h, name.encode(), native.ptr(dt), dt.numel()), name)
            native.check(self._lib.s2s_whisper_finalize(h), "s2s_whisper_finalize")
        self.frontend = LogMelFrontend("whisper", dims.n_mels, max_batch=self.max_batch, device=self.device)

    @classmethod
    def from_random(cls, dims: WhisperDims, dtype: torch.dtype = torch.bfloat16, max_batch: int = 8,
                    device: str | torch.device = "cuda:0", seed: int = 0) -> "WhisperEngine":
        dev = torch.device(device)
        return cls(dims, None, dtype, max_batch, dev, engine_tensors=random_engine_tensors(dims, dtype, dev, seed))

    def replica(self, max_batch: int | None = None) -> "WhisperEngine":
        """A second engine instance on the same GPU sharing this one's weight tensors (no copy); it owns only its workspace (activations,
        self / cross KV caches, decode graphs) and front-end - one per extra STT worker of the session pool."""
        return WhisperEngine(self.dims, None, self.dtype, int(max_batch or self.max_batch), self.device, engine_tensors=self._weights)

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.s2s_whisper_workspace_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2s_whisper_destroy(self._h)
            self._h = None
            self.frontend.close()

    # ---- stages ---------------------------------------------------------------------------
    def features(self, audios: Sequence[np.ndarray]) -> torch.Tensor:
        """float32 utterances -> encoder-layout log-mel [B][3002][n_mels] in the model dtype."""
        B = len(audios)
        # engine-owned pinned staging block (a pinned allocation per batch would synchronise the device under the TTS lanes); the previous
        # batch's copy out of it completed before its results were read back
        lens = [min(len(a), 480000) for a in audios]
        stride = max(8, (max(lens) + 7) // 8 * 8) if lens else 8
        if getattr(self, "_pcm_host", None) is None or self._pcm_host.numel() < B * stride:
            self._pcm_host = torch.empty((max(B, self.max_batch) * max(stride, 160000),), dtype=torch.float32, pin_memory=True)
            self._ns_host = torch.empty((max(B, self.max_batch),), dtype=torch.int32, pin_memory=True)
        host = self._pcm_host[: B * stride].view(B, stride)
        for i, (a, n) in enumerate(zip(audios, lens)):
            host[i, :n] = torch.from_numpy(np.ascontiguousarray(a[:n], dtype=np.float32))
            host[i, n:] = 0.0
        self._ns_host[:B] = torch.tensor(lens, dtype=torch.int32)
        pcm = host.to(self.device, non_blocking=True)
        ns = self._ns_host[:B].to(self.device, non_blocking=True)
        out = torch.empty((B, 3002, self.dims.n_mels), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_logmel_whisper_tm(self.frontend._handle, native.stream_handle(), native.ptr(pcm),
                                                         pcm.stride(0), native.ptr(ns), B, native.ptr(out),
                                                         native.dtype_code(self.dtype)), "s2s_logmel_whisper_tm")
        return out

    def encode(self, feats_tm: torch.Tensor) -> None:
        B = feats_tm.shape[0]
        assert feats_tm.shape[1:] == (3002, self.dims.n_mels) and feats_tm.dtype == self.dtype and feats_tm.is_contiguous()
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_encode(self._h, native.stream_handle(), native.ptr(feats_tm), B), "s2s_whisper_encode")
        self._B = B

    def encoder_output(self) -> torch.Tensor:
        out = torch.empty((self._B, 1500, self.dims.d_model), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_read_encoder_output(self._h, native.stream_handle(), native.ptr(out), self._B))
        return out

    def _mask(self, ids: Sequence[int] | None, invert: bool = False):
        if ids is None or len(ids) == 0:
            return None
        key = (tuple(int(i) for i in ids), bool(invert))
        cache = self.__dict__.setdefault("_mask_cache", {})
        if key not in cache:          # the masks of a vocabulary never change: built and uploaded once
            m = torch.ones(self.dims.vocab, dtype=torch.uint8) if invert else torch.zeros(self.dims.vocab, dtype=torch.uint8)
            m[list(key[0])] = 0 if invert else 1
            cache[key] = m.to(self.device)
        return cache[key]

    def decode(self, prompts: Sequence[Sequence[int]], eos: int, max_new_tokens: int, suppress: Sequence[int] | None = None,
               begin_suppress: Sequence[int] | None = None, lang_ids: Sequence[int] | None = None, min_steps: int = 0,
               return_logits: bool = False):
        """Greedy decode of the last encoded batch. -> (ids per session, filled prompts[, last-step logits])."""
        B, P = len(prompts), len(prompts[0])
        assert B == self._B
        pr = torch.tensor(prompts, dtype=torch.int32).to(self.device)
        out_ids = torch.zeros((B, max_new_tokens + P), dtype=torch.int32, device=self.device)
        out_len = torch.zeros((B,), dtype=torch.int32, device=self.device)
        sup, bsup, fill = self._mask(suppress), self._mask(begin_suppress), self._mask(lang_ids, invert=True)
        logits = torch.empty((B, self.dims.vocab), dtype=torch.float32, device=self.device) if return_logits else None
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_decode_greedy(self._h, native.stream_handle(), B, native.ptr(pr), P,
                                                             native.ptr(sup), native.ptr(bsup), native.ptr(fill), int(eos),
                                                             int(max_new_tokens), int(min_steps), native.ptr(out_ids),
                                                             native.ptr(out_len), native.ptr(logits)), "s2s_whisper_decode_greedy")
        ids_h, len_h = out_ids.cpu().numpy(), out_len.cpu().numpy()
        ids = [ids_h[b, : len_h[b]].tolist() for b in range(B)]
        filled = [[int(ids_h[b, max_new_tokens + i]) if prompts[b][i] < 0 else int(prompts[b][i]) for i in range(P)] for b in range(B)]
        if return_logits:
            return ids, filled, logits
        return ids, filled

    def stt_begin(self, audios: Sequence[np.ndarray], options: Sequence[dict]):
        """First stage of a batch (asynchronous on the current stream): log-mel + encoder + cross K/V of every session.  The session pool
        runs it on one stream while `stt_finish` of the PREVIOUS batch - the latency-bound token loop - runs on another (two engine
        replicas; SessionPool stt_stages)."""
        vocab = options[0]["vocab"]
        max_new = max(int(o.get("max_new_tokens", 128)) for o in options)
        vocab.validate(self.dims.vocab)
        prompts = [vocab.prompt(o.get("language"), o.get("task", "transcribe")) for o in options]
        self.encode(self.features(audios))
        return vocab, prompts, max_new, list(options)

    def stt_finish(self, state) -> list[dict]:
        """Second stage: greedy decode of the batch `stt_begin` encoded on THIS engine; one dict per request."""
        vocab, prompts, max_new, options = state
        ids, filled = self.decode(prompts, vocab.eot, max_new, suppress=vocab.suppress, begin_suppress=vocab.begin_suppress,
                                  lang_ids=list(vocab.lang_to_id.values()) or None)
        out = []
        for i, o in enumerate(options):
            n = int(o.get("max_new_tokens", 128))
            out.append({"ids": ids[i][:n], "language": vocab.id_to_lang(filled[i][1]) if vocab.multilingual else "en"})
        return out

    # ---- continuous batching of the token loop (csrc/whisper.cpp s2s_whisper_slots_*; scheduler: stt_continuous.py) ----------------
    SLOT_PMAX = 8

    def slots_config(self, vocab) -> None:
        """The running loop's vocabulary: suppress / begin-suppress / language-fill masks and eos (once, while no slot is live)."""
        vocab.validate(self.dims.vocab)
        sup, bsup = self._mask(vocab.suppress), self._mask(vocab.begin_suppress)
        fill = self._mask(list(vocab.lang_to_id.values()) or None, invert=True)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_slots_config(self._h, native.stream_handle(), native.ptr(sup), native.ptr(bsup), native.ptr(fill),
                                                            int(vocab.eot)), "s2s_whisper_slots_config")
        self._slot_vocab = vocab
        self._slot_stride = int(self._lib.s2s_whisper_slots_pack_stride(self._h))
        if getattr(self, "_slot_pack", None) is None:
            self._slot_pack = torch.zeros((self.max_batch, self._slot_stride), dtype=torch.int32, device=self.device)
            self._slot_pack_host = torch.zeros((self.max_batch, self._slot_stride), dtype=torch.int32, pin_memory=True)

    def slots_encode(self, audios: Sequence[np.ndarray], slot0: int) -> torch.Tensor:
        """Log-mel + encoder of len(audios) sessions on the CURRENT stream; their cross K|V land in slots slot0 ...  Returns the feature
        tensor (keep it referenced until the stream has passed this point)."""
        feats = self.features(audios)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_encode_at(self._h, native.stream_handle(), native.ptr(feats), len(audios), int(slot0)), "s2s_whisper_encode_at")
        return feats

    def slots_admit(self, slot0: int, prompts: Sequence[Sequence[int]], max_new: Sequence[int]) -> None:
        n, P = len(prompts), len(prompts[0])
        assert all(len(p) == P for p in prompts) and len(max_new) == n and 1 <= P <= self.SLOT_PMAX
        pr = torch.tensor(prompts, dtype=torch.int32).to(self.device)
        mn = torch.tensor([int(v) for v in max_new], dtype=torch.int32).to(self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_slots_admit(self._h, native.stream_handle(), int(slot0), n, native.ptr(pr), P, native.ptr(mn),
                                                           int(max(max_new))), "s2s_whisper_slots_admit")
        self._slot_keep = (pr, mn)          # (alive until the next admission: the kernel has long read them by then)

    def slots_step(self, rows: int, n_steps: int) -> None:
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_slots_step(self._h, native.stream_handle(), int(rows), int(n_steps)), "s2s_whisper_slots_step")

    def slots_collect(self, rows: int) -> np.ndarray:
        """-> int32 [rows][2 + 8 + max_target] on the host (synchronises the current stream): finished flag (0 | 1 eos | 2 budget), number of
        ids, the prompt tokens the model filled in, the ids - the last two valid for finished rows."""
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_slots_collect(self._h, native.stream_handle(), int(rows), native.ptr(self._slot_pack)), "s2s_whisper_slots_collect")
            self._slot_pack_host[:rows].copy_(self._slot_pack[:rows], non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return self._slot_pack_host[:rows].numpy()

    def slots_release(self, slot0: int, n: int) -> None:
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_whisper_slots_release(self._h, native.stream_handle(), int(slot0), int(n)), "s2s_whisper_slots_release")

    def slot_result(self, row: np.ndarray, prompt: Sequence[int], max_new: int) -> dict:
        """One finished row of `slots_collect` -> the dict `transcribe_batch` returns for that request."""
        vocab = self._slot_vocab
        n = int(row[1])
        ids = row[2 + self.SLOT_PMAX : 2 + self.SLOT_PMAX + n].tolist()[: int(max_new)]
        filled = [int(row[2 + i]) if prompt[i] < 0 else int(prompt[i]) for i in range(len(prompt))]
        return {"ids": ids, "language": vocab.id_to_lang(filled[1]) if vocab.multilingual else "en"}

    def transcribe_batch(self, audios: Sequence[np.ndarray], options: Sequence[dict]) -> list[dict]:
        """SessionPool entry point: one dict per request {"ids": [...], "language": code | None}.
        options per request: language (code | "auto" | None), max_new_tokens, vocab (text.WhisperVocab)."""
        return self.stt_finish(self.stt_begin(audios, options))

    def transcribe_ids(self, audios: Sequence[np.ndarray], prompt: Sequence[int], eos: int, max_new_tokens: int = 128, **kw):
        self.encode(self.features(audios))
        return self.decode([list(prompt)] * len(audios), eos, max_new_tokens, **kw)
