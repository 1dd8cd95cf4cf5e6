"""Host side of the Parakeet-TDT engine (device graph: csrc/parakeet.cpp, kernels csrc/conformer.hip).

WEIGHT LAYOUT handed to ``s2s_parakeet_set_tensor`` (``W`` = model dtype; everything else float32):

  sub.conv1.{w [C][9], b}   sub.dw2.{w [C][9], b}  sub.pw2.{w [C][C] W, b}  sub.dw3.*  sub.pw3.*
  sub.lin.w [d][F/8 * C] W  columns re-ordered to the channels-last flatten (f*C + c) and multiplied by
                            sqrt(d) when scale_input (exact: powers of two for d = 1024 / 256); sub.lin.b likewise
  enc.L.ln_{ff1,att,conv,ff2,out}.{g,b}
  enc.L.ff{1,2}.fc1.{w,b}  enc.L.ff{1,2}.fc2.{w,b}
  enc.L.qkv.w [3d][d] W = q|k|v     enc.L.qkv.b = (q_bias + bias_u)|k_bias|v_bias
  enc.L.vmu  [d] = bias_v - bias_u  (q+u is what the projection produces; q+v = that + vmu)
  enc.L.R    [2*Tm-1][d] W = relative_k_proj(sinusoid table for Tm frames) * head_dim**-0.5
  enc.L.o.{w,b}  enc.L.pw1.{w [2d][d], b}  enc.L.dw.{w [d][9], b} with eval-mode BatchNorm folded in  enc.L.pw2.{w,b}
  dec.encp.{w,b}  dec.emb [V][Hd] W  dec.lstm.L.w [4Hd][2Hd] W = W_ih|W_hh  dec.lstm.L.b = b_ih + b_hh
  dec.dproj.{w,b}  dec.joint.{w [V+5][Hd] W, b}
"""

from __future__ import annotations

import logging

import ctypes as C
import math
import zlib
from typing import Any, Mapping, Sequence

import numpy as np
import torch

from . import native
from .frontend import LogMelFrontend, pack_pcm
from .synth_weights import ParakeetDims

logger = logging.getLogger(__name__)

try:      # (xxh3 runs at memory speed; crc32 is the fallback - either one sees every byte of the prefix)
    import xxhash as _xxhash
except Exception:  # noqa: BLE001  pragma: no cover
    _xxhash = None


class _PrefixDigest:
    """Streaming checksum over the samples a ProgressiveStream has uploaded so far."""

    def __init__(self):
        self._h = _xxhash.xxh3_64() if _xxhash is not None else 0

    def update(self, samples: np.ndarray) -> None:
        buf = memoryview(np.ascontiguousarray(samples, dtype=np.float32)).cast("B")
        if _xxhash is not None:
            self._h.update(buf)
        else:
            self._h = zlib.crc32(buf, self._h)

    def value(self) -> int:
        return self._h.intdigest() if _xxhash is not None else self._h


def _half_up(n: int) -> int:
    return (n - 1) // 2 + 1


def encoder_frames(feat_frames: int) -> int:
    return _half_up(_half_up(_half_up(feat_frames)))


def rel_pos_table(T: int, d: int) -> torch.Tensor:
    """[2T-1, d] interleaved sin/cos of relative positions T-1 .. -(T-1)."""
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)
    fr = pos[:, None] * inv[None, :]
    return torch.stack([fr.sin(), fr.cos()], dim=-1).reshape(2 * T - 1, d)


def convert_parakeet_state_dict(sd: Mapping[str, torch.Tensor], dims: ParakeetDims, dtype: torch.dtype, tm_enc: int) -> dict[str, torch.Tensor]:
    out: dict[str, torch.Tensor] = {}
    d, C = dims.d_model, dims.sub_channels
    D = d // dims.n_heads
    F3 = dims.n_mels // 8
    W = lambda t: t.to(dtype).contiguous()            # noqa: E731
    F32 = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
    s = "encoder.subsampling."
    out["sub.conv1.w"] = F32(sd[s + "layers.0.weight"].reshape(C, 9)); out["sub.conv1.b"] = F32(sd[s + "layers.0.bias"])
    for tag, i in (("2", 2), ("3", 5)):
        out[f"sub.dw{tag}.w"] = F32(sd[s + f"layers.{i}.weight"].reshape(C, 9)); out[f"sub.dw{tag}.b"] = F32(sd[s + f"layers.{i}.bias"])
        out[f"sub.pw{tag}.w"] = W(sd[s + f"layers.{i + 1}.weight"].reshape(C, C)); out[f"sub.pw{tag}.b"] = F32(sd[s + f"layers.{i + 1}.bias"])
    scale = math.sqrt(d) if dims.scale_input else 1.0
    lw = sd[s + "linear.weight"].reshape(d, C, F3).permute(0, 2, 1).reshape(d, F3 * C) * scale
    out["sub.lin.w"] = W(lw); out["sub.lin.b"] = F32(sd[s + "linear.bias"] * scale)
    pos = rel_pos_table(tm_enc, d)

    def ln(dst, src):
        out[dst + ".g"] = F32(sd[src + ".weight"]); out[dst + ".b"] = F32(sd[src + ".bias"])

    def lin(dst, src):
        out[dst + ".w"] = W(sd[src + ".weight"]); out[dst + ".b"] = F32(sd[src + ".bias"])

    for l in range(dims.layers):
        p, q = f"encoder.layers.{l}.", f"enc.{l}."
        for a, b in (("ln_ff1", "norm_feed_forward1"), ("ln_att", "norm_self_att"), ("ln_conv", "norm_conv"),
                     ("ln_ff2", "norm_feed_forward2"), ("ln_out", "norm_out")):
            ln(q + a, p + b)
        for a, b in (("ff1", "feed_forward1"), ("ff2", "feed_forward2")):
            lin(q + a + ".fc1", p + b + ".linear1"); lin(q + a + ".fc2", p + b + ".linear2")
        at = p + "self_attn."
        u, v = sd[at + "bias_u"].reshape(d), sd[at + "bias_v"].reshape(d)
        out[q + "qkv.w"] = W(torch.cat([sd[at + "q_proj.weight"], sd[at + "k_proj.weight"], sd[at + "v_proj.weight"]], 0))
        out[q + "qkv.b"] = F32(torch.cat([sd[at + "q_proj.bias"] + u, sd[at + "k_proj.bias"], sd[at + "v_proj.bias"]], 0))
        out[q + "vmu"] = F32(v - u)
        out[q + "R"] = W((pos @ sd[at + "relative_k_proj.weight"].T) * D ** -0.5)
        lin(q + "o", at + "o_proj")
        cv = p + "conv."
        out[q + "pw1.w"] = W(sd[cv + "pointwise_conv1.weight"].reshape(2 * d, d)); out[q + "pw1.b"] = F32(sd[cv + "pointwise_conv1.bias"])
        bn_s = sd[cv + "norm.weight"] / torch.sqrt(sd[cv + "norm.running_var"] + 1e-5)
        bn_t = sd[cv + "norm.bias"] - sd[cv + "norm.running_mean"] * bn_s
        out[q + "dw.w"] = F32(sd[cv + "depthwise_conv.weight"].reshape(d, dims.conv_kernel) * bn_s[:, None])
        out[q + "dw.b"] = F32(sd[cv + "depthwise_conv.bias"] * bn_s + bn_t)
        out[q + "pw2.w"] = W(sd[cv + "pointwise_conv2.weight"].reshape(d, d)); out[q + "pw2.b"] = F32(sd[cv + "pointwise_conv2.bias"])
    lin("dec.encp", "encoder_projector")
    out["dec.emb"] = W(sd["decoder.embedding.weight"])
    for l in range(dims.dec_layers):
        out[f"dec.lstm.{l}.w"] = W(torch.cat([sd[f"decoder.lstm.weight_ih_l{l}"], sd[f"decoder.lstm.weight_hh_l{l}"]], 1))
        out[f"dec.lstm.{l}.b"] = F32(sd[f"decoder.lstm.bias_ih_l{l}"] + sd[f"decoder.lstm.bias_hh_l{l}"])
    lin("dec.dproj", "decoder.decoder_projector")
    lin("dec.joint", "joint.head")
    return out


class ProgressiveStream:
    """Host-side token of ONE growing window of live partials (reference STT/smart_progressive_streaming.py:87-157 re-transcribes the window from the
    last fixed sentence to "now" every 0.5 s).  The engine attaches the device state on first use: the window's samples so far (only the NEW
    samples of a partial cross PCIe) and its un-normalised log-mel frames (only the frames the new samples can change are computed).  Everything
    behind the log-mel - per-feature normalisation over the whole window, the full-context encoder, TDT - is recomputed, so a partial is
    bit-identical to transcribing the whole window from scratch (tests/test_parakeet_progressive_gpu.py)."""

    def __init__(self):
        self.key = None            # the window's start in the turn: a new start is a new window
        self.n = 0                 # samples on the device
        self.pcm = self.cache = None
        self.digest = None         # checksum of those n samples: a caller whose buffer changed BEHIND the tail restarts the stream
        self.stats = {"partials": 0, "samples_uploaded": 0, "samples_total": 0, "frames_computed": 0, "frames_total": 0, "restarts": 0}


class ParakeetEngine:
    """One Parakeet-TDT replica on one GPU: log-mel + FastConformer encoder + TDT greedy decode for session batches."""

    def __init__(self, dims: ParakeetDims, state_dict: Mapping[str, torch.Tensor] | None, dtype: torch.dtype = torch.bfloat16,
                 max_batch: int = 8, max_seconds: float = 30.0, device: str | torch.device = "cuda:0",
                 engine_tensors: Mapping[str, torch.Tensor] | None = None):
        self.dims, self.dtype, self.max_batch = dims, dtype, int(max_batch)
        self.device = torch.device(device)
        self.max_feat_frames = int(max_seconds * 100) + 1
        self._lib = native.load()
        cfg = native.ParakeetConfigStruct(dims.n_mels, dims.d_model, dims.n_heads, dims.layers, dims.ffn, dims.sub_channels,
                                          dims.conv_kernel, dims.vocab, dims.dec_hidden, dims.dec_layers, dims.n_durations,
                                          self.max_feat_frames, native.dtype_code(dtype))
        h = C.c_void_p()
        native.check(self._lib.s2s_parakeet_create(C.byref(cfg), self.max_batch, C.byref(h)), "s2s_parakeet_create")
        self._h = h
        self.tm_enc = int(self._lib.s2s_parakeet_max_encoder_frames(h))
        self._weights = {}
        with torch.cuda.device(self.device):
            tensors = engine_tensors if engine_tensors is not None else convert_parakeet_state_dict(state_dict, dims, dtype, self.tm_enc)
            for name, t in tensors.items():
                dt = t.to(self.device)
                self._weights[name] = dt
                native.check(self._lib.s2s_parakeet_set_tensor(h, name.encode(), native.ptr(dt), dt.numel()), name)
            native.check(self._lib.s2s_parakeet_finalize(h), "s2s_parakeet_finalize")
        self.frontend = LogMelFrontend("parakeet", dims.n_mels, max_batch=self.max_batch, device=self.device)
        self._B = 0
        self._rows_host = self._rows_dev = self._rows_event = None
        if self.device.type == "cuda":
            self._rows_alloc(self.max_batch * 32000, self.max_batch)       # (pinned allocations synchronise the device: not in serving time)
            size = 3 * self.max_batch * 10 * max(self.tm_enc, 8) + self.max_batch      # TDT results of the longest pass (tdt_greedy)
            self._tdt_dev = torch.empty((size,), dtype=torch.int32, device=self.device)
            self._tdt_host = torch.empty((size,), dtype=torch.int32, pin_memory=True)

    def _rows_alloc(self, total: int, B: int) -> None:
        """Staging of `stream_features_rows`: a pinned block + its device twin for the new samples of a tick, the int32 / pointer tables."""
        have = self._rows_host
        if have is not None and have[0].numel() >= total and have[1].shape[1] >= B + 1:
            return
        total = max(total, have[0].numel() if have else 0)
        W = max(B, self.max_batch, (have[1].shape[1] - 1) if have else 0) + 1
        self._rows_host = (torch.empty((total,), dtype=torch.float32, pin_memory=True), torch.empty((4, W), dtype=torch.int32, pin_memory=True),
                           torch.empty((2, W), dtype=torch.int64, pin_memory=True))
        self._rows_dev = (torch.empty((total,), dtype=torch.float32, device=self.device), torch.empty((4, W), dtype=torch.int32, device=self.device),
                          torch.empty((2, W), dtype=torch.int64, device=self.device))

    def replica(self, max_batch: int | None = None) -> "ParakeetEngine":
        """A second engine instance on the same GPU sharing this one's weight tensors (no copy of what was uploaded); it owns its workspace
        (activations, TDT state, derived layouts) and front-end - one per extra STT worker of the session pool (SessionPool stt_workers)."""
        return ParakeetEngine(self.dims, None, self.dtype, int(max_batch or self.max_batch), (self.max_feat_frames - 1) / 100.0, self.device,
                              engine_tensors=self._weights)

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.s2s_parakeet_workspace_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2s_parakeet_destroy(self._h)
            self._h = None
            self.frontend.close()

    def features(self, audios: Sequence[np.ndarray]):
        """-> (feats [B][T][n_mels] model dtype, n_frames int32 [B], T)."""
        pcm, ns = pack_pcm(audios, self.device)
        T = max(8, max(len(a) for a in audios) // 160 + 1)
        feats, nf = self.frontend.parakeet(pcm, ns, T, out_dtype=self.dtype)
        return feats, nf, T

    def _stream_advance(self, st: ProgressiveStream, audio: np.ndarray, key: Any) -> tuple[int, int, np.ndarray, int]:
        """Host part of a partial: is `audio` an extension of what the stream has on the device?  -> (old_n, first_frame, new samples, checksum
        of the whole window); the stream's device buffers exist afterwards (a restarted stream counts from 0)."""
        n = len(audio)
        cap = (self.max_feat_frames - 1) * 160
        assert 0 < n <= cap
        # "same growing window" = same key, not shorter, and the caller's first st.n samples are still the ones on the device: a checksum over
        # the WHOLE prefix (comparing only the last 64 samples missed a re-based / trimmed / enhanced buffer and a tail of digital
        # silence, and then reused stale device PCM and log-mel frames); one pass over the prefix, continued over the new samples below
        dg = _PrefixDigest()
        same = st.pcm is not None and st.key == key and st.n > 0 and n >= st.n
        if same:
            dg.update(audio[: st.n])
            same = dg.value() == st.digest
        if not same:
            if st.pcm is None:
                st.pcm = torch.zeros((1, (cap + 7) // 8 * 8), dtype=torch.float32, device=self.device)
                st.cache = torch.zeros((self.max_feat_frames, self.dims.n_mels), dtype=torch.float32, device=self.device)
            st.n, st.key = 0, key
            dg = _PrefixDigest()
            st.stats["restarts"] += 1
        old_n = st.n
        new = np.ascontiguousarray(audio[old_n:n], dtype=np.float32)
        if len(new):
            dg.update(new)
        first = max(0, (old_n - 200) // 160) if old_n > 0 else 0
        return old_n, first, new, dg.value()

    def _stream_account(self, st: ProgressiveStream, n: int, n_new: int, first: int, t_max: int, digest: int) -> None:
        st.stats["partials"] += 1
        st.stats["samples_uploaded"] += n_new
        st.stats["samples_total"] += n
        st.stats["frames_computed"] += (min(t_max, n // 160 + 1) - first // 64 * 64)       # (whole 64-frame tiles from the one that holds `first`)
        st.stats["frames_total"] += n // 160 + 1
        st.n, st.digest = n, digest

    def stream_features(self, st: ProgressiveStream, audio: np.ndarray, key: Any, t_max: int, out_row: torch.Tensor, nf_row: torch.Tensor) -> None:
        """Features of a growing window into row `out_row` [1][t_max][n_mels] of a batch: uploads the samples that are new since the stream's last
        partial and computes the log-mel frames they can change (frame t depends on samples t * 160 - 200 .. t * 160 + 199 and, through the
        pre-emphasis, one more to the left), the rest comes from the stream's cache."""
        n = len(audio)
        old_n, first, new, digest = self._stream_advance(st, audio, key)
        if len(new):
            st.pcm[0, old_n:n].copy_(torch.from_numpy(new))
        ns = torch.tensor([n], dtype=torch.int32, device=self.device)
        ff = torch.tensor([first], dtype=torch.int32, device=self.device)
        self.frontend.parakeet_cached(st.pcm, ns, ff, t_max, st.cache, out_row, nf_row)
        self._stream_account(st, n, len(new), first, t_max, digest)

    def stream_features_rows(self, streams: Sequence[ProgressiveStream], audios: Sequence[np.ndarray], keys: Sequence[Any], t_max: int, out: torch.Tensor,
                             nf: torch.Tensor) -> None:
        """The growing windows of a whole tick into out [B][t_max][n_mels]: per window only host work (checksum, copy of its new samples into a
        pinned block, a row of the tables); ONE upload of the new samples, one of the tables and one s2s_logmel_parakeet_cached_rows call for all
        of them - per-window uploads and launches were 43 % of a 48-window partial pass.  Row b is bit-identical to `stream_features` on it."""
        B = len(streams)
        assert B == out.shape[0] and out.is_contiguous()
        if self._rows_event is not None:
            self._rows_event.synchronize()             # the previous call's uploads and kernels are done with the staging blocks
        meta = [self._stream_advance(st, a, k) for st, a, k in zip(streams, audios, keys)]
        total = sum(len(m[2]) for m in meta)
        self._rows_alloc(max(total, 1), B)
        stage, ints, ptrs = self._rows_host
        stage_np, ints_np, ptrs_np = stage.numpy(), ints.numpy(), ptrs.numpy()
        off = 0
        for b, (st, a, (old_n, first, new, _)) in enumerate(zip(streams, audios, meta)):
            stage_np[off : off + len(new)] = new
            ints_np[0, b], ints_np[1, b], ints_np[2, b], ints_np[3, b] = len(a), first, old_n, off
            ptrs_np[0, b], ptrs_np[1, b] = st.pcm.data_ptr(), st.cache.data_ptr()
            off += len(new)
        ints_np[3, B] = off
        d_stage, d_ints, d_ptrs = self._rows_dev      # (engine-owned: the previous tick's kernels are done - the event above / the pass's read-back)
        d_ints.copy_(ints, non_blocking=True)
        d_ptrs.copy_(ptrs, non_blocking=True)
        d_new = None
        if total:
            d_new = d_stage[:total]
            d_new.copy_(stage[:total], non_blocking=True)
        self.frontend.parakeet_cached_rows(d_ptrs[0, :B], d_ptrs[1, :B], d_ints[0], d_ints[1], d_new, d_ints[3] if total else None, d_ints[2] if total else None,
                                           t_max, out, nf)
        self._rows_event = torch.cuda.Event()          # behind the kernels that read the staging blocks
        self._rows_event.record()
        for st, a, (old_n, first, new, digest) in zip(streams, audios, meta):
            self._stream_account(st, len(a), len(new), first, t_max, digest)

    def encode(self, feats: torch.Tensor, n_frames: torch.Tensor) -> None:
        B, T, _ = feats.shape
        assert feats.dtype == self.dtype and feats.is_contiguous()
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_parakeet_encode(self._h, native.stream_handle(), native.ptr(feats), native.ptr(n_frames), B, T),
                         "s2s_parakeet_encode")
        self._B, self._T = B, encoder_frames(T)

    def encoder_output(self):
        out = torch.empty((self._B, self._T, self.dims.d_model), dtype=torch.float32, device=self.device)
        lens = torch.empty((self._B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_parakeet_read_encoder_output(self._h, native.stream_handle(), native.ptr(out), native.ptr(lens)))
        return out, lens

    def tdt_greedy(self, max_steps: int = 0):
        """-> per session (tokens incl. blanks, durations, frame index of each step)."""
        ms = max_steps if max_steps > 0 else 10 * self._T - 1
        B = self._B
        # engine-owned result blocks (device + pinned twin), grown on demand: no fill kernels, no allocation and ONE synchronisation per pass
        # (entries past a session's step count are never read)
        need = 3 * B * ms + B
        if getattr(self, "_tdt_dev", None) is None or self._tdt_dev.numel() < need:
            size = max(need, 3 * self.max_batch * 10 * 64 + self.max_batch)
            self._tdt_dev = torch.empty((size,), dtype=torch.int32, device=self.device)
            self._tdt_host = torch.empty((size,), dtype=torch.int32, pin_memory=True)
        dev = self._tdt_dev[:need]
        toks, durs, frames, n = dev[: B * ms].view(B, ms), dev[B * ms : 2 * B * ms].view(B, ms), dev[2 * B * ms : 3 * B * ms].view(B, ms), dev[3 * B * ms :]
        with torch.cuda.device(self.device):
            native.check(self._lib.s2s_parakeet_tdt_greedy(self._h, native.stream_handle(), ms, native.ptr(toks), native.ptr(durs),
                                                           native.ptr(frames), native.ptr(n)), "s2s_parakeet_tdt_greedy")
            host = self._tdt_host[:need]
            host.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        h = host.numpy()
        t_h, d_h, f_h, n_h = h[: B * ms].reshape(B, ms), h[B * ms : 2 * B * ms].reshape(B, ms), h[2 * B * ms : 3 * B * ms].reshape(B, ms), h[3 * B * ms :]
        return ([t_h[b, : n_h[b]].tolist() for b in range(B)], [d_h[b, : n_h[b]].tolist() for b in range(B)],
                [f_h[b, : n_h[b]].tolist() for b in range(B)])

    def transcribe_batch(self, audios: Sequence[np.ndarray], options: Sequence[dict]) -> list[dict]:
        """SessionPool entry point: {"ids": non-blank token ids, "starts_s" / "ends_s": start / end time of each token in
        seconds (encoder frame x 80 ms; end = start + the TDT duration, at least one frame)}."""
        max_len = (self.max_feat_frames - 1) * 160
        # A turn longer than the engine's window (max_seconds) is transcribed window by window and stitched (token times shifted by the
        # window start) - the reference transcribes turns of any length (its VAD max_speech_ms defaults to infinity); nothing is dropped.
        pieces, owner, offset = [], [], []
        for i, a in enumerate(audios):
            if len(a) > max_len:
                logger.info("Parakeet: %.1f s turn exceeds the %.1f s window: transcribing it in %d windows", len(a) / 16000.0, max_len / 16000.0,
                            -(-len(a) // max_len))
            for s0 in range(0, max(len(a), 1), max_len):
                seg = a[s0 : s0 + max_len]
                pieces.append(seg if len(seg) >= 1280 else np.pad(seg, (0, 1280 - len(seg))))
                owner.append(i)
                offset.append(s0 / 16000.0)
        blank = self.dims.blank
        out = [{"ids": [], "starts_s": [], "ends_s": []} for _ in audios]
        n_pieces = [owner.count(i) for i in range(len(audios))]
        for b0 in range(0, len(pieces), self.max_batch):
            chunk = pieces[b0 : b0 + self.max_batch]
            # live partials that carry a ProgressiveStream (options["stream"], one-window turns only): their rows come from the stream's device
            # state, everyone else's from the batched front-end call - ONE encoder pass for all of them either way
            streams = [(options[owner[b0 + j]] or {}).get("stream") if n_pieces[owner[b0 + j]] == 1 else None for j in range(len(chunk))]
            if any(s is not None for s in streams):
                T = max(8, max(len(a) for a in chunk) // 160 + 1)
                feats = torch.empty((len(chunk), T, self.dims.n_mels), dtype=self.dtype, device=self.device)
                nf = torch.empty((len(chunk),), dtype=torch.int32, device=self.device)
                plain = [j for j, s in enumerate(streams) if s is None]
                order = plain + [j for j, s in enumerate(streams) if s is not None]
                if plain:
                    pcm, ns = pack_pcm([chunk[j] for j in plain], self.device)
                    self.frontend.parakeet(pcm, ns, T, out_dtype=self.dtype, out=feats[: len(plain)], n_frames=nf[: len(plain)])
                live = order[len(plain):]
                self.stream_features_rows([streams[j] for j in live], [chunk[j] for j in live], [(options[owner[b0 + j]] or {}).get("stream_key") for j in live],
                                          T, feats[len(plain):], nf[len(plain):])
                own_c, off_c = [owner[b0 + j] for j in order], [offset[b0 + j] for j in order]
                owner[b0 : b0 + len(chunk)], offset[b0 : b0 + len(chunk)] = own_c, off_c
            else:
                feats, nf, _ = self.features(chunk)
            self.encode(feats, nf)
            toks, durs, frames = self.tdt_greedy()
            for j, (tk, du, fr) in enumerate(zip(toks, durs, frames)):
                o, t0 = out[owner[b0 + j]], offset[b0 + j]
                keep = [k for k, t in enumerate(tk) if t != blank]
                o["ids"] += [tk[k] for k in keep]
                o["starts_s"] += [t0 + fr[k] * 0.08 for k in keep]
                o["ends_s"] += [t0 + (fr[k] + max(du[k], 1)) * 0.08 for k in keep]
        return out

    def transcribe_ids(self, audios: Sequence[np.ndarray]):
        """-> per session (non-blank token ids, start frame of each token [x 80 ms])."""
        feats, nf, _ = self.features(audios)
        self.encode(feats, nf)
        toks, durs, frames = self.tdt_greedy()
        blank = self.dims.blank
        ids = [[t for t in tk if t != blank] for tk in toks]
        starts = [[f for t, f in zip(tk, fr) if t != blank] for tk, fr in zip(toks, frames)]
        return ids, starts
