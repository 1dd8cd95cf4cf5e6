"""Loading a Qwen3-TTS checkpoint directory (``--tts mi355x-qwen3 --qwen3_tts_model_name <dir>``).

What is KNOWN offline is the architecture of the sibling modules in transformers (qwen3 / qwen3_omni_moe), whose parameter
names the engine's converter (tts.convert_tts_state_dict) and the oracle consume.  The real Qwen3-TTS release (qwen_tts /
faster-qwen3-tts, absent here) may name tensors and config keys differently, so everything checkpoint-specific sits behind
data, not code:

* ``config.json`` -> ``Qwen3TTSDims``: ``read_dims`` accepts (a) this repo's export schema (``"format": "s2s-mi355x"``, the
  dataclass fields) and (b) a nested HF-style config (``talker_config`` / ``code_predictor_config`` / ``speech_tokenizer`` or
  ``code2wav_config`` blocks with the usual ``hidden_size`` / ``num_hidden_layers`` / ... keys) - UNVERIFIED against the real file;
* tensor names: ``NAME_RULES`` (regex -> oracle name) rewrites checkpoint names; a ``name_map.json`` next to the weights
  (``{"<checkpoint name regex>": "<oracle name template>"}``) overrides / extends it without touching code;
* the prompt (text tokens, speaker / language / instruct rows, codec control tokens): ``TTSPromptLayout`` (``prompt_layout.json``).
A load fails loudly with the list of missing / unexpected tensors - there is no silent fallback to random weights.
"""

from __future__ import annotations

import json
import os
import re
from dataclasses import asdict, dataclass, field, fields
from typing import Any, Mapping

import numpy as np
import torch

from .synth_weights import Qwen3TTSDims

# checkpoint-name regex -> oracle-name template (first match wins, re-applied until the name stops changing); identity for names already in oracle form
NAME_RULES: list[tuple[str, str]] = [
    (r"^(talker\.model\.(?:layers\.\d+\..*|norm\.weight|codec_embedding\.weight))$", r"\1"),
    (r"^(talker\.codec_head\.weight)$", r"\1"),
    (r"^talker\.code_predictor\.(.*)$", r"code_predictor.\1"),                       # predictor nested under the talker
    (r"^code_predictor\.small_to_mtp_projection\.(weight|bias)$", r"code_predictor.in_proj.\1"),
    (r"^(code_predictor\..*)$", r"\1"),
    (r"^(?:speech_tokenizer\.|tokenizer\.)?decoder\.(.*)$", r"code2wav.\1"),          # codec decoder shipped as the speech tokenizer's decoder
    (r"^(code2wav\..*)$", r"\1"),
]


@dataclass
class TTSPromptLayout:
    """How one utterance's prefill rows are assembled: row i = text_side[i] + codec_side[i] (either may be absent = zero row).
    ``text``: where the text tokens go; specials are token ids of the text vocabulary / codec vocabulary.  The defaults describe the
    synthetic byte-level front-end; a real checkpoint ships its own ``prompt_layout.json``."""

    text_bos: int | None = 256
    text_eos: int | None = 257
    text_pad: int | None = None                 # text-side row added to every codec-only position
    codec_prefix: list[int] = field(default_factory=list)       # codec control tokens (think / language / speaker / pad ...) in front
    codec_bos: int | None = None
    speaker_ids: dict[str, int] = field(default_factory=dict)   # speaker name -> codec-side token id
    language_ids: dict[str, int] = field(default_factory=dict)  # language code -> codec-side token id
    max_text_tokens: int = 126
    # which of the three released model kinds the checkpoint is (reference `tts_model_type`, TTS/qwen3_tts_handler.py:574-580):
    # "custom_voice" (speaker table), "voice_design" (instruct text describes the voice), "base" (voice clone from a reference)
    model_type: str | None = None
    instruct_bos: int | None = None             # text-side tokens wrapping the instruct text (None: the instruct tokens alone)
    instruct_eos: int | None = None

    @classmethod
    def load(cls, path: str) -> "TTSPromptLayout":
        f = os.path.join(path, "prompt_layout.json")
        if not os.path.exists(f):
            return cls()
        with open(f) as fh:
            d = json.load(fh)
        known = {x.name for x in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})


def read_dims(cfg: Mapping[str, Any]) -> Qwen3TTSDims:
    if cfg.get("format") == "s2s-mi355x":
        known = {f.name for f in fields(Qwen3TTSDims)}
        d = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg["dims"].items() if k in known}
        return Qwen3TTSDims(**d)
    t = cfg.get("talker_config") or cfg.get("talker") or cfg
    p = t.get("code_predictor_config") or cfg.get("code_predictor_config") or {}
    c = cfg.get("code2wav_config") or cfg.get("speech_tokenizer_config") or cfg.get("speech_tokenizer") or {}

    def g(d, *names, default=None):
        for n in names:
            if n in d and d[n] is not None:
                return d[n]
        if default is None:
            raise KeyError(f"config.json: none of {names} found")
        return default

    hid = g(t, "hidden_size")
    heads = g(t, "num_attention_heads")
    base = Qwen3TTSDims()
    return Qwen3TTSDims(
        t_hidden=hid, t_layers=g(t, "num_hidden_layers"), t_heads=heads, t_kv_heads=g(t, "num_key_value_heads", default=heads),
        t_head_dim=g(t, "head_dim", default=hid // heads), t_ffn=g(t, "intermediate_size"), t_theta=float(g(t, "rope_theta", default=base.t_theta)),
        codec_vocab=g(t, "vocab_size", "codec_vocab_size"), codec_eos=g(t, "codec_eos_token_id", "eos_token_id", default=base.codec_eos),
        p_hidden=g(p, "hidden_size", default=base.p_hidden), p_layers=g(p, "num_hidden_layers", default=base.p_layers),
        p_heads=g(p, "num_attention_heads", default=base.p_heads), p_kv_heads=g(p, "num_key_value_heads", default=base.p_kv_heads),
        p_head_dim=g(p, "head_dim", default=base.p_head_dim), p_ffn=g(p, "intermediate_size", default=base.p_ffn),
        p_theta=float(g(p, "rope_theta", default=base.p_theta)), n_groups=g(t, "num_code_groups", default=g(p, "num_code_groups", default=base.n_groups)),
        codebook=g(p, "vocab_size", "codebook_size", default=base.codebook), eps=float(g(t, "rms_norm_eps", default=base.eps)),
        c_hidden=g(c, "hidden_size", default=base.c_hidden), c_layers=g(c, "num_hidden_layers", default=base.c_layers),
        c_heads=g(c, "num_attention_heads", default=base.c_heads), c_ffn=g(c, "intermediate_size", default=base.c_ffn),
        c_window=g(c, "sliding_window", default=base.c_window), c_theta=float(g(c, "rope_theta", default=base.c_theta)),
        c_eps=float(g(c, "rms_norm_eps", default=base.c_eps)), c_decoder_dim=g(c, "decoder_dim", default=base.c_decoder_dim),
        upsample_rates=tuple(g(c, "upsample_rates", default=list(base.upsample_rates))),
        upsampling_ratios=tuple(g(c, "upsampling_ratios", default=list(base.upsampling_ratios))))


def required_names(dims: Qwen3TTSDims) -> list[str]:
    """Every tensor the converter reads (= the names of the synthetic state dict at these dims)."""
    from .synth_weights import qwen3_tts_state_dict

    small = Qwen3TTSDims(**{**asdict(dims), "t_hidden": 32, "t_ffn": 32, "t_head_dim": 8, "p_hidden": 32 if dims.p_hidden == dims.t_hidden else 16, "p_ffn": 32,
                            "p_head_dim": 8, "c_hidden": 32, "c_ffn": 32, "c_decoder_dim": 16 << len(dims.upsample_rates), "codec_vocab": 8, "codebook": 4,
                            "t_heads": 2, "t_kv_heads": 1, "p_heads": 2, "p_kv_heads": 1, "c_heads": 2, "codec_eos": 7})
    return sorted(qwen3_tts_state_dict(small, seed=0).keys())


def remap_names(sd: Mapping[str, torch.Tensor], extra_rules: Mapping[str, str] | None = None) -> dict[str, torch.Tensor]:
    rules = [(re.compile(k), v) for k, v in (extra_rules or {}).items()] + [(re.compile(k), v) for k, v in NAME_RULES]
    out: dict[str, torch.Tensor] = {}
    origin: dict[str, str] = {}
    for name, t in sd.items():
        cur = name
        for _ in range(4):                      # rules chain (e.g. un-nest the predictor, then rename its in-projection)
            for rx, tmpl in rules:
                if rx.match(cur):
                    new = rx.sub(tmpl, cur)
                    break
            else:
                new = cur
            if new == cur:
                break
            cur = new
        if cur in out:                          # two checkpoint tensors landing on one engine tensor: never pick one silently
            raise KeyError(f"name mapping collision: checkpoint tensors {origin[cur]!r} and {name!r} both map to {cur!r}; fix name_map.json")
        out[cur] = t
        origin[cur] = name
    return out


def load_qwen3_tts_checkpoint(path: str):
    """-> (dims, float32 state dict in the oracle's naming, TTSPromptLayout, text front-end tensors | None)."""
    from .checkpoints import load_safetensors_dir

    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    dims = read_dims(cfg)
    extra = {}
    nm = os.path.join(path, "name_map.json")
    if os.path.exists(nm):
        with open(nm) as f:
            extra = json.load(f)
    sd = remap_names(load_safetensors_dir(path), extra)
    need = required_names(dims)
    missing = [n for n in need if n not in sd]
    if missing:
        raise KeyError(f"{path}: {len(missing)} tensors the engine needs are missing after name mapping (first: {missing[:6]}); "
                       "add rules to name_map.json (see speech-to-speech_amd/qwen3_tts_checkpoint.py)")
    _check_shapes_cheap(sd, dims, need)
    text = {k: sd[k] for k in ("text.embedding.weight", "text.proj.0.weight", "text.proj.0.bias", "text.proj.1.weight", "text.proj.1.bias", "text.pad_embed")
            if k in sd}
    if cfg.get("format") != "s2s-mi355x" and not os.path.exists(os.path.join(path, "prompt_layout.json")):
        # a foreign checkpoint's prompt (control tokens, speaker / language rows) cannot be guessed: wrong rows give fluent-sounding wrong audio
        raise FileNotFoundError(f"{path}: prompt_layout.json is required for a checkpoint that is not in this repo's export format "
                                "(see TTSPromptLayout in speech-to-speech_amd/qwen3_tts_checkpoint.py)")
    return dims, {k: sd[k] for k in need}, TTSPromptLayout.load(path), text or None


def _expected_shapes(dims: Qwen3TTSDims) -> dict[str, tuple]:
    """name -> shape of every tensor the converter reads, from the dims alone (no allocation)."""
    E, P, H, G, CB = dims.t_hidden, dims.p_hidden, dims.c_hidden, dims.n_groups, dims.codebook
    out: dict[str, tuple] = {}

    def stack(prefix, hidden, layers, heads, kv, hd, ffn, qk_norm, layer_scale):
        for l in range(layers):
            p = f"{prefix}layers.{l}."
            out[p + "self_attn.q_proj.weight"] = (heads * hd, hidden)
            out[p + "self_attn.k_proj.weight"] = (kv * hd, hidden)
            out[p + "self_attn.v_proj.weight"] = (kv * hd, hidden)
            out[p + "self_attn.o_proj.weight"] = (hidden, heads * hd)
            if qk_norm:
                out[p + "self_attn.q_norm.weight"] = (hd,)
                out[p + "self_attn.k_norm.weight"] = (hd,)
            out[p + "mlp.gate_proj.weight"] = (ffn, hidden)
            out[p + "mlp.up_proj.weight"] = (ffn, hidden)
            out[p + "mlp.down_proj.weight"] = (hidden, ffn)
            out[p + "input_layernorm.weight"] = (hidden,)
            out[p + "post_attention_layernorm.weight"] = (hidden,)
            if layer_scale:
                out[p + "self_attn_layer_scale.scale"] = (hidden,)
                out[p + "mlp_layer_scale.scale"] = (hidden,)

    stack("talker.model.", E, dims.t_layers, dims.t_heads, dims.t_kv_heads, dims.t_head_dim, dims.t_ffn, True, False)
    out["talker.model.norm.weight"] = (E,)
    out["talker.model.codec_embedding.weight"] = (dims.codec_vocab, E)
    out["talker.codec_head.weight"] = (dims.codec_vocab, E)
    stack("code_predictor.model.", P, dims.p_layers, dims.p_heads, dims.p_kv_heads, dims.p_head_dim, dims.p_ffn, True, False)
    out["code_predictor.model.norm.weight"] = (P,)
    if P != E:
        out["code_predictor.in_proj.weight"] = (P, E)
        out["code_predictor.in_proj.bias"] = (P,)
    for g in range(G - 1):
        out[f"code_predictor.model.codec_embedding.{g}.weight"] = (CB, E)
        out[f"code_predictor.lm_head.{g}.weight"] = (CB, P)
    out["code2wav.code_embedding.weight"] = (CB * G, H)
    stack("code2wav.pre_transformer.", H, dims.c_layers, dims.c_heads, dims.c_heads, H // dims.c_heads, dims.c_ffn, False, True)
    out["code2wav.pre_transformer.norm.weight"] = (H,)
    return out


def _check_shapes_cheap(sd: Mapping[str, torch.Tensor], dims: Qwen3TTSDims, need: list[str]) -> None:
    want = _expected_shapes(dims)
    bad = [(k, tuple(sd[k].shape), want[k]) for k in need if k in want and k in sd and tuple(sd[k].shape) != want[k]]
    if bad:
        k, got, shp = bad[0]
        raise ValueError(f"{len(bad)} checkpoint tensors do not match config.json (first: {k} has shape {got}, the config implies {shp})")


def save_synthetic_checkpoint(path: str, dims: Qwen3TTSDims, sd: Mapping[str, torch.Tensor], text_table: np.ndarray | None = None,
                              pad_embed: torch.Tensor | None = None) -> None:
    """Write a checkpoint directory in this repo's export schema (tests / examples): config.json + model.safetensors."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    d = {k: (list(v) if isinstance(v, tuple) else v) for k, v in asdict(dims).items()}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump({"format": "s2s-mi355x", "dims": d}, f)
    out = {k: v.contiguous() for k, v in sd.items()}
    if text_table is not None:
        out["text.embedding.weight"] = torch.from_numpy(np.ascontiguousarray(text_table, dtype=np.float32))
    if pad_embed is not None:
        out["text.pad_embed"] = pad_embed.contiguous()
    save_file(out, os.path.join(path, "model.safetensors"))


class TableTextFrontend:
    """Text side of the prompt for a checkpoint: token ids (tokenizer.json when present, UTF-8 bytes otherwise) -> rows of the text
    embedding table, through the 2-layer projection when the checkpoint has one (text width != talker width)."""

    def __init__(self, table: np.ndarray, layout: TTSPromptLayout, tokenizer: Any = None, proj: list | None = None,
                 codec_table: np.ndarray | None = None, group_tables: list | None = None):
        self.table, self.layout, self.tokenizer, self.proj = table.astype(np.float32), layout, tokenizer, proj
        self.codec_table = None if codec_table is None else np.asarray(codec_table, dtype=np.float32)
        self.group_tables = [np.asarray(t, dtype=np.float32) for t in (group_tables or [])]      # code predictor tables, groups 1 .. G-1
        lay = layout
        if (lay.codec_prefix or lay.codec_bos is not None or lay.speaker_ids or lay.language_ids) and self.codec_table is None:
            # the layout asks for codec-side rows: without the codec embedding table they would silently be missing from the prefill
            raise ValueError("prompt_layout.json sets codec-side fields (codec_prefix / codec_bos / speaker_ids / language_ids) but the front-end "
                             "was built without the talker's codec embedding table")

    def _text_rows(self, ids: list[int]) -> np.ndarray:
        rows = self.table[np.asarray(ids, dtype=np.int64)]
        if self.proj:
            (w0, b0), (w1, b1) = self.proj
            h = rows @ w0.T + b0
            rows = (h / (1.0 + np.exp(-h))) @ w1.T + b1              # SiLU MLP
        return rows.astype(np.float32)

    def _ids(self, text: str) -> list[int]:
        if self.tokenizer is not None:
            return list(self.tokenizer.encode(text))
        return list(text.encode("utf-8"))

    def code_rows(self, codes: Any) -> np.ndarray:
        """Reference codec frames [F, G] -> talker input rows: Σ_g table_g[codes[f, g]] - exactly what the frame loop feeds back for a frame
        it generated itself (oracle/qwen3_tts.py talker_generate: sum of the G code embeddings), so in-context reference frames look like
        the model's own history.  Group 0 = the talker's codec embedding, groups 1 .. G-1 = the code predictor's tables."""
        if self.codec_table is None or not self.group_tables:
            raise ValueError("reference codec frames need the codec embedding tables (front-end built without them)")
        c = np.asarray(codes, dtype=np.int64)
        if c.ndim != 2 or c.shape[1] != 1 + len(self.group_tables):
            raise ValueError(f"reference codec frames must be [frames, {1 + len(self.group_tables)}], got {c.shape}")
        if c.min() < 0 or c[:, 0].max() >= self.codec_table.shape[0] or any(c[:, g + 1].max() >= t.shape[0] for g, t in enumerate(self.group_tables)):
            raise ValueError("reference codec frames hold ids outside the codebooks")
        rows = self.codec_table[c[:, 0]].copy()
        for g, t in enumerate(self.group_tables):
            rows += t[c[:, g + 1]]
        return rows.astype(np.float32)

    def embed(self, text: str, max_tokens: int, speaker: str | None = None, language: str | None = None, instruct: str | None = None,
              speaker_row: np.ndarray | None = None, ref_text: str | None = None, ref_rows: np.ndarray | None = None, streaming: bool = False):
        """Prefill rows, row i = text_side[i] + codec_side[i] (TTSPromptLayout):
            [codec control rows: codec_prefix, then the language token and the speaker token when given - or `speaker_row`, an explicit
             speaker embedding (voice clone: the x-vector projected to the talker width) in the speaker token's place; text side = text_pad]
            [instruct rows (custom voice style / voice design): instruct_bos, the instruct tokens, instruct_eos; no codec side]
            [text rows: text_bos, (the reference transcript,) the text tokens, text_eos; no codec side]
            [codec_bos row; text side = text_pad]
            [reference codec frames (voice clone in-context rows, `code_rows`); text side = text_pad]
        An unknown speaker / language name is an error (never a silent default voice).  The control-row order and the in-context
        placement are this repo's assumption about the absent qwen_tts prompt builder.
        streaming = True (reference non_streaming_mode = False): -> (prefill rows, trailing rows) - the text block of the prefill ends with the
        FIRST text token, the remaining text rows and text_eos are the trailing rows the frame loop consumes one per frame (the split of the
        sibling's talker, modeling_qwen3_omni_moe.py:3744-3801: "First text" in the prefill, `trailing_text_hidden` = the rest + eos)."""
        lay = self.layout
        ctrl = list(lay.codec_prefix)
        if language and language != "auto":
            if lay.language_ids:
                if language not in lay.language_ids:
                    raise ValueError(f"language {language!r} is not in prompt_layout.json's language_ids {sorted(lay.language_ids)}")
                ctrl.append(lay.language_ids[language])
        if speaker and speaker_row is None:
            if speaker not in lay.speaker_ids:
                raise ValueError(f"speaker {speaker!r} is not in prompt_layout.json's speaker_ids {sorted(lay.speaker_ids)[:16]}")
            ctrl.append(lay.speaker_ids[speaker])
        tail = [lay.codec_bos] if lay.codec_bos is not None else []
        ids = (self._ids(ref_text.strip() + " ") if ref_text else []) + self._ids(text)
        ins = []
        if instruct:
            ins = ([lay.instruct_bos] if lay.instruct_bos is not None else []) + self._ids(instruct) + ([lay.instruct_eos] if lay.instruct_eos is not None else [])
        n_ref = 0 if ref_rows is None else len(ref_rows)
        n_fixed = (lay.text_bos is not None) + (lay.text_eos is not None) + len(ctrl) + len(tail) + len(ins) + (speaker_row is not None) + n_ref
        budget = min(lay.max_text_tokens, max_tokens - n_fixed)
        if len(ids) > budget:
            from .text import PromptTooLong

            raise PromptTooLong(f"TTS input of {len(ids)} text tokens exceeds the prompt budget of {budget}; split the text into shorter sentences")
        n_text = len(ids)
        ids = ([lay.text_bos] if lay.text_bos is not None else []) + ids + ([lay.text_eos] if lay.text_eos is not None else [])
        rows = self._text_rows(ins + ids)
        trailing = None
        if streaming:
            k = len(ins) + (lay.text_bos is not None) + min(1, n_text)
            rows, trailing = rows[:k], rows[k:].astype(np.float32)
        if not ctrl and not tail and speaker_row is None and not n_ref:
            return (rows, trailing) if streaming else rows
        E = rows.shape[1]
        pad = self._text_rows([lay.text_pad])[0] if lay.text_pad is not None else np.zeros(E, np.float32)
        if (ctrl or tail) and self.codec_table.shape[1] != E:
            raise ValueError(f"codec embedding width {self.codec_table.shape[1]} != text row width {E}")
        head = self.codec_table[np.asarray(ctrl, dtype=np.int64)] + pad if ctrl else np.zeros((0, E), np.float32)
        if speaker_row is not None:
            sr = np.asarray(speaker_row, np.float32).reshape(-1)
            if sr.shape[0] != E:
                raise ValueError(f"speaker embedding has {sr.shape[0]} values, the talker rows are {E} wide")
            head = np.concatenate([head, (sr + pad)[None]])
        foot = self.codec_table[np.asarray(tail, dtype=np.int64)] + pad if tail else np.zeros((0, E), np.float32)
        ref = (np.asarray(ref_rows, np.float32).reshape(-1, E) + pad) if n_ref else np.zeros((0, E), np.float32)
        full = np.concatenate([head, rows, foot, ref]).astype(np.float32)
        return (full, trailing) if streaming else full
