"""ctypes binding of ``libs2s_mi355x.so`` (the C ABI declared in ``include/s2s_mi355x.h``).

The product path has no CPU fallback: if the shared library is missing, or a call
returns a non-zero status, this module raises.  PyTorch-ROCm tensors are used only as
device-memory containers; ``ptr(t)`` turns one into the raw pointer the ABI takes.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

from . import PACKAGE_DIR, REPO_ROOT

LIB_PATH = os.path.join(PACKAGE_DIR, "lib", "libs2s_mi355x.so")
HEADER_PATH = os.path.join(REPO_ROOT, "include", "s2s_mi355x.h")
DEV_HEADER_PATH = os.path.join(REPO_ROOT, "include", "s2s_mi355x_dev.h")

S2S_F32, S2S_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU, ACT_RELU = 0, 1, 2, 3
FRONTEND_WHISPER, FRONTEND_PARAKEET = 0, 1


class WhisperConfigStruct(C.Structure):
    """Mirror of ``s2s_whisper_config``."""

    _fields_ = [(n, C.c_int32) for n in ("n_mels", "d_model", "n_heads", "enc_layers", "dec_layers", "ffn", "vocab",
                                          "max_target", "dtype")]


class ParakeetConfigStruct(C.Structure):
    """Mirror of ``s2s_parakeet_config``."""

    _fields_ = [(n, C.c_int32) for n in ("n_mels", "d_model", "n_heads", "layers", "ffn", "sub_channels", "conv_kernel",
                                          "vocab", "dec_hidden", "dec_layers", "n_durations", "max_feat_frames", "dtype")]


class TTSConfigStruct(C.Structure):
    """Mirror of ``s2s_tts_config``."""

    _fields_ = ([(n, C.c_int32) for n in ("t_hidden", "t_layers", "t_heads", "t_kv_heads", "t_head_dim", "t_ffn", "codec_vocab",
                                           "codec_eos", "p_hidden", "p_layers", "p_heads", "p_kv_heads", "p_head_dim", "p_ffn",
                                           "n_groups", "codebook", "c_hidden", "c_layers", "c_heads", "c_ffn", "c_window",
                                           "c_decoder_dim", "n_dec_blocks")]
                + [("up_rate", C.c_int32 * 4), ("up_ratio", C.c_int32 * 2)]
                + [(n, C.c_int32) for n in ("total_upsample", "max_prompt", "max_frames", "max_pos", "max_codec_frames")]
                + [("eps", C.c_float), ("c_eps", C.c_float), ("dtype", C.c_int32)])


class NativeError(RuntimeError):
    """A libs2s_mi355x call failed (status code + the library's message)."""


_lib = None
_lock = threading.Lock()

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (restype is int32 status unless listed in _RESTYPES)
_SIGNATURES = {
    "s2s_last_error": [],
    "s2s_abi_version": [],
    "s2s_device_count": [],
    "s2s_graphs_enable": [_i32],
    "s2s_frontend_create": [_i32, _i32, _vp, _i32, C.POINTER(_vp)],
    "s2s_frontend_destroy": [_vp],
    "s2s_logmel_whisper": [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32],
    "s2s_logmel_parakeet": [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _i32, _vp],
    "s2s_logmel_parakeet_cached": [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp],
    "s2s_logmel_parakeet_cached_rows": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "s2s_op_gemm": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32],
    "s2s_op_gemm_ex": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32],
    "s2s_op_gemm_dec": [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _i32, _vp, _i64, _i32, _vp, _i64,
                        _vp, _i64, _vp, _i32, _i32, _i32],
    "s2s_op_gemm_dec_parts": [_i32, _i32, _i32, _i32, _i32],
    "s2s_op_gemm_dec_ln": [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _i32, _vp, _i64, _i32, _i32, _vp],
    "s2s_op_rows_prep_pairs": [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i64],
    "s2s_op_rows_prep": [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32],
    "s2s_op_to_frag": [_vp, _vp, _i64, _vp, _vp, _i32, _i32],
    "s2s_op_layernorm": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32],
    "s2s_op_rmsnorm": [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32],
    "s2s_logmel_whisper_tm": [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i32],
    "s2s_logmel_whisper_window": [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i32],
    "s2s_op_conv1d_tm": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "s2s_op_conv1d_causal_tm": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "s2s_op_attention": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32,
                         _i32, _i32, _i32, _vp],
    "s2s_whisper_create": [_vp, _i32, C.POINTER(_vp)],
    "s2s_whisper_destroy": [_vp],
    "s2s_whisper_set_tensor": [_vp, C.c_char_p, _vp, _i64],
    "s2s_whisper_finalize": [_vp],
    "s2s_whisper_workspace_bytes": [_vp],
    "s2s_whisper_encode": [_vp, _vp, _vp, _i32],
    "s2s_whisper_read_encoder_output": [_vp, _vp, _vp, _i32],
    "s2s_parakeet_create": [_vp, _i32, C.POINTER(_vp)],
    "s2s_parakeet_destroy": [_vp],
    "s2s_parakeet_max_encoder_frames": [_vp],
    "s2s_parakeet_set_tensor": [_vp, C.c_char_p, _vp, _i64],
    "s2s_parakeet_finalize": [_vp],
    "s2s_parakeet_workspace_bytes": [_vp],
    "s2s_parakeet_encode": [_vp, _vp, _vp, _vp, _i32, _i32],
    "s2s_parakeet_read_encoder_output": [_vp, _vp, _vp, _vp],
    "s2s_parakeet_tdt_greedy": [_vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "s2s_tts_create": [_vp, _i32, C.POINTER(_vp)],
    "s2s_tts_destroy": [_vp],
    "s2s_tts_set_tensor": [_vp, C.c_char_p, _vp, _i64],
    "s2s_tts_finalize": [_vp],
    "s2s_tts_workspace_bytes": [_vp],
    "s2s_tts_prefill": [_vp, _vp, _vp, _vp, _i32, _i32],
    "s2s_tts_generate": [_vp, _vp, _i32, _i32, _vp, _vp, C.POINTER(_i32)],
    "s2s_tts_admit": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32],
    "s2s_tts_step": [_vp, _vp, _i32, _i32],
    "s2s_tts_read_codes": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "s2s_tts_release": [_vp, _vp, _i32, _i32],
    "s2s_tts_migrate": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "s2s_tts_read_x": [_vp, _vp, _i32, _i32, _vp],
    "s2s_tts_read_codes_rows": [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp],
    "s2s_codec_decode_rows_at": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32],
    "s2s_op_sample_tokens": [_vp, _vp, _i32, _i32, _f32, _i32, _f32, _f32, C.c_uint64, _vp, _i32, _vp],
    "s2s_tts_set_sampling": [_vp, _i32, _f32, _i32, _f32, _f32, C.c_uint64],
    "s2s_tts_set_trailing": [_vp, _vp, _i32, _i32, _vp, _vp, _i32],
    "s2s_tts_generate_forced": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp],
    "s2s_codec_decode": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32],
    "s2s_codec_samples": [_vp, _i32],
    "s2s_codec_stream_reset": [_vp],
    "s2s_codec_stream_decode": [_vp, _vp, _vp, _i32, _i32, _vp, _i32],
    "s2s_codec_decode_rows": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32],
    "s2s_resample_24k16k_int16": [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _i32],
    "s2s_resample_poly_int16": [_vp, _vp, _i32, _i64, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _i32],
    "s2s_whisper_decode_greedy": [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "s2s_whisper_slots_config": [_vp, _vp, _vp, _vp, _vp, _i32],
    "s2s_whisper_encode_at": [_vp, _vp, _vp, _i32, _i32],
    "s2s_whisper_slots_admit": [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32],
    "s2s_whisper_slots_step": [_vp, _vp, _i32, _i32],
    "s2s_whisper_slots_collect": [_vp, _vp, _i32, _vp],
    "s2s_whisper_slots_release": [_vp, _vp, _i32, _i32],
    "s2s_whisper_slots_pack_stride": [_vp],
}
# development / measurement hooks (include/s2s_mi355x_dev.h): bench.py's kernel timers, tools/, the parity sweeps of tests/
_DEV_SIGNATURES = {
    "s2s_prof_enable": [C.c_uint32],
    "s2s_prof_event_overhead_us": [],
    "s2s_prof_collect": [_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)],
    "s2s_op_gemm_dec_tune": [_i32, _i32, _i32],
    "s2s_op_gemm_dec_debug": [_vp],
    "s2s_op_gemm_tile_debug": [_vp],
    "s2s_graph_bench_begin": [C.POINTER(_vp)],
    "s2s_graph_bench_end": [_i32, _i32, C.POINTER(C.c_double)],
    "s2s_op_gemm_decw_tune": [_i32, _i32, _i32],
    "s2s_guard_mode": [],
    "s2s_guard_check": [C.c_char_p, _i32],
    "s2s_op_attn_step": [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32],
    "s2s_op_attn_step_tune": [_i32],
    "s2s_op_conv7_tune": [_i32],
}
_RESTYPES = {"s2s_last_error": C.c_char_p, "s2s_whisper_workspace_bytes": C.c_int64,
             "s2s_parakeet_workspace_bytes": C.c_int64, "s2s_tts_workspace_bytes": C.c_int64,
             "s2s_codec_samples": C.c_int64, "s2s_prof_event_overhead_us": C.c_double}


def load():
    """Load (once) and return the ctypes library; raises NativeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C speech-to-speech_amd/csrc`). There is no CPU fallback for the MI355X path."
            )
        try:
            import torch  # noqa: F401  (load torch's HIP runtime first so both share one libamdhip64)
        except Exception:  # pragma: no cover - torch is optional for symbol checks
            pass
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in list(_SIGNATURES.items()) + list(_DEV_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i32)
        _lib = lib
        return lib


def guard_check() -> tuple[int, str]:
    """S2S_GUARD development mode (include/s2s_mi355x_dev.h): -> (number of poisoned gaps a kernel wrote into, report text)."""
    lib = load()
    buf = C.create_string_buffer(16384)
    n = int(lib.s2s_guard_check(buf, len(buf)))
    if n < 0:
        check(n, "s2s_guard_check")
    return n, buf.value.decode(errors="replace")


def declared_symbols(header: str | None = None) -> list[str]:
    """Every function name the public header (or `header`) declares (used by the export test)."""
    import re

    with open(header or HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s2s_[a-z0-9_]+)\s*\(", text)))


PROF_GEMM_TILE, PROF_GEMM_SKINNY, PROF_ATTN_MFMA, PROF_ATTN_ROWS, PROF_LOGMEL = 0, 1, 2, 3, 4


def prof_enable(*categories: int) -> None:
    mask = 0
    for c in categories:
        mask |= 1 << c
    check(load().s2s_prof_enable(mask), "s2s_prof_enable")


def graphs_enable(on: bool) -> None:
    """Replay captured hipGraphs in the autoregressive loops (default) or launch every kernel on the stream."""
    check(load().s2s_graphs_enable(1 if on else 0), "s2s_graphs_enable")


def prof_collect(category: int) -> dict:
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), _i64()
    check(load().s2s_prof_collect(category, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "s2s_prof_collect")
    return {"ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value}


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().s2s_last_error()
        raise NativeError(f"{what or 'libs2s_mi355x'} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Raw device/host pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return 0
    return int(t.data_ptr())


def stream_handle() -> int:
    """hipStream_t of torch's current stream on the current device."""
    import torch

    return int(torch.cuda.current_stream().cuda_stream)


def to_frag(w, gamma=None):
    """Row-major bf16 [rows][K] device tensor -> the fragment-major copy the decode projections read (csrc/gemm_dec.hip): rows
    padded to a multiple of 16 with zeros, optional norm weight ``gamma`` [K] (float32) folded into the columns."""
    import torch

    assert w.dim() == 2 and w.dtype == torch.bfloat16 and w.is_cuda and w.stride(1) == 1
    rows, K = w.shape
    out = torch.empty(((rows + 15) // 16 * 16) * K, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        check(load().s2s_op_to_frag(stream_handle(), ptr(w), w.stride(0), ptr(gamma), ptr(out), rows, K), "s2s_op_to_frag")
    return out


def dtype_code(dt) -> int:
    import torch

    if dt == torch.float32:
        return S2S_F32
    if dt == torch.bfloat16:
        return S2S_BF16
    raise NativeError(f"unsupported dtype {dt}")
