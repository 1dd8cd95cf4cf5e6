"""Where engines get their weights.  ``synthetic:<name>[:seed]`` builds the seeded random checkpoint at the named
dimensions (the only option offline); a directory with ``*.safetensors`` is read as a transformers-named state dict."""

from __future__ import annotations

import glob
import os

import torch

from . import synth_weights as SW

WHISPER_DIMS = {"test-tiny": SW.WhisperDims.test_tiny, "tiny.en": SW.WhisperDims.tiny_en, "large-v3": SW.WhisperDims.large_v3,
                "distil-large-v3": SW.WhisperDims.distil_large_v3}
PARAKEET_DIMS = {"test-tiny": SW.ParakeetDims.test_tiny, "tdt-0.6b": SW.ParakeetDims.tdt_0_6b}
TTS_DIMS = {"test-tiny": SW.Qwen3TTSDims.test_tiny, "1.7b": SW.Qwen3TTSDims.tts_1_7b}


def parse_model_name(name: str):
    """-> ("synthetic", dims_key, seed) | ("dir", path, None)"""
    if name.startswith("synthetic:"):
        parts = name.split(":")
        return "synthetic", parts[1], int(parts[2]) if len(parts) > 2 else 0
    if os.path.isdir(name):
        return "dir", name, None
    raise FileNotFoundError(
        f"checkpoint {name!r} is not a local directory and this build has no network access; pass a directory holding "
        f"*.safetensors or 'synthetic:<dims>' (seeded random weights at the true dimensions)")


def load_safetensors_dir(path: str) -> dict[str, torch.Tensor]:
    from safetensors.torch import load_file

    sd: dict[str, torch.Tensor] = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for f in files:
        sd.update(load_file(f))
    return {k: v.float() for k, v in sd.items()}
