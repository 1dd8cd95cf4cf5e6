"""Codec-decoder timing at the benchmark dims (development aid): B sessions x 8-frame streamed chunks, as the TTS lanes run it.

    python tools/bench_codec.py [B] [chunks]        wall ms per chunk; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dims = Qwen3TTSDims.tts_1_7b()
eng = TTSEngine.from_random(dims, torch.bfloat16, max_batch=B, max_prompt=32, max_frames=8 * n + 8, max_codec_frames=33, device="cuda:0", seed=0)
g = torch.Generator().manual_seed(0)
codes = torch.randint(0, dims.codebook, (B, 8, dims.n_groups), generator=g, dtype=torch.int32).to("cuda:0")
for rep in range(2):
    eng.codec_stream_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        wav = eng.codec_stream_decode(codes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"codec stream: B={B}, 8-frame chunks: {dt * 1e3:.2f} ms per chunk ({B * 8 * 0.08 / dt:.0f} x real time), wav {tuple(wav.shape)}")
