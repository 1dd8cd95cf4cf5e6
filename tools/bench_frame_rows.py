"""Qwen3-TTS-1.7B dims (random weights, bf16): ms per codec frame of the captured frame loop (talker step + 15-pass code predictor) against the
number of live rows - how much a row costs once the launch-bound part is paid.  python tools/bench_frame_rows.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

dims = Qwen3TTSDims.tts_1_7b()
eng = TTSEngine.from_random(dims, torch.bfloat16, max_batch=512, max_prompt=128, max_frames=128, max_codec_frames=8, device="cuda:0", seed=0)
g = np.random.default_rng(0)
for B in (64, 128, 256, 384, 512):
    prompts = [TTSPrompt(g.standard_normal((100 + (i % 28), dims.t_hidden)).astype(np.float32)) for i in range(B)]
    eng.prefill(prompts)
    eng.generate(8, min_total_frames=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(4):
        eng.generate(8, min_total_frames=128)
        n += 8
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{B:4d} rows: {dt * 1e3:6.2f} ms per frame = {dt / B * 1e6:6.2f} us per row-frame, {B / dt / 12.5:7.0f} x real time")
    eng.release(0, B)
