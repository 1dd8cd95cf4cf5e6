"""TDT greedy loop of Parakeet-TDT-0.6B dims in isolation (random weights): one 30 s utterance (375 encoder frames) and a batch of 32 windows of
5 s, wall time of s2s_parakeet_tdt_greedy after an encoder pass.  A/B: S2S_TDT_FUSED=0 python tools/bench_tdt.py (the 14-launch step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd.parakeet import ParakeetEngine
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

pd = ParakeetDims.tdt_0_6b()
eng = ParakeetEngine(pd, parakeet_state_dict(pd, seed=12, gain=1.0), dtype=torch.bfloat16, max_batch=32, max_seconds=30.0)
for name, audios in (("1 x 30 s", [synth_audio(480000, 1)]), ("32 x 5 s", [synth_audio(80000, 10 + i) for i in range(32)])):
    feats, nf, _ = eng.features(audios)
    best, steps = 1e9, 0
    for it in range(6):
        eng.encode(feats, nf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks, _, _ = eng.tdt_greedy()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
        steps = max(len(t) for t in toks)
    print(f"TDT greedy {name}: {best * 1e3:7.2f} ms for {steps} steps ({best / max(steps, 1) * 1e6:.1f} us per step)  S2S_TDT_FUSED={os.environ.get('S2S_TDT_FUSED', '1')} S2S_TDT_GRAPH={os.environ.get('S2S_TDT_GRAPH', '0')}")
