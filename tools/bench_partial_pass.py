"""Where a live-partial pass of Parakeet-TDT-0.6B dims spends its time (random weights): B growing windows spread over 0.5 .. 5 s, one
ProgressiveStream each (incremental log-mel), a tick = every window grows by 0.5 s.  Wall time (device-synchronised) of the three stages of
ParakeetEngine.transcribe_batch: per-window cached features, encoder pass, TDT greedy loop.  python tools/bench_partial_pass.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from speech_to_speech_amd.parakeet import ParakeetEngine, ProgressiveStream
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
pd = ParakeetDims.tdt_0_6b()
eng = ParakeetEngine(pd, parakeet_state_dict(pd, seed=12, gain=1.0), dtype=torch.bfloat16, max_batch=64, max_seconds=30.0)
turns = [synth_audio(6 * 16000, 100 + i) for i in range(B)]
start = [8000 * (1 + i % 9) for i in range(B)]               # 0.5 .. 4.5 s at the first tick
streams = [ProgressiveStream() for _ in range(B)]
acc = {"features": [], "encode": [], "tdt": [], "total": [], "steps": [], "rows": []}
for tick in range(4):
    wins = [t[: min(len(t), s + 8000 * tick)] for t, s in zip(turns, start)]
    T = max(8, max(len(a) for a in wins) // 160 + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats = torch.empty((B, T, pd.n_mels), dtype=torch.bfloat16, device="cuda:0")
    nf = torch.empty((B,), dtype=torch.int32, device="cuda:0")
    if os.environ.get("ROWS", "1") != "0":
        eng.stream_features_rows(streams, wins, list(range(B)), T, feats, nf)      # one call for the tick (round 6)
    else:
        for r in range(B):
            eng.stream_features(streams[r], wins[r], r, T, feats[r : r + 1], nf[r : r + 1])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng.encode(feats, nf)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    toks, _, _ = eng.tdt_greedy()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if tick:
        for k, v in (("features", t1 - t0), ("encode", t2 - t1), ("tdt", t3 - t2), ("total", t3 - t0), ("steps", max(len(t) for t in toks)), ("rows", B * eng._T)):
            acc[k].append(v)
m = {k: float(np.mean(v)) for k, v in acc.items()}
print(f"partial pass, {B} windows (0.5 .. 5.5 s): total {m['total'] * 1e3:.2f} ms = features {m['features'] * 1e3:.2f} + encoder {m['encode'] * 1e3:.2f} ({m['rows']:.0f} rows) "
      f"+ TDT {m['tdt'] * 1e3:.2f} ms ({m['steps']:.0f} steps, {m['tdt'] / max(m['steps'], 1) * 1e6:.1f} us per step)   S2S_TILE3_MID={os.environ.get('S2S_TILE3_MID', '1')} ROWS={os.environ.get('ROWS', '1')}")
