"""GPU parity of the Parakeet-TDT engine (log-mel -> FastConformer -> TDT greedy, through the C ABI) vs the oracle.

float32 engine: encoder output within 2e-3 abs on valid frames, TDT tokens / durations / step frames BIT-EXACT.
bf16 engine: encoder within 8e-2 abs / 1.5e-2 mean; token+duration steps must agree until the first step where the
oracle's own top-2 margin (token or duration) is below BF16_MARGIN.
"""

import numpy as np
import pytest
import torch

from oracle import logmel as OL
from oracle import parakeet as OP
from speech_to_speech_amd.parakeet import ParakeetEngine
from speech_to_speech_amd.synth import synth_audio
from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict

pytestmark = pytest.mark.gpu
BF16_MARGIN = 0.3


@pytest.fixture(scope="module")
def ref():
    dims = ParakeetDims.test_tiny()
    sd = parakeet_state_dict(dims, seed=12, gain=1.5)
    lens = [48000, 30500, 80000, 16000]
    audios = [synth_audio(n, 300 + i) for i, n in enumerate(lens)]
    tmax = max(lens) // 160 + 1
    feats = torch.from_numpy(np.stack([OL.parakeet_logmel(a, 128, t_total=tmax)[0] for a in audios]).astype(np.float32))
    flen = torch.tensor([n // 160 for n in lens])
    with torch.no_grad():
        enc, elen = OP.encode(sd, feats, flen, dims.n_heads, dims.layers)
        toks, durs, frames, logits = OP.tdt_greedy(sd, enc, elen, dims.vocab, dims.blank, return_logits=True)
    return dims, sd, audios, enc, elen, toks, durs, frames, logits


def test_f32_engine_is_step_exact(ref):
    dims, sd, audios, enc, elen, toks, durs, frames, logits = ref
    eng = ParakeetEngine(dims, sd, dtype=torch.float32, max_batch=4, max_seconds=10.0)
    feats, nf, T = eng.features(audios)
    eng.encode(feats, nf)
    got, glen = eng.encoder_output()
    got, glen = got.cpu(), glen.cpu()
    assert glen.tolist() == elen.tolist()
    worst = 0.0
    for b in range(len(audios)):
        n = int(elen[b])
        worst = max(worst, (got[b, :n] - enc[b, :n]).abs().max().item())
    print(f"f32 parakeet encoder max_abs_err={worst:.3e} (|enc| mean {enc.abs().mean():.3f})")
    assert worst < 2e-3
    gt, gd, gf = eng.tdt_greedy()
    print("oracle", [len(t) for t in toks], toks[0][:24], durs[0][:24])
    print("engine", [len(t) for t in gt], gt[0][:24], gd[0][:24])
    assert gt == toks and gd == durs and gf == frames
    ids, starts = eng.transcribe_ids(audios)
    assert ids == [[t for t in tk if t != dims.blank] for tk in toks]
    eng.close()


def test_bf16_engine_matches_within_margin(ref):
    dims, sd, audios, enc, elen, toks, durs, frames, logits = ref
    eng = ParakeetEngine(dims, sd, dtype=torch.bfloat16, max_batch=4, max_seconds=10.0)
    feats, nf, T = eng.features(audios)
    eng.encode(feats, nf)
    got, _ = eng.encoder_output()
    got = got.cpu()
    mx, mean = 0.0, 0.0
    for b in range(len(audios)):
        n = int(elen[b])
        d = (got[b, :n] - enc[b, :n]).abs()
        mx, mean = max(mx, d.max().item()), max(mean, d.mean().item())
    print(f"bf16 parakeet encoder max_abs_err={mx:.3e} mean_abs_err={mean:.3e}")
    assert mx < 8e-2 * max(1.0, enc.abs().max().item()) and mean < 1.5e-2
    gt, gd, gf = eng.tdt_greedy()
    agree = 0
    V = dims.vocab
    for b in range(len(audios)):
        for s in range(min(len(gt[b]), len(toks[b]))):
            if gt[b][s] == toks[b][s] and gd[b][s] == durs[b][s]:
                agree += 1
                continue
            lg = logits[s][b]
            t2, d2 = torch.topk(lg[:V], 2).values, torch.topk(lg[V:], 2).values
            margin = min(float(t2[0] - t2[1]), float(d2[0] - d2[1]))
            print(f"bf16 divergence session {b} step {s}: oracle margin {margin:.4f}")
            assert margin < BF16_MARGIN
            break
    total = sum(len(t) for t in toks)
    print(f"bf16 parakeet agreed on {agree}/{total} steps")
    assert agree >= total // 3
    eng.close()


def test_tdt_step_variants_agree_step_for_step():
    """The fused 8-launch TDT step (default), the 14-launch step of rounds 1 - 5 (S2S_TDT_FUSED=0) and the captured step (S2S_TDT_GRAPH=1, with
    and without fusion) emit the same (token, duration, frame) steps for the same float32 engine - the switches are read once per process, so
    every variant runs in its own interpreter."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import json, sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from speech_to_speech_amd.parakeet import ParakeetEngine\n"
        "from speech_to_speech_amd.synth import synth_audio\n"
        "from speech_to_speech_amd.synth_weights import ParakeetDims, parakeet_state_dict\n"
        "d = ParakeetDims.test_tiny()\n"
        "e = ParakeetEngine(d, parakeet_state_dict(d, seed=12, gain=1.5), dtype=torch.float32, max_batch=3, max_seconds=8.0)\n"
        "f, nf, _ = e.features([synth_audio(n, 40 + i) for i, n in enumerate((96000, 40000, 70000))])\n"
        "e.encode(f, nf)\n"
        "print(json.dumps([list(x) for x in e.tdt_greedy()]))\n")
    outs = {}
    for name, env in (("fused", {}), ("unfused", {"S2S_TDT_FUSED": "0"}), ("graph", {"S2S_TDT_GRAPH": "1"}), ("graph-unfused", {"S2S_TDT_GRAPH": "1", "S2S_TDT_FUSED": "0"})):
        r = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    n_steps = [len(t) for t in outs["fused"][0]]
    print(f"TDT step variants: {n_steps} steps per session, all four variants identical")
    assert min(n_steps) > 8
    for name in ("unfused", "graph", "graph-unfused"):
        assert outs[name] == outs["fused"], name
