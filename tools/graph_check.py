"""GPU check: captured frame graph vs stream launches at the serving dims (f32: codes must be identical)."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from speech_to_speech_amd import native
from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

dims = Qwen3TTSDims.tts_1_7b()
print("dims", dims)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
eng = TTSEngine.from_random(dims, dt, max_batch=B, max_prompt=128, max_frames=128, max_codec_frames=33, device=torch.device("cuda", 0), seed=0)
g = torch.Generator().manual_seed(1)
prompts = [TTSPrompt(torch.randn(20 + 3 * (b % 5), dims.t_hidden, generator=g).numpy()) for b in range(B)]
outs = []
for graphs in (False, True):
    native.graphs_enable(graphs)
    eng.prefill(prompts)
    parts = []
    for _ in range(3):
        got, nfr, done = eng.generate(8, min_total_frames=24)
        parts.append(got.cpu())
    torch.cuda.synchronize()
    outs.append(torch.cat(parts, 1))
    print("graphs", graphs, "ok", outs[-1][0, :2].tolist())
same = (outs[0] == outs[1]).float().mean().item()
print(f"identical fraction {same:.4f}")
