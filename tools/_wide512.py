import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dims = Qwen3TTSDims.tts_1_7b()
dims.t_layers = int(os.environ.get("TL", "2"))
eng = TTSEngine.from_random(dims, torch.bfloat16, max_batch=512, max_prompt=128, max_frames=128, max_codec_frames=8, device="cuda:0", seed=0)
torch.cuda.synchronize(); print("engine ok", flush=True)
g = np.random.default_rng(0)
prompts = [TTSPrompt(g.standard_normal((100 + (i % 28), dims.t_hidden)).astype(np.float32)) for i in range(B)]
eng.prefill(prompts); torch.cuda.synchronize(); print("prefill ok", flush=True)
eng.codec_stream_reset()
for it in range(3):
    codes, nfr, _ = eng.generate(8, min_total_frames=128); torch.cuda.synchronize(); print("generate ok", it, flush=True)
    wav = eng.codec_stream_decode(codes); torch.cuda.synchronize(); print("codec ok", it, tuple(wav.shape), float(wav.abs().max()), flush=True)
    pcm = eng.resample_int16(wav); torch.cuda.synchronize(); print("resample ok", it, flush=True)
