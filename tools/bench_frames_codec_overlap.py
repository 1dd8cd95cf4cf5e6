"""Does the codec pass of round r hide under the frame loop of round r + 1?  Qwen3-TTS-1.7B dims, B live rows, 8-frame rounds: frames then codec on ONE
stream (what a wide lane does) against frames on one stream / engine and the codec of the previous round on another stream / engine replica.
python tools/bench_frames_codec_overlap.py [B]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = 6
dims = Qwen3TTSDims.tts_1_7b()
eng = TTSEngine.from_random(dims, torch.bfloat16, max_batch=B, max_prompt=128, max_frames=128, max_codec_frames=8, device="cuda:0", seed=0)
ceng = eng.replica()
g = np.random.default_rng(0)
prompts = [TTSPrompt(g.standard_normal((100 + (i % 28), dims.t_hidden)).astype(np.float32)) for i in range(B)]
codes = torch.randint(0, dims.codebook, (B, 8, dims.n_groups), dtype=torch.int32).to("cuda:0")


def frames(n_rounds):
    for _ in range(n_rounds):
        eng.generate(8, min_total_frames=128)


def codec(e, n_rounds):
    for _ in range(n_rounds):
        e.codec_stream_decode(codes)


eng.prefill(prompts)
eng.codec_stream_reset(); ceng.codec_stream_reset()
frames(1); codec(eng, 1); codec(ceng, 1)
torch.cuda.synchronize()
t = time.perf_counter(); frames(R); torch.cuda.synchronize(); t_f = (time.perf_counter() - t) / R
t = time.perf_counter(); codec(eng, R); torch.cuda.synchronize(); t_c = (time.perf_counter() - t) / R
eng.release(0, B); eng.prefill(prompts); eng.codec_stream_reset(); ceng.codec_stream_reset()
frames(1); codec(ceng, 1); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def on(stream, fn, *a):
    torch.cuda.set_device(0)
    with torch.cuda.stream(stream):
        fn(*a)
        stream.synchronize()


t = time.perf_counter()
th = [threading.Thread(target=on, args=(s1, frames, R)), threading.Thread(target=on, args=(s2, codec, ceng, R))]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize(); t_both = (time.perf_counter() - t) / R
print(f"{B} rows, per 8-frame round: frames {t_f * 1e3:.1f} ms, codec {t_c * 1e3:.1f} ms, serial {1e3 * (t_f + t_c):.1f} ms; on two streams {t_both * 1e3:.1f} ms "
      f"({(t_f + t_c) / t_both:.2f} x; perfect overlap would be {1e3 * max(t_f, t_c):.1f} ms)")
