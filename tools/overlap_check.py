"""GPU experiment: do two TTS engines on two streams/threads overlap on the device? (graphs on)"""
import sys, time, threading
import torch

sys.path.insert(0, ".")
from speech_to_speech_amd import native
from speech_to_speech_amd.synth_weights import Qwen3TTSDims
from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

dims = Qwen3TTSDims.tts_1_7b()
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NF = 32
e0 = TTSEngine.from_random(dims, torch.bfloat16, max_batch=B, max_prompt=64, max_frames=128, max_codec_frames=33, device=dev, seed=0)
e1 = TTSEngine(dims, None, torch.bfloat16, B, 64, 128, 33, None, dev, engine_tensors=e0._weights)
g = torch.Generator().manual_seed(1)
prompts = [TTSPrompt(torch.randn(24, dims.t_hidden, generator=g).numpy()) for b in range(B)]


def run(eng, stream, n):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(stream):
        eng.prefill(prompts)
        eng.generate(n, min_total_frames=128)
        stream.synchronize()


s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for graphs in (True, False):
    native.graphs_enable(graphs)
    run(e0, s0, 4); run(e1, s1, 4)
    torch.cuda.synchronize()
    t = time.perf_counter(); run(e0, s0, NF); torch.cuda.synchronize(); one = time.perf_counter() - t
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(e, s, NF)) for e, s in ((e0, s0), (e1, s1))]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); two = time.perf_counter() - t
    print(f"graphs={graphs} B={B}: one engine {one*1e3:.1f} ms ({one/NF*1e3:.2f} ms/frame); two engines concurrently {two*1e3:.1f} ms (ratio {two/one:.2f})")
