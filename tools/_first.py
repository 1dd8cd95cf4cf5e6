import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_dec as BD
from speech_to_speech_amd import native
mode = sys.argv[1]
dev = "cuda:0"
rows, K = 4096, 2048
Ws = []
for i in range(24):
    w = (torch.randn(rows, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    if mode == "sync":
        torch.cuda.synchronize(); print("made", i, hex(w.data_ptr()), flush=True)
    if mode != "nofrag":
        w = BD.to_frag(w)
    if mode == "sync":
        torch.cuda.synchronize(); print("frag", i, hex(w.data_ptr()), flush=True)
    Ws.append(w)
torch.cuda.synchronize()
print("ok", mode, flush=True)
