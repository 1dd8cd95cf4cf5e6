"""One wide projection, a few graph replays (counter runs): python tools/decw_one.py M N K [nst] [splitk]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_dec as BD
from speech_to_speech_amd import native

lib = native.load()
M, N, K = (int(v) for v in sys.argv[1:4])
nst = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
native.check(lib.s2s_op_gemm_decw_tune(nst, sk, 129 if M > 128 else 65), "tune")
nbuf = max(2, min(32, int(400e6 // (N * K * 2)) + 1))
Ws = [BD.to_frag((torch.randn(N, K, device="cuda:0") * K ** -0.5).to(torch.bfloat16)) for _ in range(nbuf)]
A = torch.randn((M + 15) // 16 * 16, K, device="cuda:0").to(torch.bfloat16)
Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda:0")


def fn(st):
    for W in Ws:
        BD.dec(st, A, W, M, N, K, mode=0, Cb=Cb, layout=1)
    return len(Ws)


us, n = BD.graph_time(fn, warm=1, reps=3)
print(f"M={M} N={N} K={K} nst={nst} sk={sk} flags={os.environ.get('S2S_DECW_FLAGS', '0')}: {us / n:.2f} us per launch ({N * K * 2 / 1e6:.1f} MB of weights, {nbuf} buffers)")
