"""Where a row-block decode launch (csrc/gemm_decw.hip) spends its time: per-workgroup phase timestamps written by the kernel itself
(s2s_op_gemm_dec_debug; 100 MHz counter), medians over the workgroups of one launch on fresh (HBM) weights.

    python tools/decw_phases.py [M] [shape]     (shape: 1 = the library's choice, 44 / 42 / 24 / 22)

stamps: 0 start, 1 ring primed + row statistics reduced, 2 first k-step landed (barrier passed), 3 main loop done, 4 ring drained,
5 K-slices combined (last arriver) / = 4, 6 stores drained."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_dec as BD
from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"


def run(M, name, N, K, mode, glu, nst, sk):
    rows = 2 * N if glu else N
    nbuf = max(4, min(32, int(400e6 // (rows * K * 2)) + 1))
    Ws = [BD.to_frag((torch.randn(rows, K, device=dev) * K ** -0.5).to(torch.bfloat16)) for _ in range(nbuf)]
    Mp = (M + 15) // 16 * 16
    A = torch.randn(Mp, K, device=dev).to(torch.bfloat16)
    ss = torch.rand(128, max(128, M), device=dev)
    Cb = torch.zeros(Mp, N, dtype=torch.bfloat16 if mode == 0 else torch.float32, device=dev)
    X = torch.zeros(M, N, device=dev)
    XB = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev)
    sso = torch.zeros(4096, max(128, M), device=dev)
    per = 8 * 4096
    dbg = torch.zeros(2 * nbuf * per, dtype=torch.int64, device=dev)
    native.check(lib.s2s_op_gemm_decw_tune(nst, sk, 65 if M <= 128 else 129), "tune")
    st = native.stream_handle()
    torch.cuda.synchronize()
    for it in range(2 * nbuf):
        native.check(lib.s2s_op_gemm_dec_debug(dbg.data_ptr() + it * per * 8), "dbg")
        W = Ws[it % nbuf]
        if mode == 2:
            BD.dec(st, A, W, M, N, K, mode=2, X=X, XB=XB, sso=sso, layout=5)
        else:
            BD.dec(st, A, W, M, N, K, ss=ss, parts=16, mode=mode, Cb=Cb, glu=glu, layout=1)
    torch.cuda.synchronize()
    native.check(lib.s2s_op_gemm_dec_debug(None), "dbg")
    native.check(lib.s2s_op_gemm_decw_tune(1, 0, 129), "tune")
    allt = dbg.view(2 * nbuf, -1, 8).cpu()
    outs, starts = [], []
    for it in range(nbuf, 2 * nbuf):
        t = allt[it]
        t = t[t[:, 0] > 0].double() * 0.01          # us
        t0 = t[:, 0].min()
        starts.append(float(t0))
        fin = t[:, 6] > 0
        d = {"wgs": float(t.shape[0]), "skew_max": float((t[:, 0] - t0).max()), "primed": float((t[:, 1] - t[:, 0]).median()),
             "first_data": float((t[:, 2] - t[:, 0]).median()), "loop": float((t[:, 3] - t[:, 2]).median()), "loop_max": float((t[:, 3] - t[:, 2]).max()),
             "drain": float((t[:, 4] - t[:, 3]).median())}
        if fin.any():
            d["combine"] = float((t[fin, 5] - t[fin, 4]).median())
            d["epilogue"] = float((t[fin, 6] - t[fin, 5]).median())
            d["total"] = float(t[fin, 6].max() - t0)
        outs.append(d)
    med = {k: sorted(r[k] for r in outs if k in r)[len(outs) // 2] for k in outs[0]}
    period = (starts[-1] - starts[0]) / max(1, len(starts) - 1)
    steps = K // 32
    print(f"{name:8s} M={M} N={N} K={K} shape={nst} sk={sk}: launch-to-launch {period:.2f} us | " + "  ".join(f"{k}={v:.2f}" if k != "wgs" else f"wgs={int(v)}" for k, v in med.items())
          + f" | {med['loop'] / steps * 1e3:.0f} ns per k-step of the whole K", flush=True)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    nsts = [int(sys.argv[2])] if len(sys.argv) > 2 else [1]
    for name, M_, N, K, mode, glu in BD.gemm_shapes(M)[:8]:
        for nst in nsts:
            for sk in (0,):
                run(M, name, N, K, mode, glu, nst, sk)
