"""Where a column-block decode launch spends its time (development aid): per-workgroup phase timestamps written by the kernel itself
(s2s_op_gemm_dec_debug; 100 MHz counter), reduced to medians / maxima over the workgroups of ONE launch on fresh (HBM) weights.

    python tools/dec2_phases.py [M]

phases: launch skew (first start -> this start), A staged (start -> activations in LDS + barrier), main (weights waited + MFMAs),
publish (slab stores drained), ticket (atomic returned), combine (last arriver: slabs read and added), epilogue (stores drained)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native
from tools.bench_dec import dec, gemm_shapes, lib, to_frag, tune

dev = "cuda:0"


def run(M, name, N, K, mode, glu, sk, nt):
    rows = 2 * N if glu else N
    wbytes = rows * K * 2
    nbuf = max(4, min(64, int(400e6 // wbytes) + 1))          # > 256 MB of weights in rotation: every launch streams from HBM
    Ws = [to_frag((torch.randn(rows, K, device=dev) * K ** -0.5).to(torch.bfloat16)) for _ in range(nbuf)]
    A = torch.randn(16 * ((M + 15) // 16), K, device=dev).to(torch.bfloat16)
    ss = torch.rand(128, max(128, M), device=dev)
    Cb = torch.zeros(M, N, dtype=torch.bfloat16 if mode == 0 else torch.float32, device=dev)
    X = torch.zeros(M, N, device=dev)
    XB = torch.zeros(16 * ((M + 15) // 16), N, dtype=torch.bfloat16, device=dev)
    sso = torch.zeros(4096, max(128, M), device=dev)
    per = 8 * 1024
    dbg = torch.zeros(2 * nbuf * per, dtype=torch.int64, device=dev)
    tune(2, sk, nt)
    st = native.stream_handle()
    torch.cuda.synchronize()
    for it in range(2 * nbuf):          # back to back, no host sync: activations / residual stay hot, weights do not
        native.check(lib.s2s_op_gemm_dec_debug(dbg.data_ptr() + it * per * 8), "dbg")
        W = Ws[it % nbuf]
        if mode == 2:
            dec(st, A, W, M, N, K, mode=2, X=X, XB=XB, sso=sso, layout=5)
        else:
            dec(st, A, W, M, N, K, ss=ss, parts=32, mode=mode, Cb=Cb, glu=glu, layout=1)
    torch.cuda.synchronize()
    native.check(lib.s2s_op_gemm_dec_debug(None), "dbg")
    allt = dbg.view(2 * nbuf, -1, 8).cpu()
    rows_out = []
    starts = []
    for it in range(nbuf, 2 * nbuf):
        t = allt[it]
        t = t[t[:, 0] > 0].double() * 0.01          # us
        t0 = t[:, 0].min()
        starts.append(float(t0))
        last = t[:, 5] > 0
        end = torch.where(t[:, 6] > 0, t[:, 6], torch.where(t[:, 4] > 0, t[:, 4], t[:, 2]))
        d = {"wgs": t.shape[0], "skew_max": float((t[:, 0] - t0).max()), "ss_med": float((t[:, 7] - t[:, 0]).median()),
             "A_med": float((t[:, 1] - t[:, 7]).median()), "A_max": float((t[:, 1] - t[:, 0]).max()),
             "main_med": float((t[:, 2] - t[:, 1]).median()), "main_max": float((t[:, 2] - t[:, 1]).max()), "total": float(end.max() - t0)}
        if sk != 1 and (t[:, 3] > 0).any():
            d["publish_med"] = float((t[:, 3] - t[:, 2]).median())
            d["ticket_med"] = float((t[:, 4] - t[:, 3]).median())
            if last.any():
                d["combine_med"] = float((t[last, 5] - t[last, 4]).median())
                d["epi_med"] = float((t[last, 6] - t[last, 5]).median())
                d["last_start_med"] = float((t[last, 4] - t0).median())
        else:
            d["epi_med"] = float((t[:, 6] - t[:, 2]).median())
        rows_out.append(d)
    tune(1, 0, 0)
    keys = list(rows_out[0].keys())
    med = {k: sorted(r[k] for r in rows_out if k in r)[len(rows_out) // 2] for k in keys}
    period = (starts[-1] - starts[0]) / max(1, len(starts) - 1)
    print(f"{name:8s} M={M} N={N} K={K} sk={sk} nt={nt}: period={period:.2f}  " + "  ".join(f"{k}={v:.2f}" if k != "wgs" else f"wgs={int(v)}" for k, v in med.items()), flush=True)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    want = {"t.qkv": (1, 2, 4), "t.o": (4, 8), "t.down": (8,), "p.qkv": (4,), "w.logits": (1,)}
    for name, M_, N, K, mode, glu in gemm_shapes(M):
        for sk in want.get(name, ()):
            run(M, name, N, K, mode, glu, sk, 1)
