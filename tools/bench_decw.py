"""Wide decode projections (csrc/gemm_decw.hip) inside a captured hipGraph: what a frame-loop row costs as the loop gets wider.

    python tools/bench_decw.py chain            talker / predictor layer chains (4 projections, attention not included) at 64 .. 512 rows:
                                                the <= 128-row kernels (and their 128-row slabs above) vs the row-block kernel
    python tools/bench_decw.py gemm [M]         single projections at M rows: 128-row slabs vs the row-block kernel at every K-slice count

VERDICT r3 next #1 asks for: talker chain at M = 512 <= 3 x the M = 64 time (r3: M = 128 was already 1.73 x).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench_dec as BD
from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"


def wtune(enable=1, splitk=0, min_m=129):
    native.check(lib.s2s_op_gemm_decw_tune(enable, splitk, min_m), "wide tune")


def chain():
    base = {}
    for M in (64, 128, 256, 384, 512):
        for wide in ((0, 1) if M == 128 else (0,) if M < 128 else (0, 1)):
            if not wide and M % 128 and M > 128:
                continue
            wtune(wide, 0, 65 if M <= 128 else 129)
            for tag, hid, ffn, nq, hd, L, steps in (("talker", 2048, 6144, 4096, 2048, 28, 1), ("predictor", 1024, 3072, 4096, 2048, 5, 15)):
                def w(r, c):
                    return BD.to_frag((torch.randn(r, c, device=dev) * c ** -0.5).to(torch.bfloat16))

                Wq = [w(nq, hid) for _ in range(L)]
                Wo = [w(hid, hd) for _ in range(L)]
                Wg = [w(2 * ffn, hid) for _ in range(L)]
                Wd = [w(hid, ffn) for _ in range(L)]
                Mp = max(128, (M + 15) // 16 * 16)
                X = torch.randn(M, hid, device=dev) * 0.1
                XB = torch.zeros(Mp, hid, dtype=torch.bfloat16, device=dev)
                ssA = torch.ones(256, Mp, device=dev)
                ssB = torch.ones(256, Mp, device=dev)
                qkv = torch.zeros(Mp, nq, dtype=torch.bfloat16, device=dev)
                act = torch.zeros(Mp, ffn, dtype=torch.bfloat16, device=dev)
                parts = lib.s2s_op_gemm_dec_parts(M, hid, hd, 0, 1)

                def fn(st):
                    for _ in range(steps):
                        for l in range(L):
                            BD.dec(st, XB, Wq[l], M, nq, hid, ss=ssA, parts=parts, mode=0, Cb=qkv, layout=1 | 2, ldc=nq)
                            BD.dec(st, qkv, Wo[l], M, hid, hd, mode=2, X=X, XB=XB, sso=ssB, layout=1 | 4, ldxb=hid)
                            BD.dec(st, XB, Wg[l], M, ffn, hid, ss=ssB, parts=parts, mode=0, Cb=act, glu=True, layout=1 | 2, ldc=ffn)
                            BD.dec(st, act, Wd[l], M, hid, ffn, mode=2, X=X, XB=XB, sso=ssA, layout=1 | 4, ldxb=hid)
                    return steps * L

                us, n = BD.graph_time(fn, warm=2, reps=10)
                per = us / n
                if M == 64:
                    base[tag] = per
                wb = (nq * hid + hid * hd + 2 * ffn * hid + hid * ffn) * 2
                fl = 2.0 * M * (nq * hid + hid * hd + 2 * ffn * hid + hid * ffn)
                kind = f"row-block kernel/{wide}" if wide else ("<= 128-row kernels" if M <= 128 else "128-row slabs")
                print(f"{tag:9s} layer chain M={M:3d} ({kind:18s}): {per:7.2f} us per layer = {per / base[tag]:4.2f} x M=64, {per / M * 1e3:6.1f} ns per row, "
                      f"{wb / (per * 1e-6) / 1e12:5.2f} TB/s of weights, {fl / (per * 1e-6) / 1e12:6.1f} TFLOP/s", flush=True)
                del Wq, Wo, Wg, Wd
    wtune(1)


def gemm(M=512):
    for name, M_, N, K, mode, glu in BD.gemm_shapes(M)[:10]:
        rows = 2 * N if glu else N
        wbytes = rows * K * 2
        nbuf = max(2, min(32, int(400e6 // wbytes) + 1))
        Ws = [BD.to_frag((torch.randn(rows, K, device=dev) * K ** -0.5).to(torch.bfloat16)) for _ in range(nbuf)]
        Mp = (M_ + 15) // 16 * 16
        A = torch.randn(Mp, K, device=dev).to(torch.bfloat16)
        ss = torch.rand(128, max(128, M_), device=dev)
        Cb = torch.zeros(Mp, N, dtype=torch.bfloat16 if mode == 0 else torch.float32, device=dev)
        X = torch.zeros(M_, N, device=dev)
        XB = torch.zeros(Mp, N, dtype=torch.bfloat16, device=dev)
        sso = torch.zeros(4096, max(128, M_), device=dev)
        steps = K // 32
        variants = ([("slabs", (0, 0))] if M_ % 128 == 0 or M_ <= 128 else []) + [("auto", (1, 0))] + [(f"{sh}/sk{sk}", (sh, sk)) for sh in ((44, 24) if glu else (44, 42, 24, 22)) for sk in (1, 2, 4) if steps % sk == 0 and steps // sk >= 4]
        res = []
        for tag, (en, sk) in variants:
            wtune(en, sk, 65 if M_ <= 128 else 129)

            def fn(st):
                for W in Ws:
                    if mode == 2:
                        BD.dec(st, A, W, M_, N, K, mode=2, X=X, XB=XB, sso=sso, layout=5)
                    else:
                        BD.dec(st, A, W, M_, N, K, ss=ss, parts=32, mode=mode, Cb=Cb, glu=glu, layout=1)
                return len(Ws)

            us, n = BD.graph_time(fn)
            fl = 2.0 * M_ * rows * K
            res.append(f"{tag}: {us / n:6.2f} us {fl / (us / n * 1e-6) / 1e12:6.1f} TF/s")
        wtune(1)
        print(f"{name:9s} M={M_:3d} N={N:6d} K={K:5d} {wbytes / 1e6:6.1f} MB x{nbuf:2d}  " + "   ".join(res), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "chain"
    if which == "chain":
        chain()
    else:
        gemm(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
