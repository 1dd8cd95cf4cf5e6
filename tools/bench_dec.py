"""Decode-path microbenchmarks INSIDE a captured hipGraph (development aid; not part of the graded bench).

A python loop of ctypes launches is host-bound (>5 us per call), which hid the real per-kernel cost in round 1
(tools/bench_ops.py).  Here every measured sequence is captured once (s2s_graph_bench_begin / _end) and replayed, so the
numbers are device time per kernel including the in-graph kernel boundary.

    python tools/bench_dec.py floor            trivial-kernel boundary cost inside a graph
    python tools/bench_dec.py gemm             single decode GEMMs, weights cycled through > 256 MB (HBM, not MALL)
    python tools/bench_dec.py layer [M]        norm-free decoder layer chains (talker 2048/6144 x 28, predictor 1024/3072 x 5 x 15), both kernels
    python tools/bench_dec.py sweep [M]        single GEMMs: 16-column kernel vs the column-block kernel at every K-slice count / block width
`gemm-old` / `layer-old` time the r1 skinny kernel (row-major operands, rmsnorm launches) on the same shapes.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"


def graph_time(fn, warm=3, reps=20):
    st = C.c_void_p()
    native.check(lib.s2s_graph_bench_begin(C.byref(st)), "begin")
    n = fn(st.value)
    us = C.c_double()
    native.check(lib.s2s_graph_bench_end(warm, reps, C.byref(us)), "end")
    return us.value, n


def to_frag(W):
    rows, K = W.shape
    out = torch.zeros(((rows + 15) // 16 * 16) * K, dtype=torch.bfloat16, device=W.device)
    native.check(lib.s2s_op_to_frag(native.stream_handle(), native.ptr(W), W.stride(0), None, native.ptr(out), rows, K), "to_frag")
    return out


def dec(st, A, Wf, M, N, K, *, ss=None, parts=0, mode=0, Cb=None, X=None, XB=None, sso=None, glu=False, nf=0, bias=None, layout=7, ldc=None, ldxb=None):
    """Wf: fragment-major weights; layout bits: 1 A fragment-major, 2 C fragment-major, 4 XB fragment-major."""
    native.check(lib.s2s_op_gemm_dec(st, native.ptr(A), K, native.ptr(Wf), K, native.ptr(bias), M, N, K, native.ptr(ss), parts, max(128, M),
                                     1.0 / K, 1e-6, mode, native.ptr(Cb), ldc or N, 0, native.ptr(X), 0 if X is None else X.stride(0),
                                     native.ptr(XB), ldxb or N, native.ptr(sso), 1 if glu else 0, nf, layout), "gemm_dec")


def floor():
    x = torch.randn(4, 256, device=dev)
    xb = torch.zeros(4, 256, dtype=torch.bfloat16, device=dev)
    ss = torch.zeros(128, device=dev)

    def fn(st):
        for _ in range(200):
            native.check(lib.s2s_op_rows_prep(st, native.ptr(x), 256, native.ptr(xb), 256, native.ptr(ss), 4, 256, 0), "prep")
        return 200

    us, n = graph_time(fn)
    print(f"trivial kernel (rows_prep 4x256) inside a graph: {us / n:.2f} us per launch")


def gemm_shapes(M):
    return [("t.qkv", M, 4096, 2048, 0, False), ("t.o", M, 2048, 2048, 2, False), ("t.gu", M, 6144, 2048, 0, True), ("t.down", M, 2048, 6144, 2, False),
            ("p.qkv", M, 4096, 1024, 0, False), ("p.o", M, 1024, 2048, 2, False), ("p.gu", M, 3072, 1024, 0, True), ("p.down", M, 1024, 3072, 2, False),
            ("t.head", M, 3072, 2048, 1, False), ("p.head", M, 2048, 1024, 1, False), ("w.qkv", M, 3840, 1280, 0, False), ("w.fc1", M, 5120, 1280, 0, False),
            ("w.fc2", M, 1280, 5120, 2, False), ("w.logits", M, 51866, 1280, 1, False)]


def tune(enable=1, splitk=0, nt=0):
    native.check(lib.s2s_op_gemm_dec_tune(enable, splitk, nt), "tune")


def gemm(M=64, old=False, sweep=False):
    """per shape: the 16-column kernel (nf = 1), the column-block kernel with the library's choice, and (sweep) every K-slice count / block width"""
    for name, M_, N, K, mode, glu in gemm_shapes(M):
        rows = 2 * N if glu else N
        wbytes = rows * K * 2
        nbuf = max(2, min(64, int(400e6 // wbytes) + 1))
        Ws = [(torch.randn(rows, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nbuf)]
        if not old:
            Ws = [to_frag(W) for W in Ws]
        A = torch.randn(max(M_, 16 * ((M_ + 15) // 16)), K, device=dev).to(torch.bfloat16)      # same bytes either layout
        ss = torch.rand(128, max(128, M_), device=dev)
        Cb = torch.zeros(M_, N, dtype=torch.bfloat16 if mode == 0 else torch.float32, device=dev)
        X = torch.zeros(M_, N, device=dev)
        XB = torch.zeros(16 * ((M_ + 15) // 16), N, dtype=torch.bfloat16, device=dev)
        sso = torch.zeros(4096, max(128, M_), device=dev)
        chunks = K // 32 // (8 if min(M_, 128) <= 64 else 4)
        variants = [("old", None)] if old else [("nf=1", (0, 0, 0)), ("dec2", (1, 0, 0))]
        if sweep and not old:
            variants += [(f"sk{sk}/nt{nt}", (1, sk, nt)) for sk in (1, 2, 3, 4, 5, 6, 8, 12) if chunks % sk == 0 for nt in ((1,) if glu else (1, 2))]
        res = []
        for tag, tn in variants:
            if tn is not None:
                tune(*tn)
            nf = 1 if tag == "nf=1" else 0

            def fn(st):
                for W in Ws:
                    if old:
                        flags = 1 if glu else 0
                        native.check(lib.s2s_op_gemm_ex(st, native.ptr(A), K, native.ptr(W), None, native.ptr(X) if mode == 2 else None,
                                                        native.ptr(X) if mode == 2 else native.ptr(Cb), N, M_, N, K, native.S2S_BF16,
                                                        native.S2S_F32 if mode else native.S2S_BF16, 0, flags, 0.0), "gemm_ex")
                    elif mode == 2:
                        dec(st, A, W, M_, N, K, mode=2, X=X, XB=XB, sso=sso, nf=nf, layout=5)
                    else:
                        dec(st, A, W, M_, N, K, ss=ss, parts=32, mode=mode, Cb=Cb, glu=glu, nf=nf, layout=1)
                return len(Ws)

            us, n = graph_time(fn)
            res.append(f"{tag}: {us / n:6.2f} us {wbytes / (us / n * 1e-6) / 1e12:5.2f} TB/s")
        tune(1, 0, 0)
        print(f"{name:9s} M={M_:3d} N={N:6d} K={K:5d} {wbytes / 1e6:6.1f} MB x{nbuf:2d}  " + "   ".join(res), flush=True)


def layer(M=64, old=False):
    for tag, hid, ffn, nq, hd, L, steps in (("talker", 2048, 6144, 4096, 2048, 28, 1), ("predictor", 1024, 3072, 4096, 2048, 5, 15)):
        def w(r, c):
            return (torch.randn(r, c, device=dev) * c ** -0.5).to(torch.bfloat16)

        cv = (lambda t: t) if old is True else to_frag
        Wq = [cv(w(nq, hid)) for _ in range(L)]
        Wo = [cv(w(hid, hd)) for _ in range(L)]
        Wg = [cv(w(2 * ffn, hid)) for _ in range(L)]
        Wd = [cv(w(hid, ffn)) for _ in range(L)]
        Mp = max(128, M)
        X = torch.randn(M, hid, device=dev) * 0.1
        XB = torch.zeros(Mp, hid, dtype=torch.bfloat16, device=dev)
        H = torch.zeros(M, hid, dtype=torch.bfloat16, device=dev)
        g = torch.ones(hid, device=dev)
        ssA = torch.ones(256, Mp, device=dev)
        ssB = torch.ones(256, Mp, device=dev)
        qkv = torch.zeros(Mp, nq, dtype=torch.bfloat16, device=dev)
        act = torch.zeros(Mp, ffn, dtype=torch.bfloat16, device=dev)
        tune(0 if old == "nf1" else 1, 0, 0)
        parts = lib.s2s_op_gemm_dec_parts(M, hid, hd, 0, 1)
        old_k = old is True

        def fn(st):
            for _ in range(steps):
                for l in range(L):
                    if old_k:
                        native.check(lib.s2s_op_rmsnorm(st, native.ptr(X), native.ptr(g), native.ptr(H), M, hid, 1e-6, native.S2S_BF16), "norm")
                        native.check(lib.s2s_op_gemm_ex(st, native.ptr(H), hid, native.ptr(Wq[l]), None, None, native.ptr(qkv), nq, M, nq, hid, 1, 1, 0, 0, 0.0), "q")
                        native.check(lib.s2s_op_gemm_ex(st, native.ptr(qkv), nq, native.ptr(Wo[l]), None, native.ptr(X), native.ptr(X), hid, M, hid, hd, 1, 0, 0, 0, 0.0), "o")
                        native.check(lib.s2s_op_rmsnorm(st, native.ptr(X), native.ptr(g), native.ptr(H), M, hid, 1e-6, native.S2S_BF16), "norm")
                        native.check(lib.s2s_op_gemm_ex(st, native.ptr(H), hid, native.ptr(Wg[l]), None, None, native.ptr(act), ffn, M, ffn, hid, 1, 1, 0, 1, 0.0), "g")
                        native.check(lib.s2s_op_gemm_ex(st, native.ptr(act), ffn, native.ptr(Wd[l]), None, native.ptr(X), native.ptr(X), hid, M, hid, ffn, 1, 0, 0, 0, 0.0), "d")
                    else:
                        dec(st, XB, Wq[l], M, nq, hid, ss=ssA, parts=parts, mode=0, Cb=qkv, layout=1 | 2, ldc=nq)   # (the engine writes qkv row-major for the attention)
                        dec(st, qkv, Wo[l], M, hid, hd, mode=2, X=X, XB=XB, sso=ssB, layout=1 | 4, ldxb=hid)
                        dec(st, XB, Wg[l], M, ffn, hid, ss=ssB, parts=parts, mode=0, Cb=act, glu=True, layout=1 | 2, ldc=ffn)
                        dec(st, act, Wd[l], M, hid, ffn, mode=2, X=X, XB=XB, sso=ssA, layout=1 | 4, ldxb=hid)
            return steps * L

        us, n = graph_time(fn, warm=2, reps=10)
        wb = (nq * hid + hid * hd + 2 * ffn * hid + hid * ffn) * 2
        tune(1, 0, 0)
        kind = 'r1: 2 norms + 4 skinny GEMMs' if old is True else ('4 gemm_dec (16-column kernel), no norms' if old == 'nf1' else '4 gemm_dec (column-block kernel), no norms')
        print(f"{tag:9s} layer chain M={M} ({kind}; attention not included): {us / n:7.2f} us per layer, "
              f"{wb / 1e6:.1f} MB weights -> {wb / (us / n * 1e-6) / 1e12:.2f} TB/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    if which in ("all", "floor"):
        floor()
    if which in ("all", "gemm"):
        gemm(M)
    if which in ("sweep",):
        gemm(M, sweep=True)
    if which in ("all", "layer"):
        layer(M, old="nf1")
        layer(M, old=False)
    if which in ("layer-old",):
        layer(M, old=True)
    if which in ("gemm-old",):
        gemm(M, old=True)
