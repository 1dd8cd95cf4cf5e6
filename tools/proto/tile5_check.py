"""GPU parity + timing of the staggered 8-phase 256x256 tile GEMM (csrc/gemm.hip gemm_tile5_kernel, switch S2S_TILE5=1).

    python tools/proto/tile5_check.py                 parity vs float64 on ragged / odd-K-tile / encoder shapes, then timing (variant 0)
    S2S_TILE5_VAR=1 | 2 python tools/proto/tile5_check.py    A/B arms: 1 = lock-step (no stagger), 2 = no s_setprio
    S2S_TILE5=0 python tools/proto/tile5_check.py     same shapes on the round-3 gemm_tile4 / gemm_tile3 path (the A/B reference)
    T5_REPS=n                                         repeat the parity block n times (race screen: the schedule is hand-counted)

Parity bar: f32 accumulation order only (<= 2e-4 of the row magnitude) plus one bf16 rounding (2^-8) for bf16 outputs.
"""
import ctypes as C
import os
import sys

os.environ.setdefault("S2S_TILE5", "2")        # 2: force the kernel on every shape it can run (1: production dispatch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"


def gemm(st, A, W, bias, residual, Cout, M, N, K, act):
    c_dt = native.S2S_F32 if Cout.dtype == torch.float32 else native.S2S_BF16
    native.check(lib.s2s_op_gemm(st, native.ptr(A), A.stride(0), native.ptr(W), native.ptr(bias), native.ptr(residual), native.ptr(Cout), Cout.stride(0),
                                 M, N, K, native.S2S_BF16, c_dt, act, 0), "s2s_op_gemm")


def parity():
    g = torch.Generator().manual_seed(0)
    for M, N, K, act, f32out in ((300, 200, 64, 0, False), (2048, 512, 128, 0, False), (2048, 768, 192, 0, False), (4096, 1280, 1280, native.ACT_GELU, False), (4000, 1300, 1280, 0, False),
                                 (4096, 1280, 5120, 0, True), (2300, 3840, 1280, 0, False), (2049, 513, 320, 0, True), (96000, 1280, 1280, 0, True)):
        A = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        if f32out:
            X = torch.randn(M, N, generator=g).to(dev)
            want = X.double() + A.double() @ W.double().T + bias.double()
            gemm(native.stream_handle(), A, W, bias, X, X, M, N, K, act)          # in-place f32 residual
            got = X.double()
            tol = 3e-4
        else:
            out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
            want = A.double() @ W.double().T + bias.double()
            if act == native.ACT_GELU:
                want = torch.nn.functional.gelu(want)
            gemm(native.stream_handle(), A, W, bias, None, out, M, N, K, act)
            got = out.double()
            tol = 6e-3
        torch.cuda.synchronize()
        err = (got - want).abs().max().item() / want.abs().max().item()
        print(f"parity M={M} N={N} K={K} act={act} {'f32+res' if f32out else 'bf16'}: max rel err {err:.2e} (tol {tol:.0e}) {'OK' if err <= tol else 'FAIL'}")
        assert err <= tol


def timing():
    for name, M, N, K, f32out in (("enc.qkv", 96000, 3840, 1280, False), ("enc.o", 96000, 1280, 1280, True), ("enc.fc1", 96000, 5120, 1280, False),
                                  ("enc.fc2", 96000, 1280, 5120, True), ("cross.kv", 96000, 2560, 1280, False)):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        Cout = torch.zeros(M, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
        bias = torch.zeros(N, device=dev)
        act = native.ACT_GELU if name == "enc.fc1" else 0

        def fn(st):
            for _ in range(4):
                gemm(st, A, W, bias, Cout if f32out else None, Cout, M, N, K, act)
            return 4

        st = C.c_void_p()
        native.check(lib.s2s_graph_bench_begin(C.byref(st)), "begin")
        n = fn(st.value)
        us = C.c_double()
        native.check(lib.s2s_graph_bench_end(2, 5, C.byref(us)), "end")
        t = us.value / n
        print(f"{name:9s} M={M} N={N} K={K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s  ({'tile5 variant ' + os.environ.get('S2S_TILE5_VAR', '0') if os.environ['S2S_TILE5'] != '0' else 'tile4'})")


if __name__ == "__main__":
    for _ in range(int(os.environ.get("T5_REPS", "1"))):
        parity()
    if os.environ.get("T5_NO_TIMING") is None:
        timing()
