"""Micro-benchmarks of single kernels through the C ABI (development aid; not part of the graded bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"


def time_gemm(M, N, K, dt=torch.bfloat16, cdt=torch.float32, iters=200, resid=False):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    C = torch.zeros(M, N, device=dev, dtype=cdt)
    args = (native.stream_handle(), native.ptr(A), K, native.ptr(W), None, native.ptr(C) if resid else None, native.ptr(C), N, M, N, K,
            native.dtype_code(dt), native.dtype_code(cdt), 0)
    for _ in range(5):
        lib.s2s_op_gemm(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.s2s_op_gemm(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    gb = N * K * W.element_size() / 1e9
    tf = 2.0 * M * N * K / 1e12
    print(f"gemm M={M:6d} N={N:6d} K={K:6d} {str(dt)[6:]:>8s}->{str(cdt)[6:]:<8s} {us:9.1f} us  {gb / (us * 1e-6) / 1e3:6.2f} TB/s(W)  {tf / (us * 1e-6):8.1f} TFLOP/s")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "skinny"):
        for (M, N, K) in [(64, 4096, 1024), (64, 1024, 2048), (64, 6144, 1024), (64, 1024, 3072), (64, 4096, 2048), (64, 2048, 2048),
                          (64, 12288, 2048), (64, 2048, 6144), (64, 3840, 1280), (64, 1280, 1280), (64, 5120, 1280), (64, 1280, 5120),
                          (64, 51866, 1280), (1, 4096, 2048), (16, 4096, 2048), (128, 4096, 1024)]:
            time_gemm(M, N, K, cdt=torch.bfloat16)
    if which in ("all", "tile"):
        for (M, N, K) in [(96000, 3840, 1280), (96000, 1280, 1280), (96000, 5120, 1280), (96000, 1280, 5120), (8192, 8192, 8192), (4096, 4096, 4096),
                          (983040, 96, 672), (327680, 192, 1344), (81920, 384, 2688), (16384, 768, 5376), (983040, 96, 96)]:
            time_gemm(M, N, K, cdt=torch.bfloat16, iters=10)
    if which in ("mid",):      # FastConformer encoder projections at live-partial batch sizes (C5: a few hundred .. a few thousand rows)
        for M in (256, 512, 1024, 2048, 3072, 4096):
            for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (2048, 1024)]:
                time_gemm(M, N, K, cdt=torch.bfloat16, iters=100)
