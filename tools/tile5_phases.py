"""Where a 256x256 tile of gemm_tile5_kernel spends its time (development hook s2s_op_gemm_tile_debug, include/s2s_mi355x_dev.h).

    S2S_TILE5_VAR=0|4|12 python tools/tile5_phases.py [shape ...]        shapes: qkv o fc1 fc2 kv (default: qkv o)

Per tile the kernel writes {start, prologue landed, main loop done, epilogue issued} on the 100 MHz counter plus HW_ID / XCC_ID.  Printed:
the phase durations (mean / p50 / p95), and per compute unit (XCC, SE, SH, CU from the hardware ids) the idle gap between one tile's
last store and the next tile's first instruction (workgroup launch for one-tile-per-workgroup grids, the loop turn for the persistent
arms), the number of tiles per unit and the span of the launch.
"""
import ctypes as C
import os
import sys

os.environ.setdefault("S2S_TILE5", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"
SHAPES = {"qkv": (96000, 3840, 1280, False, 0), "o": (96000, 1280, 1280, True, 0), "fc1": (96000, 5120, 1280, False, native.ACT_GELU),
          "fc2": (96000, 1280, 5120, True, 0), "kv": (96000, 2560, 1280, False, 0)}


def run(name):
    M, N, K, f32out, act = SHAPES[name]
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    Cout = torch.zeros(M, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    c_dt = native.S2S_F32 if f32out else native.S2S_BF16
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(tiles * 6, dtype=torch.int64, device=dev)

    def go():
        native.check(lib.s2s_op_gemm(native.stream_handle(), native.ptr(A), A.stride(0), native.ptr(W), native.ptr(bias), native.ptr(Cout) if f32out else None,
                                     native.ptr(Cout), Cout.stride(0), M, N, K, native.S2S_BF16, c_dt, act, 0), "s2s_op_gemm")

    for _ in range(3):
        go()
    torch.cuda.synchronize()
    native.check(lib.s2s_op_gemm_tile_debug(native.ptr(dbg)), "debug on")
    go()
    torch.cuda.synchronize()
    native.check(lib.s2s_op_gemm_tile_debug(None), "debug off")
    d = dbg.cpu().numpy().reshape(tiles, 6)
    d = d[d[:, 0] != 0]
    t = d[:, :4].astype(np.float64) / 100.0         # us
    t -= t[:, 0].min()
    hw, xcc = d[:, 4], d[:, 5] & 0xF
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    var = os.environ.get("S2S_TILE5_VAR", "0")
    span = t[:, 3].max()
    print(f"== {name} M={M} N={N} K={K} {'f32+res' if f32out else 'bf16'} variant {var}: {len(d)} of {tiles} tiles stamped, {len(np.unique(cu))} compute units, "
          f"span {span:.1f} us -> {2.0 * M * N * K / span / 1e6:.0f} TFLOP/s in the stamped launch")
    for label, x in (("prologue (start -> first tile landed)", t[:, 1] - t[:, 0]), ("main loop", t[:, 2] - t[:, 1]), ("epilogue (issue)", t[:, 3] - t[:, 2]),
                     ("whole tile", t[:, 3] - t[:, 0])):
        print(f"   {label:40s} mean {x.mean():7.2f}  p50 {np.percentile(x, 50):7.2f}  p95 {np.percentile(x, 95):7.2f}  max {x.max():7.2f} us")
    gaps, per_cu, first, last = [], [], [], []
    for c in np.unique(cu):
        tt = t[cu == c]
        tt = tt[np.argsort(tt[:, 0])]
        per_cu.append(len(tt))
        first.append(tt[0, 0])
        last.append(tt[-1, 3])
        if len(tt) > 1:
            gaps.extend(tt[1:, 0] - tt[:-1, 3])
    gaps = np.array(gaps) if gaps else np.zeros(1)
    per_cu, first, last = np.array(per_cu), np.array(first), np.array(last)
    print(f"   gap between two tiles of a compute unit  mean {gaps.mean():7.2f}  p50 {np.percentile(gaps, 50):7.2f}  p95 {np.percentile(gaps, 95):7.2f}  max {gaps.max():7.2f} us"
          f"   (negative = two workgroups overlapped on the unit)")
    print(f"   tiles per compute unit min {per_cu.min()} / mean {per_cu.mean():.2f} / max {per_cu.max()};  first start p95 {np.percentile(first, 95):.2f} us;  "
          f"unit finishes: first {last.min():.1f}, median {np.median(last):.1f}, last {last.max():.1f} us")
    main = t[:, 2] - t[:, 1]
    print(f"   main loop alone = {2.0 * 256 * 256 * K / np.percentile(main, 50) / 1e6 * len(np.unique(cu)):.0f} TFLOP/s at the median tile over {len(np.unique(cu))} units")


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["qkv", "o"]):
        run(nm)
