"""The codec decoder's k = 7 dilated convolutions as the engines launch them at 512 live rows (one 8-frame chunk per session), graph-replayed:
time per launch against the three floors - MFMA (2 M N K at 2.5 PFLOP/s), HBM (x rows in + y rows out at 6.3 TB/s) and the LDS-fill bytes the
tap-by-tap GEMM moves (A rows re-staged once per tap).   python tools/bench_conv.py [sessions=512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native

lib = native.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda:0"
for Cc, T in ((768, 256), (384, 1280), (192, 5120), (96, 15360)):
    for dil in (1, 9):
        lpad = 56
        x = (torch.randn(B, lpad + T, Cc, device=dev) * 0.5).bfloat16()
        w = (torch.randn(Cc, 7, Cc, device=dev) * (7 * Cc) ** -0.5).bfloat16()
        bias = torch.zeros(Cc, device=dev)
        y = torch.empty(B, T, Cc, device=dev, dtype=torch.bfloat16)
        t_us = {}
        for on in (0, 1):                       # 0: tap-by-tap GEMM (rounds 1 - 5), 1: tap-reuse kernel (round 6)
            native.check(lib.s2s_op_conv7_tune(on))
            st = C.c_void_p()
            native.check(lib.s2s_graph_bench_begin(C.byref(st)), "begin")
            native.check(lib.s2s_op_conv1d_causal_tm(st, native.ptr(x), native.ptr(w), native.ptr(bias), native.ptr(y), B, T, Cc, Cc, 7, dil, lpad, native.S2S_BF16, native.S2S_BF16, native.ACT_NONE), "conv")
            us = C.c_double()
            native.check(lib.s2s_graph_bench_end(2, 5, C.byref(us)), "end")
            t_us[on] = us.value
        lib.s2s_op_conv7_tune(-1)
        M = B * T
        flop = 2.0 * M * Cc * 7 * Cc
        hbm = 2.0 * M * Cc * 2
        tn = 256 if False else 128
        fill = (M / 256) * ((-(-Cc // tn)) * 256 * 7 * Cc * 2 + (-(-Cc // tn)) * tn * 7 * Cc * 2)
        fill7 = (M / 256) * (-(-Cc // tn)) * ((256 + 6 * dil) * Cc * 2 + tn * 7 * Cc * 2)
        print(f"C={Cc:4d} T={T:6d} dil={dil}: tap-by-tap {t_us[0]:8.1f} us ({flop / t_us[0] / 1e6 / 2500:.3f} of MFMA peak) -> tap-reuse {t_us[1]:8.1f} us  {flop / t_us[1] / 1e6:7.1f} TFLOP/s "
              f"({flop / t_us[1] / 1e6 / 2500:.3f}), LDS fill {fill7 / 1e9:5.2f} GB   floors: MFMA {flop / 2.5e15 * 1e6:6.1f} us, "
              f"HBM {hbm / 6.3e12 * 1e6:6.1f} us, LDS fill {fill / 1e9:5.2f} GB = {fill / (256 * 60e9) * 1e6:6.1f} us at 60 GB/s per CU")
        del x, y
