"""Decode-step attention at the talker's 512-row shape (16 q / 8 kv heads x 128, cache positions ~ prompt + frame), inside a captured graph:
one workgroup per query head (rounds 2 - 5) against one workgroup per KV head serving both heads of the group (round 6).
    python tools/bench_attn_step.py [rows=512] [pos=165]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native

lib = native.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 165
dev = "cuda:0"
for name, H, Hkv, D, cap, layers in (("talker", 16, 8, 128, 256, 28), ("predictor", 16, 8, 128, 32, 5)):
    pos_v = min(P, cap - 1) if name == "talker" else 8
    qkv = torch.randn(B, (H + 2 * Hkv) * D, device=dev).bfloat16()
    # one cache per layer so that consecutive launches stream different memory (as the frame loop does)
    K = torch.randn(layers, B, cap, Hkv * D, device=dev).bfloat16()
    V = torch.randn(layers, B, cap, Hkv * D, device=dev).bfloat16()
    O = torch.empty(B, H * D, device=dev, dtype=torch.bfloat16)
    pos = torch.full((B,), pos_v, dtype=torch.int32, device=dev)
    qw = torch.ones(D, device=dev)
    cos, sin = torch.ones(cap, D // 2, device=dev), torch.zeros(cap, D // 2, device=dev)
    for share in ((0, 1, 0, 1) if name == "talker" else (1, 3, 1, 3)):      # predictor: workgroup kernel (1) vs one wave per item (3)
        native.check(lib.s2s_op_attn_step_tune(share))
        st = C.c_void_p()
        native.check(lib.s2s_graph_bench_begin(C.byref(st)), "begin")
        for l in range(layers):
            native.check(lib.s2s_op_attn_step(st, native.ptr(qkv), qkv.stride(0), native.ptr(K[l]), native.ptr(V[l]), K[l].stride(0), K[l].stride(1), native.ptr(O), O.stride(0),
                                              native.ptr(qw), native.ptr(qw), 1e-6, native.ptr(cos), native.ptr(sin), native.ptr(pos), B, H, Hkv, D, cap, 0, D ** -0.5, 1, native.S2S_BF16))
        us = C.c_double()
        native.check(lib.s2s_graph_bench_end(3, 10, C.byref(us)), "end")
        kv_bytes = B * Hkv * (pos_v + 1) * D * 2 * 2
        print(f"{name:9s} rows {B} pos {pos_v} share={share}: {us.value / layers:7.2f} us per launch  ({kv_bytes / 1e6:.1f} MB of cache rows once -> {kv_bytes / (us.value / layers) / 1e6:.2f} TB/s)")
    lib.s2s_op_attn_step_tune(-1)
