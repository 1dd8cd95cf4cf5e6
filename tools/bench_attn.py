"""Flash attention (attn_mfma_kernel through s2s_op_attention, use_mfma = 1) on the encoders' shapes, graph-replayed:
    python tools/bench_attn.py                      Whisper-large-v3 encoder (64 x 20 heads x 1500 x 64) and FastConformer-0.6B (8 heads x 128, 375 frames)
    S2S_ATTN_DMA=0 python tools/bench_attn.py       the register-staged loop of rounds 2 - 4 (A/B)
The time includes the V transpose that precedes the kernel (one launch each)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"
for name, B, H, T, D in (("whisper-large-v3 enc", 64, 20, 1500, 64), ("whisper B=16", 16, 20, 1500, 64), ("fastconformer 30 s", 64, 8, 375, 128)):
    q = torch.randn(B, T, H * D, device=dev).to(torch.bfloat16)
    k = torch.randn(B, T, H * D, device=dev).to(torch.bfloat16)
    v = torch.randn(B, T, H * D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros(B * H * D * Tp, dtype=torch.bfloat16, device=dev)

    def fn(st):
        native.check(lib.s2s_op_attention(st, native.ptr(q), q.stride(1), native.ptr(k), native.ptr(v), k.stride(1), native.ptr(o), None, None, B, H, H, T, T, D,
                                          D ** -0.5, 0, 0, native.S2S_BF16, 1, native.ptr(vt)), "s2s_op_attention")

    fn(native.stream_handle())
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q[:2].view(2, T, H, D).transpose(1, 2).float(), k[:2].view(2, T, H, D).transpose(1, 2).float(),
                                                           v[:2].view(2, T, H, D).transpose(1, 2).float()).transpose(1, 2).reshape(2, T, H * D)
    err = (o[:2].float() - ref).abs().max().item()
    st = C.c_void_p()
    native.check(lib.s2s_graph_bench_begin(C.byref(st)), "begin")
    fn(st.value)
    us = C.c_double()
    native.check(lib.s2s_graph_bench_end(2, 5, C.byref(us)), "end")
    fl = 4.0 * B * H * T * T * D
    print(f"{name:22s} B={B} H={H} T={T} D={D}: {us.value:8.1f} us (transpose + attention)  {fl / us.value / 1e6:7.1f} TFLOP/s  max abs err vs f32 SDPA {err:.2e}  "
          f"(S2S_ATTN_DMA={os.environ.get('S2S_ATTN_DMA', '1')})")
