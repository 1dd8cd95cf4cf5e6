"""GPU parity of the primitive operators exported by the C ABI against plain float64/float32 references."""

import numpy as np
import pytest
import torch

from speech_to_speech_amd import native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gemm(A, W, bias, resid, act, c_dtype, M, N, K):
    lib = native.load()
    C = torch.empty((M, N), dtype=c_dtype, device=DEV)
    native.check(lib.s2s_op_gemm(native.stream_handle(), native.ptr(A), A.stride(0), native.ptr(W), native.ptr(bias),
                                 native.ptr(resid), native.ptr(C), C.stride(0), M, N, K, native.dtype_code(A.dtype),
                                 native.dtype_code(c_dtype), act))
    torch.cuda.synchronize()
    return C


def _ref(A, W, bias, resid, act):
    y = A.double().cpu() @ W.double().cpu().T
    if bias is not None:
        y = y + bias.double().cpu()
    if act == native.ACT_GELU:
        y = torch.nn.functional.gelu(y)
    elif act == native.ACT_SILU:
        y = torch.nn.functional.silu(y)
    elif act == native.ACT_RELU:
        y = torch.relu(y)
    if resid is not None:
        y = y + resid.double().cpu()
    return y


SHAPES = [(128, 128, 64), (300, 200, 136), (1500, 384, 384), (64, 1280, 1280), (1, 51864, 384), (17, 100, 72),
          (65, 130, 264),
          # mid-sized launches (fewer than 192 big tiles): the 128 x 128 tile with the FOUR-stage ring (round 6) - 3, 4 and 5 K-slabs (ring
          # start-up and tail), K not a multiple of the slab (zero-page path), ragged M / N edges
          (1109, 1024, 192), (700, 1000, 200), (333, 257, 320),
          # large-M shapes that take the 256x128 / 3-stage ring kernel (ragged M and N edges; 3, 5 and 6 K-slabs)
          (9000, 650, 192), (8192, 1024, 320),
          # encoder-sized bf16 projections: the 256x256 tile kernel (8 waves of 128x64, 4-stage ring of 32-wide K slabs) - ragged M / N
          # edges, 40 and 5 K-slabs
          (12700, 1300, 1280), (16000, 3840, 160)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_matches_reference(M, N, K, dt):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    # asymmetric, non-identity operands (transpose-detecting)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1 + torch.arange(N)[:, None] * 1e-4).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    for act, use_res, cdt in [(native.ACT_NONE, False, torch.float32), (native.ACT_GELU, False, dt),
                              (native.ACT_NONE, True, torch.float32), (native.ACT_SILU, False, torch.float32)]:
        C = _gemm(A, W, bias, resid if use_res else None, act, cdt, M, N, K)
        ref = _ref(A, W, bias, resid if use_res else None, act)
        err = (C.double().cpu() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = (2e-5 if cdt == torch.float32 and dt == torch.float32 else 1e-2) * max(1.0, scale)
        if dt == torch.bfloat16 and cdt == torch.float32:
            tol = 2e-4 * max(1.0, scale)   # bf16 products are exact in f32; only summation order differs
        print(f"gemm M={M} N={N} K={K} {dt} act={act} res={use_res} c={cdt}: max_err={err:.3e} (scale {scale:.2f})")
        assert err < tol


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 256), (6144, 2048, 128), (2560, 5120, 1280)])
def test_tile_gemm_straight_line_epilogues(M, N, K):
    """Every interior-tile epilogue of the 256 x 256 tile GEMM (csrc/gemm.hip tile256_epilogue_fast) plus what falls back to the general
    one: bf16 / f32 outputs, with and without bias, f32 residual into a bf16 output, the residual added IN PLACE (C == residual, the
    encoders' x += proj(...)), GELU (packed polynomial for bf16 outputs: within bf16 rounding of the exact erf form), SiLU, ReLU."""
    lib = native.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W = (torch.randn(N, K, generator=g) * (2.0 / K ** 0.5) + torch.arange(N)[:, None] * 1e-5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    for act, use_bias, res_mode, cdt in [(native.ACT_NONE, True, None, torch.bfloat16), (native.ACT_NONE, False, None, torch.bfloat16),
                                         (native.ACT_NONE, True, "f32", torch.bfloat16), (native.ACT_NONE, True, "inplace", torch.float32),
                                         (native.ACT_NONE, False, "f32", torch.float32), (native.ACT_GELU, True, None, torch.bfloat16),
                                         (native.ACT_GELU, True, None, torch.float32), (native.ACT_SILU, True, None, torch.bfloat16),
                                         (native.ACT_RELU, True, None, torch.bfloat16)]:
        C = resid.clone() if res_mode == "inplace" else torch.empty((M, N), dtype=cdt, device=DEV)
        rptr = native.ptr(C) if res_mode == "inplace" else (native.ptr(resid) if res_mode else None)
        native.check(lib.s2s_op_gemm(native.stream_handle(), native.ptr(A), A.stride(0), native.ptr(W), native.ptr(bias) if use_bias else None, rptr,
                                     native.ptr(C), C.stride(0), M, N, K, native.S2S_BF16, native.dtype_code(cdt), act))
        torch.cuda.synchronize()
        ref = _ref(A, W, bias if use_bias else None, resid if res_mode else None, act)
        err = (C.double().cpu() - ref).abs().max().item()
        scale = ref.abs().max().item()
        tol = (2e-4 if cdt == torch.float32 else 6e-3) * max(1.0, scale)       # f32 out: summation order only; bf16 out: + one rounding (2^-8)
        print(f"tile gemm M={M} N={N} K={K} act={act} bias={use_bias} res={res_mode} c={cdt}: max_err={err:.3e} (scale {scale:.2f})")
        assert err < tol


def test_gelu_poly_matches_erf_form_over_a_wide_range():
    """the packed polynomial GELU of the 256 x 256 tile epilogue (csrc/gemm.hip gelu_poly2, interior tiles with a bf16 destination) against
    the exact erf form on inputs far outside the fitted interval - large negative values must come out as (almost) zero, not as -2e-5 |x|.
    out[m, n] = gelu(v[m] * 1) with v spanning [-1e4, 1e4]: A[:, 0] = v, W[:, 0] = 1, everything else zero."""
    M, N, K = 512, 512, 64
    v = torch.cat([torch.linspace(-8, 8, 256), -torch.logspace(1, 4, 128), torch.logspace(1, 4, 128)]).to(torch.bfloat16)
    A = torch.zeros(M, K, dtype=torch.bfloat16)
    A[:, 0] = v
    W = torch.zeros(N, K, dtype=torch.bfloat16)
    W[:, 0] = 1.0
    C = _gemm(A.to(DEV), W.to(DEV), None, None, native.ACT_GELU, torch.bfloat16, M, N, K).float().cpu()
    ref = torch.nn.functional.gelu(v.double())
    err = (C.double() - ref[:, None]).abs().max(dim=1).values
    tol = 2e-4 + ref.abs() * 2.0 ** -8          # the polynomial's 7.9e-5 + one bf16 rounding of the result
    worst = int((err - tol).argmax())
    print(f"gelu_poly2: max abs error {err.max().item():.3e}; at v = -1e4: {C[v.float() == v.float().min()][0, 0].item():.3e}")
    assert bool((err <= tol).all()), (float(v[worst]), float(C[worst, 0]), float(ref[worst]))
    assert C[v.float() < -4].abs().max().item() < 2e-4        # no linear growth on the negative tail


def test_gemm_identity_detects_transposes():
    M = N = K = 128
    A = torch.eye(M, K).to(DEV)
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0).to(DEV)
    C = _gemm(A, W, None, None, native.ACT_NONE, torch.float32, M, N, K)
    assert torch.equal(C.cpu(), W.cpu().T.contiguous())
    Ab, Wb = A.bfloat16(), W.bfloat16()
    Cb = _gemm(Ab, Wb, None, None, native.ACT_NONE, torch.float32, M, N, K)
    assert torch.equal(Cb.cpu(), Wb.float().cpu().T.contiguous())


@pytest.mark.parametrize("rows,cols", [(5, 384), (1500, 1280), (33, 1024), (4, 2048), (3, 640)])
def test_norms_match_reference(rows, cols):
    lib = native.load()
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.3).to(DEV)
    gamma = torch.randn(cols, generator=g).to(DEV)
    beta = torch.randn(cols, generator=g).to(DEV)
    for odt in (torch.float32, torch.bfloat16):
        y = torch.empty((rows, cols), dtype=odt, device=DEV)
        native.check(lib.s2s_op_layernorm(native.stream_handle(), native.ptr(x), native.ptr(gamma), native.ptr(beta),
                                          native.ptr(y), rows, cols, 1e-5, native.dtype_code(odt)))
        ref = torch.nn.functional.layer_norm(x.double().cpu(), (cols,), gamma.double().cpu(), beta.double().cpu(), 1e-5)
        err = (y.double().cpu() - ref).abs().max().item()
        print(f"layernorm {rows}x{cols} {odt}: {err:.3e}")
        assert err < (2e-5 if odt == torch.float32 else 5e-2)
        native.check(lib.s2s_op_rmsnorm(native.stream_handle(), native.ptr(x), native.ptr(gamma), native.ptr(y), rows,
                                        cols, 1e-6, native.dtype_code(odt)))
        xd = x.double().cpu()
        ref = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * gamma.double().cpu()
        err = (y.double().cpu() - ref).abs().max().item()
        print(f"rmsnorm {rows}x{cols} {odt}: {err:.3e}")
        assert err < (2e-5 if odt == torch.float32 else 5e-2)
