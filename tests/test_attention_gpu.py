"""GPU parity of the attention / conv-as-GEMM operators against float64 torch references."""

import math

import pytest
import torch

from speech_to_speech_amd import native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_attention(q, k, v, H, Hkv, scale, kv_len=None, bias=None, causal=False, window=0):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    D = q.shape[2] // H
    qh = q.double().view(B, Tq, H, D).transpose(1, 2)
    kh = k.double().view(B, Tk, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    vh = v.double().view(B, Tk, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    s = (qh * scale) @ kh.transpose(-1, -2)
    t = torch.arange(Tq)[:, None]
    j = torch.arange(Tk)[None, :]
    if bias is not None:
        idx = (Tk - 1) - t + j
        s = s + torch.gather(bias.double(), 3, idx.expand(B, H, Tq, Tk))
    allowed = torch.ones(B, 1, Tq, Tk, dtype=torch.bool)
    if kv_len is not None:
        allowed &= (j[None, None] < kv_len.view(B, 1, 1, 1))
    off = Tk - Tq
    if causal:
        allowed &= (j <= t + off)[None, None]
    if window > 0:
        allowed &= (j > t + off - window)[None, None]
    s = s.masked_fill(~allowed, -math.inf)
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, H * D)


def _run(q, k, v, H, Hkv, D, scale, kv_len=None, bias=None, causal=0, window=0, use_mfma=0):
    lib = native.load()
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    o = torch.empty((B, Tq, H * D), dtype=q.dtype, device=DEV)
    vt = torch.empty(B * H * D * ((Tk + 63) // 64 * 64), dtype=q.dtype, device=DEV) if use_mfma else None
    native.check(lib.s2s_op_attention(native.stream_handle(), native.ptr(q), q.stride(1), native.ptr(k), native.ptr(v),
                                      k.stride(1), native.ptr(o), native.ptr(kv_len), native.ptr(bias), B, H, Hkv, Tq, Tk, D,
                                      scale, causal, window, native.dtype_code(q.dtype), use_mfma, native.ptr(vt)))
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Hkv,Tq,Tk,D,causal,window", [
    (2, 2, 2, 70, 70, 64, 0, 0), (1, 4, 2, 1, 333, 128, 0, 0), (2, 4, 4, 37, 37, 32, 1, 0), (1, 2, 1, 50, 90, 64, 1, 24),
    (3, 6, 6, 1, 1500, 64, 0, 0)])
def test_row_attention(dt, B, H, Hkv, Tq, Tk, D, causal, window):
    g = torch.Generator().manual_seed(B * 100 + Tq + Tk)
    q = torch.randn(B, Tq, H * D, generator=g).to(dt)
    k = torch.randn(B, Tk, Hkv * D, generator=g).to(dt)
    v = torch.randn(B, Tk, Hkv * D, generator=g).to(dt)
    kv_len = torch.randint(max(1, Tk // 2), Tk + 1, (B,), generator=g).to(torch.int32) if not causal else None
    scale = D ** -0.5
    o = _run(q.to(DEV), k.to(DEV), v.to(DEV), H, Hkv, D, scale, kv_len.to(DEV) if kv_len is not None else None, None, causal, window)
    ref = _ref_attention(q, k, v, H, Hkv, scale, kv_len, None, bool(causal), window)
    err = (o.double().cpu() - ref).abs().max().item()
    print(f"row attention {dt} B{B} H{H}/{Hkv} Tq{Tq} Tk{Tk} D{D} causal{causal} w{window}: {err:.3e}")
    assert err < (2e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("B,H,T,D,with_bias,with_len", [(1, 2, 1500, 64, False, False), (2, 3, 200, 64, False, True),
                                                        (2, 2, 375, 128, True, True), (1, 8, 125, 128, True, False),
                                                        (2, 2, 64, 64, False, False), (1, 1, 129, 64, False, False)])
def test_mfma_attention(B, H, T, D, with_bias, with_len):
    g = torch.Generator().manual_seed(T + D)
    # fused QKV rows (stride 3*H*D), the layout the encoders produce
    qkv = (torch.randn(B, T, 3 * H * D, generator=g) * 1.2).bfloat16().to(DEV)
    q, k, v = qkv[:, :, : H * D], qkv[:, :, H * D : 2 * H * D], qkv[:, :, 2 * H * D :]
    bias = (torch.randn(B, H, T, 2 * T - 1, generator=g) * 0.7).to(DEV) if with_bias else None
    kv_len = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32).to(DEV) if with_len else None
    scale = D ** -0.5
    o = _run(q, k, v, H, H, D, scale, kv_len, bias, 0, 0, use_mfma=1)
    ref = _ref_attention(q.cpu(), k.cpu(), v.cpu(), H, H, scale, kv_len.cpu() if with_len else None,
                         bias.cpu() if with_bias else None)
    err = (o.double().cpu() - ref).abs().max().item()
    o2 = _run(q, k, v, H, H, D, scale, kv_len, bias, 0, 0, use_mfma=0)
    err2 = (o2.double().cpu() - ref).abs().max().item()
    print(f"mfma attention B{B} H{H} T{T} D{D} bias{with_bias} len{with_len}: mfma {err:.3e} rows {err2:.3e}")
    assert err < 3e-2 and err2 < 2e-2


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,Cin,Cout,k,stride", [(2, 300, 80, 128, 3, 1), (2, 300, 128, 64, 3, 2), (1, 97, 16, 24, 7, 1)])
def test_conv1d_as_gemm(dt, B, T, Cin, Cout, k, stride):
    lib = native.load()
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(T + Cin)
    x = torch.randn(B, Cin, T, generator=g).to(dt)
    w = (torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)).to(dt)
    b = torch.randn(Cout, generator=g)
    xp = torch.zeros(B, T + 2 * pad, Cin, dtype=dt)
    xp[:, pad : pad + T] = x.permute(0, 2, 1)
    T_out = (T + 2 * pad - k) // stride + 1
    y = torch.zeros(B, T_out + 2, Cout, dtype=dt, device=DEV)
    xd, wd, bd = xp.to(DEV), w.permute(0, 2, 1).contiguous().to(DEV), b.to(DEV)   # keep alive across the async launch
    native.check(lib.s2s_op_conv1d_tm(native.stream_handle(), native.ptr(xd), native.ptr(wd),
                                      native.ptr(bd), native.ptr(y), B, T, Cin, Cout, k, stride, pad, T_out + 2, 1,
                                      native.dtype_code(dt), native.dtype_code(dt), native.ACT_GELU))
    torch.cuda.synchronize()
    ref = torch.nn.functional.gelu(torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=stride, padding=pad)).permute(0, 2, 1)
    err = (y[:, 1 : 1 + T_out].double().cpu() - ref).abs().max().item()
    print(f"conv1d {dt} T{T} {Cin}->{Cout} k{k} s{stride}: {err:.3e}")
    assert err < (2e-5 if dt == torch.float32 else 3e-2)
    assert torch.all(y[:, 0] == 0) and torch.all(y[:, -1] == 0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,Cin,Cout,k,dil", [(2, 200, 32, 48, 7, 3),          # small: 128x128 tile kernel, per-slab tap arithmetic
                                                (4, 16000, 96, 96, 7, 9),        # codec-sized: 256x128 ring kernel, taps straddle K-slabs, K % 64 != 0
                                                (2, 30000, 192, 128, 7, 1),      # tap_len multiple of the slab
                                                (3, 20000, 64, 200, 1, 1)])      # pointwise (no taps), ragged N
def test_causal_dilated_conv_as_gemm(dt, B, T, Cin, Cout, k, dil):
    """The codec decoder's causal dilated convolution as ONE GEMM with tap addressing over the history-padded buffer."""
    lib = native.load()
    lpad = (k - 1) * dil + 2
    g = torch.Generator().manual_seed(T + Cin + dil)
    x = torch.randn(B, Cin, T, generator=g).to(dt)
    hist = torch.randn(B, Cin, lpad, generator=g).to(dt)          # non-zero history rows: they must be read, not assumed zero
    w = (torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)).to(dt)
    b = torch.randn(Cout, generator=g)
    xp = torch.cat([hist, x], dim=2).permute(0, 2, 1).contiguous()                # [B][lpad + T][Cin]
    y = torch.zeros(B, T, Cout, dtype=dt, device=DEV)
    xd, wd, bd = xp.to(DEV), w.permute(0, 2, 1).contiguous().to(DEV), b.to(DEV)
    native.check(lib.s2s_op_conv1d_causal_tm(native.stream_handle(), native.ptr(xd), native.ptr(wd), native.ptr(bd), native.ptr(y),
                                             B, T, Cin, Cout, k, dil, lpad, native.dtype_code(dt), native.dtype_code(dt), native.ACT_NONE))
    torch.cuda.synchronize()
    full = torch.cat([hist, x], dim=2).double()
    ref = torch.nn.functional.conv1d(full, w.double(), b.double(), dilation=dil)[:, :, lpad - (k - 1) * dil:].permute(0, 2, 1)
    assert ref.shape == (B, T, Cout)
    err = (y.double().cpu() - ref).abs().max().item()
    print(f"causal conv {dt} B{B} T{T} {Cin}->{Cout} k{k} d{dil}: {err:.3e}")
    assert err < (3e-5 if dt == torch.float32 else 4e-2)


@pytest.mark.parametrize("B,T,C,Cout,dil", [(8, 4096, 96, 96, 9), (8, 4096, 96, 96, 1), (4, 5120, 192, 192, 3), (16, 1280, 384, 384, 9), (128, 256, 768, 768, 1),
                                            (8, 4096, 64, 200, 3), (8, 4096, 96, 1, 1)])
def test_tap_reuse_conv7_matches_reference_and_the_tap_by_tap_gemm(B, T, C, Cout, dil):
    """The codec's k = 7 dilated convolutions on the tap-reuse kernel (csrc/gemm.hip conv7_tile_kernel: the tall input tile staged once per channel
    chunk, seven taps read from LDS at row offsets; items of 256-row multiples) against a float64 reference and against the tap-by-tap GEMM of
    rounds 1 - 5 (s2s_op_conv7_tune(0)): non-zero history rows, every dilation of the decoder, 64- and 32-channel chunks (C = 96), ragged /
    single output columns, the lpad > 6 d offset of the engines (56 rows)."""
    lib = native.load()
    k, lpad, dt = 7, 56, torch.bfloat16
    g = torch.Generator().manual_seed(T + C + dil)
    x = torch.randn(B, C, T, generator=g).to(dt)
    hist = torch.randn(B, C, lpad, generator=g).to(dt)
    w = (torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)).to(dt)
    b = torch.randn(Cout, generator=g)
    xd = torch.cat([hist, x], dim=2).permute(0, 2, 1).contiguous().to(DEV)
    wd, bd = w.permute(0, 2, 1).contiguous().to(DEV), b.to(DEV)
    outs = {}
    for on in (1, 0):
        y = torch.zeros(B, T, Cout, dtype=dt, device=DEV)
        native.check(lib.s2s_op_conv7_tune(on))
        try:
            native.check(lib.s2s_op_conv1d_causal_tm(native.stream_handle(), native.ptr(xd), native.ptr(wd), native.ptr(bd), native.ptr(y), B, T, C, Cout, k, dil, lpad,
                                                     native.S2S_BF16, native.S2S_BF16, native.ACT_NONE))
            torch.cuda.synchronize()
        finally:
            lib.s2s_op_conv7_tune(-1)
        outs[on] = y.double().cpu()
    full = torch.cat([hist, x], dim=2).double()
    ref = torch.nn.functional.conv1d(full, w.double(), b.double(), dilation=dil)[:, :, lpad - (k - 1) * dil:].permute(0, 2, 1)
    e1, e0 = (outs[1] - ref).abs().max().item(), (outs[0] - ref).abs().max().item()
    d10 = (outs[1] - outs[0]).abs().max().item()
    print(f"conv7 B{B} T{T} {C}->{Cout} d{dil}: tap-reuse err {e1:.3e}, tap-by-tap GEMM err {e0:.3e}, between them {d10:.3e}")
    assert e1 < 4e-2 and e0 < 4e-2 and d10 < 4e-2
    assert e1 < 1.5 * e0 + 1e-3                 # same products, another f32 summation order: the same error class


# ---- decode-step attention: grouped heads share one workgroup per KV head (round 6) ---------------------------------------------------
def _attn_step(share, qkv, K, V, pos, H, Hkv, D, kv_cap, qw, kw, cos, sin, window=0, o_frag=False):
    lib = native.load()
    B = qkv.shape[0]
    Kc, Vc = K.clone(), V.clone()
    O = torch.zeros(((B + 15) // 16 * 16, H * D), dtype=qkv.dtype, device=DEV)      # (fragment-major rows live in whole 16-row tiles)
    native.check(lib.s2s_op_attn_step_tune(share), "tune")
    try:
        native.check(lib.s2s_op_attn_step(native.stream_handle(), native.ptr(qkv), qkv.stride(0), native.ptr(Kc), native.ptr(Vc), Kc.stride(0), Kc.stride(1),
                                          native.ptr(O), O.stride(0), native.ptr(qw), native.ptr(kw), 1e-6, native.ptr(cos), native.ptr(sin), native.ptr(pos), B, H, Hkv, D,
                                          kv_cap, window, D ** -0.5, int(o_frag), native.dtype_code(qkv.dtype)), "s2s_op_attn_step")
        torch.cuda.synchronize()
    finally:
        lib.s2s_op_attn_step_tune(-1)
    return (O if o_frag else O[:B]), Kc, Vc


@pytest.mark.parametrize("H,Hkv,D,dt", [(16, 8, 128, torch.bfloat16), (16, 8, 128, torch.float32), (8, 2, 64, torch.bfloat16), (4, 4, 64, torch.float32)])
def test_attn_step_grouped_heads_match_the_per_head_kernel_bit_for_bit_and_the_reference(H, Hkv, D, dt):
    """One workgroup per (KV head, session) serving the group's 2 / 4 query heads (every K / V row loaded once) against one workgroup per
    query head: same outputs BIT FOR BIT, same cache rows appended; and both against a float64 reference of the fused step (head RMSNorm,
    rotary embedding, append, attention over the cache + the new row).  Ragged positions incl. an empty cache and a full one, sliding window."""
    B, kv_cap = 37, 96
    g = torch.Generator().manual_seed(H * 100 + D)
    qkv = (torch.randn(B, (H + 2 * Hkv) * D, generator=g)).to(dt).to(DEV)
    K = (torch.randn(B, kv_cap, Hkv * D, generator=g)).to(dt).to(DEV)
    V = (torch.randn(B, kv_cap, Hkv * D, generator=g)).to(dt).to(DEV)
    pos = torch.randint(0, kv_cap - 1, (B,), generator=g).to(torch.int32)
    pos[0], pos[1] = 0, kv_cap - 1
    qw, kw = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    ang = torch.arange(kv_cap)[:, None] * (10000.0 ** (-torch.arange(D // 2) / (D // 2)))[None, :]
    cos, sin = ang.cos().float().contiguous().to(DEV), ang.sin().float().contiguous().to(DEV)
    pos_d = pos.to(DEV)
    for window in (0, 24):
        o1, k1, v1 = _attn_step(1, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, window)
        o0, k0, v0 = _attn_step(0, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, window)
        assert torch.equal(o1, o0) and torch.equal(k1, k0) and torch.equal(v1, v0)
        # float64 reference
        x = qkv.double().cpu().view(B, H + 2 * Hkv, D)
        q, k, v = x[:, :H], x[:, H:H + Hkv], x[:, H + Hkv:]

        def norm_rope(t, w, p):
            t = t * torch.rsqrt((t * t).mean(-1, keepdim=True) + 1e-6) * w.double().cpu()
            a, c = t[..., : D // 2], t[..., D // 2:]
            cs, sn = cos.double().cpu()[p][:, None], sin.double().cpu()[p][:, None]
            return torch.cat([a * cs - c * sn, c * cs + a * sn], -1)

        pl = pos.long()
        qr, kr = norm_rope(q, qw, pl), norm_rope(k, kw, pl)
        if dt == torch.bfloat16:       # the kernel rounds q and the new k through the storage type
            qr, kr = qr.to(dt).double(), kr.to(dt).double()
        Kd, Vd = K.double().cpu().view(B, kv_cap, Hkv, D).clone(), V.double().cpu().view(B, kv_cap, Hkv, D).clone()
        want = torch.zeros(B, H, D, dtype=torch.float64)
        for b in range(B):
            Kd[b, pl[b]], Vd[b, pl[b]] = kr[b], v[b]
            lo = max(0, int(pl[b]) - window + 1) if window > 0 else 0
            for h in range(H):
                hk = h // (H // Hkv)
                s = (Kd[b, lo: pl[b] + 1, hk] @ qr[b, h]) * D ** -0.5
                want[b, h] = torch.softmax(s, 0) @ Vd[b, lo: pl[b] + 1, hk]
        err = (o1.double().cpu().view(B, H, D) - want).abs().max().item()
        print(f"attn_step H={H} Hkv={Hkv} D={D} {dt} window={window}: grouped == per-head bit for bit; max err vs float64 {err:.2e}")
        assert err < (2e-5 if dt == torch.float32 else 3e-2)
        new_k = torch.stack([k1.view(B, kv_cap, Hkv, D)[b, pl[b]] for b in range(B)]).double().cpu()
        assert (new_k - kr).abs().max().item() < (2e-5 if dt == torch.float32 else 3e-2)
    # fragment-major output (the operand layout of the out-projection): the same values, permuted
    of, _, _ = _attn_step(1, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, 0, o_frag=True)
    of0, _, _ = _attn_step(0, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, 0, o_frag=True)
    assert torch.equal(of, of0)


@pytest.mark.parametrize("H,Hkv,D,dt", [(16, 8, 128, torch.bfloat16), (16, 8, 128, torch.float32), (8, 2, 64, torch.bfloat16), (4, 4, 64, torch.float32)])
@pytest.mark.parametrize("kv_cap", [17, 32])
def test_attn_step_one_wave_per_item_matches_the_workgroup_kernel(H, Hkv, D, dt, kv_cap):
    """Short caches (the code predictor: <= 17 positions): one WAVE per (KV head, session) - attn_step_small_kernel - against the workgroup kernel:
    the same cache rows appended bit for bit, outputs equal up to the f32 summation order of the softmax denominator and of PV (one wave owns all
    keys instead of four waves a quarter each); every position from an empty cache to a full one, fragment-major output too."""
    B = 41
    g = torch.Generator().manual_seed(H * 10 + D + kv_cap)
    qkv = (torch.randn(B, (H + 2 * Hkv) * D, generator=g)).to(dt).to(DEV)
    K = (torch.randn(B, kv_cap, Hkv * D, generator=g)).to(dt).to(DEV)
    V = (torch.randn(B, kv_cap, Hkv * D, generator=g)).to(dt).to(DEV)
    pos = (torch.arange(B) % kv_cap).to(torch.int32)
    qw, kw = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (1 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    ang = torch.arange(kv_cap)[:, None] * (10000.0 ** (-torch.arange(D // 2) / (D // 2)))[None, :]
    cos, sin = ang.cos().float().contiguous().to(DEV), ang.sin().float().contiguous().to(DEV)
    pos_d = pos.to(DEV)
    tol = 2e-6 if dt == torch.float32 else 1.6e-2
    for frag in (False, True):
        big, kb, vb = _attn_step(1, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, 0, o_frag=frag)
        for tune in (3, 2):            # grouped heads per wave / one query head per wave
            small, ks, vs = _attn_step(tune, qkv, K, V, pos_d, H, Hkv, D, kv_cap, qw, kw, cos, sin, 0, o_frag=frag)
            assert torch.equal(ks, kb) and torch.equal(vs, vb)
            err = (small.double() - big.double()).abs().max().item()
            print(f"attn_step small H={H} Hkv={Hkv} D={D} {dt} cap={kv_cap} tune={tune} frag={frag}: max |small - workgroup| {err:.2e}")
            assert err <= tol * max(1.0, big.abs().max().item())
