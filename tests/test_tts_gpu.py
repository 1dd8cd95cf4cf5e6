"""GPU parity of the Qwen3-TTS engine (talker + code predictor frame loop, codec decoder, 24->16 kHz int16
post-processing; all through the C ABI) against the oracle.

float32 engine: greedy codec codes BIT-EXACT over every frame and group; waveform within 2e-3 abs of the oracle.
bf16 engine: waveform (teacher-forced on the oracle's codes) within 1e-1 abs / 2e-2 mean (|wav| mean 0.14, ~60 conv layers deep); codes must agree on the
first frame and stay >= 50 % identical overall (random-weight margins are small; divergence is then self-reinforcing).
Resampler: same tap order and separately rounded multiply/add as scipy's upfirdn -> float output and int16 samples
are compared for exact equality with scipy.signal.resample_poly(float32) + clip/astype(int16).
"""

import numpy as np
import pytest
import torch

from oracle import qwen3_tts as OT
from speech_to_speech_amd.synth_weights import Qwen3TTSDims, qwen3_tts_state_dict
from speech_to_speech_amd.tts import AudioBlocker, TTSEngine, TTSPrompt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    dims = Qwen3TTSDims.test_tiny()
    sd = qwen3_tts_state_dict(dims, seed=5)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randn(6, dims.t_hidden, generator=g), torch.randn(4, dims.t_hidden, generator=g), torch.randn(9, dims.t_hidden, generator=g)]
    pad = torch.randn(dims.t_hidden, generator=g) * 0.5
    n_frames = 10
    codes = []
    with torch.no_grad():
        for p in prompts:   # the oracle runs one session at a time (no padding semantics to mirror)
            c, n = OT.talker_generate(sd, dims.talker_cfg(), dims.predictor_cfg(), p[None], dims.n_groups, n_frames, dims.codec_eos,
                                      suppress=[i for i in range(dims.codebook, dims.codec_vocab) if i != dims.codec_eos], pad_embed=pad,
                                      min_frames=n_frames)
            codes.append(c[0])
        codes = torch.stack(codes)                                 # [B, F, G]
        wav = OT.codec_decode(sd, dims.codec_cfg(), codes.permute(0, 2, 1))[:, 0]
    return dims, sd, prompts, pad, n_frames, codes, wav


def _engine(ref, dtype):
    dims, sd, prompts, pad, n_frames, codes, wav = ref
    return TTSEngine(dims, sd, dtype=dtype, max_batch=4, max_prompt=16, max_frames=32, max_codec_frames=24, pad_embed=pad)


def test_f32_codes_exact_and_waveform_close(ref):
    dims, sd, prompts, pad, n_frames, codes, wav = ref
    eng = _engine(ref, torch.float32)
    eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
    got, nfr, done = eng.generate(n_frames, min_total_frames=n_frames)
    got = got.cpu()
    print("oracle codes[0]", codes[0].tolist())
    print("engine codes[0]", got[0].tolist())
    assert nfr.cpu().tolist() == [n_frames] * 3 and not done
    assert torch.equal(got.long(), codes)
    w = eng.codec_decode(got.to(eng.device).contiguous(), 0).cpu()
    err = (w - wav).abs().max().item()
    print(f"f32 codec waveform max_abs_err={err:.3e} (|wav| mean {wav.abs().mean():.3f}, samples {w.shape[1]})")
    assert w.shape == wav.shape and err < 2e-3
    # incremental generation (two calls) continues the same stream
    eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
    a, _, _ = eng.generate(4, min_total_frames=n_frames)
    b, _, _ = eng.generate(6, min_total_frames=n_frames)
    assert torch.equal(torch.cat([a, b], 1).cpu().long(), codes)
    # chunked decode with left context == the sibling's chunked_decode pattern
    ctx = 3
    wc = eng.codec_decode(got[:, 2:10].to(eng.device).contiguous(), ctx).cpu()
    with torch.no_grad():
        ref_c = OT.codec_decode(sd, dims.codec_cfg(), codes[:, 2:10].permute(0, 2, 1))[:, 0][:, ctx * dims.total_upsample :]
    assert wc.shape == ref_c.shape and (wc - ref_c).abs().max().item() < 2e-3
    # exact causal streaming: chunks of 3, 3 and 4 frames concatenate to the whole-utterance decode
    eng.codec_stream_reset()
    parts = [eng.codec_stream_decode(got[:, a:b].to(eng.device).contiguous()).cpu() for a, b in ((0, 3), (3, 6), (6, 10))]
    ws = torch.cat(parts, 1)
    err_s = (ws - wav).abs().max().item()
    print(f"f32 streamed codec (3+3+4 frames) vs whole-utterance oracle: max_abs_err={err_s:.3e}, chunk samples {[p.shape[1] for p in parts]}")
    assert ws.shape == wav.shape and err_s < 2e-3
    eng.close()


def test_bf16_waveform_and_codes(ref):
    dims, sd, prompts, pad, n_frames, codes, wav = ref
    eng = _engine(ref, torch.bfloat16)
    w = eng.codec_decode(codes.to(torch.int32).to(eng.device).contiguous(), 0).cpu()
    d = (w - wav).abs()
    print(f"bf16 codec waveform max_abs_err={d.max():.3e} mean_abs_err={d.mean():.3e}")
    assert d.max() < 1e-1 and d.mean() < 2e-2
    eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
    got, _, _ = eng.generate(n_frames, min_total_frames=n_frames)
    got = got.cpu().long()
    same = (got == codes).float().mean().item()
    print(f"bf16 talker/predictor codes identical: {same * 100:.1f} %  first frame: {got[:, 0].tolist()} vs {codes[:, 0].tolist()}")
    assert torch.equal(got[:, 0, 0], codes[:, 0, 0]) and same >= 0.5
    eng.close()


def test_eos_stops_sessions(ref):
    dims, sd, prompts, pad, n_frames, codes, wav = ref
    eng = _engine(ref, torch.float32)
    eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
    got, nfr, done = eng.generate(16, min_total_frames=0)
    got, nfr = got.cpu(), nfr.cpu().tolist()
    with torch.no_grad():
        for b, p in enumerate(prompts):
            c, n = OT.talker_generate(sd, dims.talker_cfg(), dims.predictor_cfg(), p[None], dims.n_groups, 16, dims.codec_eos,
                                      suppress=[i for i in range(dims.codebook, dims.codec_vocab) if i != dims.codec_eos], pad_embed=pad)
            assert nfr[b] == int(n[0])
            assert torch.equal(got[b, : nfr[b]].long(), c[0, : nfr[b]])
            assert bool((got[b, nfr[b] :] == -1).all())
    eng.close()


def test_graph_replay_equals_stream_launches(ref):
    """The captured frame graph (default) and plain stream launches must give identical codes, also when the frames
    are produced over several generate() calls and the eos gate (min_total_frames) opens mid-way."""
    from speech_to_speech_amd import native

    dims, sd, prompts, pad, n_frames, codes, wav = ref
    outs = []
    for graphs in (True, False):
        native.graphs_enable(graphs)
        try:
            eng = _engine(ref, torch.float32)
            eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
            parts = []
            for n in (3, 4, 3):
                got, nfr, done = eng.generate(n, min_total_frames=5)
                parts.append(got.cpu())
            outs.append((torch.cat(parts, 1), nfr.cpu().tolist()))
        finally:
            native.graphs_enable(True)
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    # the first 5 frames cannot be eos-terminated and equal the oracle's (its eos is suppressed for all 10 frames)
    assert torch.equal(outs[0][0][:, :5].long(), codes[:, :5])


def test_resample_and_blocking_match_reference_post_processing(ref):
    from scipy.signal import resample_poly

    eng = _engine(ref, torch.float32)
    rng = np.random.default_rng(0)
    n = 14805
    x = (0.3 * rng.standard_normal((3, n))).astype(np.float32)
    x[1, :4000] = 0.0     # leading silence for the trim logic
    x = np.clip(x, -1, 1)
    y16, yf = eng.resample_int16(torch.from_numpy(x).to(eng.device), want_float=True)
    y16, yf = y16.cpu().numpy(), yf.cpu().numpy()
    for b in range(3):
        r = resample_poly(x[b], up=2, down=3)
        assert r.dtype == np.float32 and r.shape == yf[b].shape
        assert np.array_equal(r, yf[b])
        r16 = OT.to_int16(r)
        diff = np.abs(r16.astype(np.int32) - y16[b].astype(np.int32))
        print(f"resample row {b}: float max err {np.abs(r - yf[b]).max():.2e}, int16 mismatches {(diff > 0).mean() * 100:.3f} % (max {diff.max()} LSB)")
        assert diff.max() == 0
        # blocking: same logic as Qwen3TTSHandler._stream on the reference int16 stream
        blk = AudioBlocker()
        blocks = list(blk.push(y16[b])) + list(blk.flush())
        assert all(bl.shape == (512,) and bl.dtype == np.int16 for bl in blocks)
        thr = int(32768 * 0.01)
        above = np.abs(y16[b]) > thr
        start = max(0, int(np.argmax(above)) - 640)
        flat = np.concatenate(blocks)
        assert np.array_equal(flat[: len(y16[b]) - start], y16[b][start:])
    eng.close()


def test_streaming_text_mode_codes_match_the_oracle(ref):
    """Streaming-text mode (reference `non_streaming_mode = False`; C ABI s2s_tts_set_trailing, ABI v5): the prompt ends with the first text
    token, the remaining text rows are added one per frame to the next talker input in place of `pad_embed` (sibling
    modeling_qwen3_omni_moe.py:3169-3174).  float32 engine, ragged trailing lengths (12 rows > the 10 frames generated, 4 rows, none): every code
    equals the oracle's `talker_generate(..., trailing=...)`; a slot admitted later WITHOUT rows into a slot that had them is back in the
    pre-filled-text mode; continuous admission (s2s_tts_admit) carries the rows too."""
    dims, sd, prompts, pad, n_frames, codes_plain, _ = ref
    g = torch.Generator().manual_seed(11)
    trails = [torch.randn(12, dims.t_hidden, generator=g), torch.randn(4, dims.t_hidden, generator=g), None]
    sup = [i for i in range(dims.codebook, dims.codec_vocab) if i != dims.codec_eos]
    want = []
    with torch.no_grad():
        for p, t in zip(prompts, trails):
            c, _ = OT.talker_generate(sd, dims.talker_cfg(), dims.predictor_cfg(), p[None], dims.n_groups, n_frames, dims.codec_eos, suppress=sup,
                                      trailing=None if t is None else torch.cat([t, pad[None].expand(n_frames, -1)])[None], pad_embed=pad, min_frames=n_frames)
            want.append(c[0])
    want = torch.stack(want)
    assert not torch.equal(want[0], codes_plain[0]) and torch.equal(want[2], codes_plain[2])      # the rows change the stream; no rows = the plain mode
    eng = _engine(ref, torch.float32)
    mk = lambda: [TTSPrompt(p.numpy(), None if t is None else t.numpy()) for p, t in zip(prompts, trails)]      # noqa: E731
    eng.prefill(mk())
    got, _, _ = eng.generate(n_frames, min_total_frames=n_frames)
    assert torch.equal(got.cpu().long(), want)
    # the same slots again without rows: admission resets them to the pre-filled-text mode
    eng.prefill([TTSPrompt(p.numpy()) for p in prompts])
    got, _, _ = eng.generate(n_frames, min_total_frames=n_frames)
    assert torch.equal(got.cpu().long(), codes_plain)
    # continuous batching: session 1 (4 rows) joins slot 3 while slots 0 - 2 are live
    eng.release(0, 4)
    eng.admit(mk(), 0, 2 + n_frames, min_frames=n_frames)
    eng.step(3, 2)
    eng.admit([TTSPrompt(prompts[1].numpy(), trails[1].numpy())], 3, n_frames, min_frames=n_frames)
    eng.step(4, n_frames)
    late, _, _ = eng.read_codes(3, 1, 0, n_frames)
    assert torch.equal(late.cpu().long()[0], want[1])
    first, _, _ = eng.read_codes(0, 3, 0, n_frames)
    assert torch.equal(first.cpu().long(), want)
    eng.close()
