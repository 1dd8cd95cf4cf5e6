"""GPU tests of the TTS frame loop's sampler (csrc/tts.hip block_sample_256): transformers' processor semantics (repetition penalty ->
temperature -> top-k -> top-p -> multinomial) checked against a numpy restatement - the SUPPORT of the samples must equal the
processed distribution's support exactly, the empirical frequencies must match it, sampling is reproducible per seed and
independent of batch composition, and degenerate settings reduce to the greedy pick."""

import numpy as np
import pytest
import torch

from oracle.sampling import processed_probs      # pinned against transformers' logits processors in tests/test_oracle_sampling.py
from speech_to_speech_amd import native

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sample(logits, temperature, top_k=0, top_p=1.0, rep=1.0, seed=1, hist=None):
    lib = native.load()
    B, V = logits.shape
    lg = logits.to(DEV).contiguous()
    out = torch.empty(B, dtype=torch.int32, device=DEV)
    h = None if hist is None else hist.to(torch.int32).to(DEV).contiguous()
    native.check(lib.s2s_op_sample_tokens(native.stream_handle(), native.ptr(lg), B, V, float(temperature), int(top_k), float(top_p), float(rep), int(seed),
                                          native.ptr(h), 0 if h is None else h.shape[1], native.ptr(out)), "s2s_op_sample_tokens")
    return out.cpu().numpy()


@pytest.mark.parametrize("V,temperature,top_k,top_p,rep", [(64, 1.0, 0, 1.0, 1.0), (2048, 0.9, 50, 1.0, 1.0), (3072, 1.0, 50, 0.8, 1.05), (2048, 0.7, 0, 0.9, 1.0),
                                                             (300, 1.3, 7, 0.5, 1.3)])
def test_sampler_matches_the_processed_distribution(V, temperature, top_k, top_p, rep):
    g = torch.Generator().manual_seed(V + top_k)
    row = torch.randn(V, generator=g) * 2.0
    hist = torch.randint(0, V, (9,), generator=g)
    hist[3] = -1
    B = 16384
    got = sample(row[None].repeat(B, 1), temperature, top_k, top_p, rep, seed=11, hist=hist[None].repeat(B, 1))
    p = processed_probs(row.numpy(), temperature, top_k, top_p, rep, hist.numpy())
    counts = np.bincount(got, minlength=V).astype(np.float64)
    assert set(np.nonzero(counts)[0]) <= set(np.nonzero(p)[0]), "sampled a token the logits processors removed"
    freq = counts / B
    big = p > 5e-3
    assert np.all(np.abs(freq[big] - p[big]) <= 5.0 * np.sqrt(p[big] * (1 - p[big]) / B) + 1e-3), np.abs(freq - p).max()
    assert abs(freq[p > 0].sum() - 1.0) < 1e-12


def test_sampler_is_seeded_and_row_keyed():
    g = torch.Generator().manual_seed(0)
    lg = torch.randn(257, 1000, generator=g)
    a, b, c = sample(lg, 1.0, 40, 0.9, seed=5), sample(lg, 1.0, 40, 0.9, seed=5), sample(lg, 1.0, 40, 0.9, seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    # the draw of a row depends on (seed, row index) only, not on what else is in the batch
    assert np.array_equal(sample(lg[:100], 1.0, 40, 0.9, seed=5), a[:100])
    # degenerate settings reduce to the greedy argmax
    am = lg.argmax(-1).numpy()
    assert np.array_equal(sample(lg, 1.0, top_k=1, seed=3), am)
    assert np.array_equal(sample(lg, 1e-4, seed=3), am)
    assert np.array_equal(sample(lg, 1.0, top_p=1e-6, seed=3), am)


def test_engine_sampling_top_k_1_equals_greedy_and_is_reproducible():
    from speech_to_speech_amd.synth_weights import Qwen3TTSDims, qwen3_tts_state_dict
    from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

    dims = Qwen3TTSDims.test_tiny()
    sd = qwen3_tts_state_dict(dims, seed=5)
    g = torch.Generator().manual_seed(3)
    prompts = [TTSPrompt(torch.randn(n, dims.t_hidden, generator=g).numpy()) for n in (6, 4, 9)]
    eng = TTSEngine(dims, sd, dtype=torch.float32, max_batch=4, max_prompt=16, max_frames=32, max_codec_frames=24)

    def run():
        eng.prefill(prompts)
        codes, _, _ = eng.generate(8, min_total_frames=8)
        return codes.cpu()

    greedy = run()
    for which in ("talker", "predictor"):
        eng.set_sampling(which, temperature=1.0, top_k=1, seed=9)
    assert torch.equal(run(), greedy)                       # top-k 1 == argmax, through the re-captured graph
    for which in ("talker", "predictor"):
        eng.set_sampling(which, temperature=1.0, top_k=50, top_p=0.8, repetition_penalty=1.05, seed=9)
    a, b = run(), run()
    # every utterance draws its own stream (keyed on seed, utterance id, frame, group): a second utterance does not replay the first one's draws
    assert not torch.equal(a, b) and not torch.equal(a, greedy)

    def run_slots(slot0, utt):
        eng.release(0, 4)
        eng.admit(prompts[:2], slot0, 8, 8, utterance_id0=utt)
        eng.step(4, 8)
        return eng.read_codes(slot0, 2, 0, 8)[0].cpu()

    # ... and the stream belongs to the utterance, not to the slot it landed in or to its batch-mates: reproducible per (seed, utterance id)
    u0, u0_other_slot, u1 = run_slots(0, 100), run_slots(2, 100), run_slots(0, 200)
    assert torch.equal(u0, u0_other_slot) and not torch.equal(u0, u1)
    for which in ("talker", "predictor"):
        eng.set_sampling(which, temperature=0.0)
    assert torch.equal(run(), greedy)
    eng.close()
