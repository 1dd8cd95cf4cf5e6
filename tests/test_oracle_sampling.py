"""The sampler oracle (oracle/sampling.py) against transformers' own logits processors (CPU; the chain every
`generate(do_sample=True)` applies: RepetitionPenalty -> Temperature -> TopK -> TopP)."""

import numpy as np
import pytest
import torch

from oracle.sampling import processed_logits, processed_probs

lp = pytest.importorskip("transformers.generation.logits_process")


def hf_chain(row, temperature, top_k, top_p, rep, hist):
    scores = torch.tensor(row, dtype=torch.float64)[None]
    ids = torch.tensor([[int(i) for i in hist if i >= 0]], dtype=torch.long)
    if rep != 1.0:
        scores = lp.RepetitionPenaltyLogitsProcessor(penalty=rep)(ids, scores)
    scores = lp.TemperatureLogitsWarper(temperature)(ids, scores)
    if top_k > 0:
        scores = lp.TopKLogitsWarper(top_k=top_k)(ids, scores)
    if top_p < 1.0:
        scores = lp.TopPLogitsWarper(top_p=top_p)(ids, scores)
    return scores[0].numpy()


@pytest.mark.parametrize("V,temperature,top_k,top_p,rep", [(64, 1.0, 0, 1.0, 1.0), (2048, 0.9, 50, 1.0, 1.0), (3072, 1.0, 50, 0.8, 1.05), (2048, 0.7, 0, 0.9, 1.0),
                                                             (300, 1.3, 7, 0.5, 1.3), (17, 0.5, 3, 0.2, 2.0)])
def test_processed_logits_match_transformers(V, temperature, top_k, top_p, rep):
    g = np.random.default_rng(V + top_k)
    for trial in range(4):
        row = g.standard_normal(V) * 2.0
        if trial == 1:
            row[: V // 3] = np.round(row[: V // 3])            # ties at the top-k boundary and in the sort
        hist = g.integers(0, V, size=9)
        hist[3] = -1                                           # padding entries of the device history are skipped
        want = hf_chain(row, temperature, top_k, top_p, rep, hist)
        got = processed_logits(row, temperature, top_k, top_p, rep, hist)
        assert np.array_equal(np.isfinite(got), np.isfinite(want)), "kept-token sets differ"
        keep = np.isfinite(want)
        np.testing.assert_allclose(got[keep], want[keep], rtol=1e-12, atol=1e-12)
        p = processed_probs(row, temperature, top_k, top_p, rep, hist)
        ref = torch.softmax(torch.tensor(want), dim=-1).numpy()
        np.testing.assert_allclose(p, ref, rtol=1e-10, atol=1e-14)


def test_top_p_keeps_at_least_the_most_probable_token():
    row = np.array([0.0, 10.0, -3.0, 1.0])
    p = processed_probs(row, 1.0, 0, 1e-6, 1.0)
    assert p[1] == 1.0 and p.sum() == 1.0
