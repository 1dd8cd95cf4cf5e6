"""Oracle: the logits-processor chain of sampled generation in numpy float64 (test infrastructure, see oracle/__init__.py).

The reference's Qwen3-TTS backends sample inside packages that are not available offline (faster-qwen3-tts / qwen_tts,
reference TTS/qwen3_tts_handler.py:213-249); what IS available is the chain every Hugging Face `generate(do_sample=True)` applies,
transformers/generation/logits_process.py (5.15.0 in this container; the reference pins transformers>=4.57):
    RepetitionPenaltyLogitsProcessor :306-414   score > 0 ? score / penalty : score * penalty for every id already generated
    TemperatureLogitsWarper          :238-304   scores / temperature
    TopKLogitsWarper                 :542-596   remove everything below the k-th largest score (ties with it are kept)
    TopPLogitsWarper                 :473-540   sort ascending, remove the tokens whose cumulative probability <= 1 - top_p,
                                                always keep the most probable one
in exactly that order (GenerationMixin._get_logits_processor), followed by softmax and one multinomial draw.

PINNING: tests/test_oracle_sampling.py runs this restatement against those transformers classes themselves (CPU, random rows,
ties, penalties on negative scores).  The DRAW cannot be pinned: the device sampler uses a counter-based stream
(splitmix64 of seed, row, step), the backends use their own generators - tests compare the distribution, not the draws.
"""

from __future__ import annotations

import numpy as np


def processed_logits(row: np.ndarray, temperature: float, top_k: int, top_p: float, repetition_penalty: float, history=None) -> np.ndarray:
    """Scores after RepetitionPenalty -> Temperature -> TopK -> TopP (removed tokens are -inf)."""
    x = np.asarray(row, dtype=np.float64).copy()
    if repetition_penalty != 1.0 and history is not None:
        for t in set(int(i) for i in history if i >= 0):
            x[t] = x[t] / repetition_penalty if x[t] > 0 else x[t] * repetition_penalty
    x = x / temperature
    if 0 < top_k < len(x):
        kth = np.sort(x)[-top_k]
        x[x < kth] = -np.inf
    if top_p < 1.0:
        order = np.argsort(x, kind="stable")                  # ascending, like torch.sort(descending=False)
        p = np.exp(x[order] - x.max())
        p /= p.sum()
        remove = np.cumsum(p) <= 1.0 - top_p
        remove[-1] = False                                    # min_tokens_to_keep = 1
        x[order[remove]] = -np.inf
    return x


def processed_probs(row: np.ndarray, temperature: float, top_k: int, top_p: float, repetition_penalty: float, history=None) -> np.ndarray:
    """The distribution the multinomial draw is taken from."""
    x = processed_logits(row, temperature, top_k, top_p, repetition_penalty, history)
    p = np.exp(x - x[np.isfinite(x)].max())
    p[~np.isfinite(x)] = 0.0
    return p / p.sum()
