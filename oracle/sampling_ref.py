"""ORACLE (test infrastructure): logits processors and token selection used by the decode loop.

Restates the transformers==4.44.2 processors that indextts/gpt/model.py:710-715 reaches through generate()
(third-party; call-site order: repetition penalty -> temperature -> top-k -> top-p -> softmax -> multinomial|argmax):
  RepetitionPenaltyLogitsProcessor  score<0 ? score*p : score/p on every id present in input_ids
                                    (input_ids = the fake prefix of ones + 8192 + generated codes, model.py:658-667)
  TemperatureLogitsWarper           scores / t
  TopKLogitsWarper                  scores < kth-largest -> -inf   (ties with the k-th value are kept)
  TopPLogitsWarper                  ascending sort; remove while cumulative softmax <= 1-top_p; keep >= 1
Pinned by tests/golden/sampling.npz (produced with the installed transformers processors).

The random draw itself cannot match torch.multinomial; the product defines its own, reproduced here bit-for-bit:
Philox4x32-10 keyed by (seed_lo, seed_hi) with counter (row, step, 0, 0) -> u = (x0 >> 8) * 2^-24; candidates
ordered by (score desc, id asc); pick the first candidate whose running fp32 sum of exp(score-max) exceeds u*total.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf


def repetition_penalty(scores: np.ndarray, history: np.ndarray, penalty: float) -> np.ndarray:
    out = scores.astype(np.float32).copy()
    for b in range(out.shape[0]):
        ids = np.unique(history[b])
        s = out[b, ids]
        out[b, ids] = np.where(s < 0, s * np.float32(penalty), s / np.float32(penalty))
    return out


def temperature(scores, t):
    return (scores / np.float32(t)).astype(np.float32)


def top_k(scores, k):
    k = min(int(k), scores.shape[-1])
    out = scores.copy()
    kth = np.sort(scores, axis=-1)[:, -k][:, None]
    out[scores < kth] = NEG_INF
    return out


def _softmax(x):
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp((x - m).astype(np.float32))
    return e / e.sum(-1, keepdims=True, dtype=np.float32)


def top_p(scores, p, min_keep=1):
    out = scores.copy()
    for b in range(scores.shape[0]):
        order = np.argsort(scores[b], kind="stable")  # ascending
        sp = _softmax(scores[b][order][None])[0]
        cum = np.cumsum(sp, dtype=np.float32)
        remove = cum <= np.float32(1.0 - p)
        remove[-min_keep:] = False
        out[b, order[remove]] = NEG_INF
    return out


def typical(scores, mass=0.9, min_keep=1):
    """TypicalLogitsWarper (transformers 4.44.2 logits_process.py; reached from indextts/gpt/model.py:704-708 when
    typical_sampling=True): keep the tokens whose surprisal is closest to the entropy of the distribution until their
    probability mass exceeds `mass`."""
    out = scores.astype(np.float32).copy()
    for b in range(scores.shape[0]):
        x = out[b].astype(np.float64)
        m = np.max(x)
        lse = m + np.log(np.sum(np.exp(x - m)))
        logp = (x - lse).astype(np.float32)
        p_ = np.exp(logp)
        ent = -np.nansum(logp * p_, dtype=np.float32)
        shifted = np.abs(-logp - ent)
        order = np.argsort(shifted, kind="stable")               # ascending
        sl = out[b][order]
        cum = np.cumsum(_softmax(sl[None])[0], dtype=np.float32)
        last = min(int(np.sum(cum < np.float32(mass))), scores.shape[1] - 1)
        remove_sorted = shifted[order] > shifted[order][last]
        remove_sorted[:min_keep] = False
        out[b, order[remove_sorted]] = NEG_INF
    return out


def process(scores, history, rep_pen=10.0, temp=1.0, k=30, p=0.8):
    s = repetition_penalty(scores, history, rep_pen) if rep_pen != 1.0 else scores.astype(np.float32)
    if temp != 1.0:
        s = temperature(s, temp)
    if k and k > 0:
        s = top_k(s, k)
    if p is not None and p < 1.0:
        s = top_p(s, p)
    return s


# ------------------------------------------------------------------ Philox4x32-10 (Salmon et al., SC'11)
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter, key):
    c = [np.uint32(x) for x in counter]
    k = [np.uint32(x) for x in key]
    for _ in range(10):
        p0 = _M0 * np.uint64(c[0])
        p1 = _M1 * np.uint64(c[2])
        hi0, lo0 = np.uint32(p0 >> np.uint64(32)), np.uint32(p0 & np.uint64(0xFFFFFFFF))
        hi1, lo1 = np.uint32(p1 >> np.uint64(32)), np.uint32(p1 & np.uint64(0xFFFFFFFF))
        c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
        with np.errstate(over="ignore"):
            k = [np.uint32(k[0] + _W0), np.uint32(k[1] + _W1)]
    return c


def uniform01(seed: int, row: int, step: int) -> np.float32:
    x = philox4x32((row, step, 0, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return np.float32(int(x[0]) >> 8) * np.float32(1.0 / 16777216.0)


def pick(processed_row: np.ndarray, u: np.float32) -> int:
    """Inverse-CDF draw over the finite entries of one processed score row (see module docstring)."""
    ids = np.nonzero(np.isfinite(processed_row))[0]
    sc = processed_row[ids]
    order = np.lexsort((ids, -sc))
    ids, sc = ids[order], sc[order]
    e = np.exp((sc - sc[0]).astype(np.float32)).astype(np.float32)
    total = np.float32(0.0)
    for x in e:
        total = np.float32(total + x)
    thr = np.float32(u * total)
    run = np.float32(0.0)
    for i, x in enumerate(e):
        run = np.float32(run + x)
        if run > thr:
            return int(ids[i])
    return int(ids[-1])


def greedy(processed: np.ndarray) -> np.ndarray:
    """argmax with lowest-id tie-break (torch.argmax returns the first maximal index)."""
    return np.argmax(processed, axis=-1)
