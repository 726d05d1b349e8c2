"""ORACLE (test infrastructure): beam search / beam-sample bookkeeping of the decode loop (num_beams > 1).

The reference reaches this through `GenerationMixin.generate` (indextts/gpt/model.py:710-715; the defaults of
IndexTTS.infer are do_sample=True, num_beams=3, infer.py:807-814).  The algorithm lives in the third-party dependency
transformers==4.44.2 (requirements.txt:5), which is NOT installed here (the image has 5.15, whose beam search was
rewritten), so this file restates the published 4.44.2 algorithm -- `GenerationMixin._beam_search` +
`BeamSearchScorer.process/finalize` + `BeamHypotheses.add/is_done`.  Pinning: no reference-run fixture can be produced
offline, but tests/test_oracle_vs_golden.py::test_beam_oracle_matches_transformers_beam_search holds the beam-search mode
(do_sample=False: scorer bookkeeping, EOS handling, length penalty <= 1, repetition penalty on log-probabilities,
finalize) to the INSTALLED transformers' generate() on a toy GPT-2, token for token; the sampling mode differs only in how
the 2*num_beams candidates are drawn (own Philox stream below) and is unpinned.  Restated behaviour:

  per step, per batch element (num_beams rows):
    scores   = log_softmax(logits)                                   (fp32)
    scores   = RepetitionPenalty(scores, row's input_ids)            (processors run on log-probabilities)
    scores   = TopP(TopK(Temperature(scores)))   when do_sample      (min_tokens_to_keep = 2 because num_beams > 1)
    scores  += running beam score of the row     (first beam 0, others -1e9 at step 0)
    2*num_beams candidates over the num_beams*V flattened scores: multinomial WITHOUT replacement on
    softmax(scores) then sorted by score (do_sample), or the top 2*num_beams (beam search)
    BeamSearchScorer.process: walk the candidates in order; an EOS candidate among the first num_beams ranks closes a
    hypothesis (score = sum_logprobs / generated_len**length_penalty, the EOS itself is not stored), EOS candidates of
    lower rank are dropped, the first num_beams non-EOS candidates become the next beams; a batch element is done when
    it holds num_beams hypotheses and the worst of them is at least best_running / cur_len**length_penalty
  finalize: running beams of unfinished batch elements are added as hypotheses, the best hypothesis is returned,
  right-padded with pad (= EOS) and closed by one EOS when shorter than the longest.

The random draws cannot match torch.multinomial; the product defines its own stream, reproduced here bit-for-bit:
draw i of batch element b at step k uses Philox4x32-10 counter (b, k, i, 0) -> u in [0,1); the pool is ordered by
(score desc, flat index asc); draw = first not-yet-drawn entry whose running fp32 sum of exp(score - max) over the
not-yet-drawn entries exceeds u * (their total).
"""
from __future__ import annotations

import numpy as np

from . import sampling_ref as sr


def log_softmax(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m).astype(np.float32)
    s = np.float32(0.0)
    # fp32 sum (order-insensitive within tolerance; the kernel's tree order differs by a few ulp)
    s = e.sum(-1, keepdims=True, dtype=np.float32)
    return (x - m - np.log(s).astype(np.float32)).astype(np.float32)


def top_p_min2(scores, p):
    return sr.top_p(scores, p, min_keep=2)


def process_row(logits_row, history_row, sp, num_beams):
    """Processed log-probabilities of one row (−inf = removed)."""
    s = log_softmax(logits_row[None])
    if sp["repetition_penalty"] != 1.0:
        s = sr.repetition_penalty(s, history_row[None], sp["repetition_penalty"])
    if sp["do_sample"]:
        if sp["temperature"] != 1.0:
            s = sr.temperature(s, sp["temperature"])
        if sp["top_k"] and sp["top_k"] > 0:
            s = sr.top_k(s, max(int(sp["top_k"]), 2))
        if sp["top_p"] is not None and sp["top_p"] < 1.0:
            s = top_p_min2(s, sp["top_p"])
    return s[0]


def uniform01(seed, b, k, i):
    x = sr.philox4x32((b, k, i, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return np.float32(int(x[0]) >> 8) * np.float32(1.0 / 16777216.0)


def draw_without_replacement(pool_scores, n_draw, seed, b, k):
    """pool_scores sorted (score desc, index asc).  Returns the drawn pool positions, in draw order."""
    e = np.exp((pool_scores - pool_scores[0]).astype(np.float32)).astype(np.float32)
    alive = np.ones(e.shape[0], dtype=bool)
    out = []
    for i in range(min(n_draw, e.shape[0])):
        total = np.float32(0.0)
        for j in range(e.shape[0]):
            if alive[j]:
                total = np.float32(total + e[j])
        thr = np.float32(uniform01(seed, b, k, i) * total)
        run = np.float32(0.0)
        pick = -1
        last = -1
        for j in range(e.shape[0]):
            if not alive[j]:
                continue
            last = j
            run = np.float32(run + e[j])
            if run > thr:
                pick = j
                break
        if pick < 0:
            pick = last
        alive[pick] = False
        out.append(pick)
    return out


class BeamHyps:
    def __init__(self, num_beams, length_penalty):
        self.nb, self.lp = num_beams, float(length_penalty)
        self.beams = []  # (score, tokens)
        self.worst = np.float32(1e9)

    def add(self, tokens, sum_logprobs, generated_len):
        score = np.float32(np.float32(sum_logprobs) / np.float32(float(generated_len) ** self.lp))
        if len(self.beams) < self.nb or score > self.worst:
            self.beams.append((score, list(tokens)))
            if len(self.beams) > self.nb:
                order = sorted(range(len(self.beams)), key=lambda i: (self.beams[i][0], i))
                del self.beams[order[0]]
                self.worst = min(s for s, _ in self.beams)
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs, cur_len_generated):
        if len(self.beams) < self.nb:
            return False
        highest = np.float32(np.float32(best_sum_logprobs) / np.float32(float(cur_len_generated) ** self.lp))
        return bool(self.worst >= highest)


class BeamSearch:
    """State of one generate() call: B batch elements x num_beams rows.  `prefix_ids` = the fake prefix row the
    reference feeds as input_ids (ones, last = start token; model.py:658-667) -- it takes part in the repetition penalty
    and in nothing else."""

    def __init__(self, B, num_beams, sp, prefix_ids, eos, length_penalty=0.0, seed=0):
        self.B, self.nb, self.sp, self.eos, self.seed = B, num_beams, dict(sp), eos, int(seed)
        self.scores = np.zeros((B, num_beams), dtype=np.float32)
        self.scores[:, 1:] = np.float32(-1e9)
        self.hist = [[list(prefix_ids) for _ in range(num_beams)] for _ in range(B)]  # full input_ids rows
        self.plen = len(prefix_ids)
        self.hyps = [BeamHyps(num_beams, length_penalty) for _ in range(B)]
        self.done = [False] * B
        self.k = 0

    def step(self, logits):
        """logits fp32 [B*nb, V] of the current rows -> (next token per row [B*nb], source row per row [B*nb])."""
        B, nb, V = self.B, self.nb, logits.shape[-1]
        tokens = np.zeros(B * nb, dtype=np.int64)
        src = np.zeros(B * nb, dtype=np.int64)
        for b in range(B):
            if self.done[b]:
                tokens[b * nb:(b + 1) * nb] = self.eos
                src[b * nb:(b + 1) * nb] = b * nb
                self.scores[b] = 0
                self.hist[b] = [self.hist[b][0] + [self.eos] for _ in range(nb)]  # every row <- row 0 + pad (beam index 0)
                continue
            pool_s, pool_i = [], []
            for k in range(nb):
                s = process_row(logits[b * nb + k], np.asarray(self.hist[b][k]), self.sp, nb)
                s = (s + self.scores[b, k]).astype(np.float32)
                ids = np.nonzero(np.isfinite(s))[0]
                if not self.sp["do_sample"]:
                    # beam search: the global top 2*nb lie within each row's top 2*nb (ties: lower id first)
                    order = np.lexsort((ids, -s[ids]))[: 2 * nb]
                    ids = ids[order]
                pool_s.append(s[ids])
                pool_i.append(k * V + ids)
            ps, pi = np.concatenate(pool_s), np.concatenate(pool_i)
            order = np.lexsort((pi, -ps))
            ps, pi = ps[order], pi[order]
            if self.sp["do_sample"]:
                picks = draw_without_replacement(ps, 2 * nb, self.seed, b, self.k)
                picks = sorted(picks, key=lambda j: (-ps[j], picks.index(j)))  # torch.sort on scores, stable in draw order
            else:
                picks = list(range(min(2 * nb, ps.shape[0])))
            cur_gen = len(self.hist[b][0]) + 1 - self.plen
            nxt = []
            for rank, j in enumerate(picks):
                tok, beam, sc = int(pi[j] % V), int(pi[j] // V), np.float32(ps[j])
                if tok == self.eos:
                    if rank >= nb:
                        continue
                    self.hyps[b].add(self.hist[b][beam][self.plen:], sc, cur_gen)
                else:
                    nxt.append((tok, beam, sc))
                if len(nxt) == nb:
                    break
            if len(nxt) < nb:
                raise ValueError("fewer live continuations than beams (HF raises here too)")
            best = np.float32(max(ps[j] for j in picks))
            self.done[b] = self.done[b] or self.hyps[b].is_done(best, cur_gen)
            new_hist = []
            for k, (tok, beam, sc) in enumerate(nxt):
                tokens[b * nb + k], src[b * nb + k] = tok, b * nb + beam
                self.scores[b, k] = sc
                new_hist.append(self.hist[b][beam] + [tok])
            self.hist[b] = new_hist
        self.k += 1
        return tokens, src

    def all_done(self):
        return all(self.done)

    def finalize(self, num_return=1):
        """The num_return best hypotheses per batch element, best first (BeamSearchScorer.finalize with
        num_beam_hyps_to_keep = num_return_sequences: sorted by score, popped from the top) -> int64 [B * num_return, n]
        right-padded with EOS (generated part only); row = b * num_return + rank."""
        if not 1 <= num_return <= self.nb:
            raise ValueError("num_return_sequences has to be in [1, num_beams]")
        import copy
        best = []
        for b in range(self.B):
            hyps = copy.deepcopy(self.hyps[b])      # finalize may be called more than once: the search state stays as it is
            if not self.done[b]:
                gen_len = len(self.hist[b][0]) - self.plen
                for k in range(self.nb):
                    hyps.add(self.hist[b][k][self.plen:], self.scores[b, k], gen_len)
            order = sorted(range(len(hyps.beams)), key=lambda i: (hyps.beams[i][0], i))
            for j in range(num_return):
                best.append(hyps.beams[order[-1 - j]][1])
        # HF finalize: width = longest hypothesis + 1 (capped by max_length by the caller); pad = EOS, and every
        # hypothesis shorter than the width is closed by one EOS -- with pad == EOS that is plain EOS padding
        n = max(len(t) for t in best)
        out = np.full((len(best), n + 1), self.eos, dtype=np.int64)
        for b, t in enumerate(best):
            out[b, : len(t)] = t
        return out
