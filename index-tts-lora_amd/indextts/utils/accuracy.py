"""Accuracy figures of a reduced-precision decode run against a reference run of the SAME teacher-forced token sequence.

Used by tests/test_configs_gpu.py (reference = the CPU oracle) and by bench.py's `accuracy` object (reference = this build's
fp32 engine, which the tests hold to the oracle within 1e-3).  Everything is plain torch on whatever device the logits are on.

For logits [steps, B, V] of both runs and the forced codes [B, steps]:
  max_abs / rms          error of the raw logits (north_star's bound is quoted on these, for the fp32 mode);
  top1_agree             fraction of (step, row) where both runs pick the same greedy token AFTER the repetition penalty,
                         counted where the reference's top-2 margin exceeds `margin` (a thinner margin is a coin flip in
                         any arithmetic);
  kl_softmax             mean KL(P_ref || P_test) of the full softmax after the repetition penalty and temperature;
  tv_sampling            mean total-variation distance of the distributions the sampler actually draws from (repetition
                         penalty -> temperature -> top-k -> top-p, the HF processor order reached from
                         indextts/gpt/model.py:710-715) -- finite even where the two kept sets differ at their edge.
"""
from __future__ import annotations

import torch


def _penalise(scores: torch.Tensor, hist: torch.Tensor, penalty: float) -> torch.Tensor:
    """scores [B,V]; hist int64 [B,n] ids seen so far (duplicates allowed)."""
    if penalty == 1.0:
        return scores
    s = scores.gather(1, hist)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, hist, s)


def _sampling_dist(scores: torch.Tensor, top_k: int, top_p: float) -> torch.Tensor:
    V = scores.shape[-1]
    s = scores
    if top_k and 0 < top_k < V:
        kth = s.topk(top_k, dim=-1).values[:, -1:]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = s.sort(dim=-1, descending=False)
        cum = srt.softmax(-1).cumsum(-1)
        remove = cum <= (1.0 - top_p)
        remove[:, -1] = False
        s = s.masked_fill(torch.zeros_like(remove).scatter(1, idx, remove), float("-inf"))
    return s.softmax(-1)


@torch.no_grad()
def logit_accuracy(test: torch.Tensor, ref: torch.Tensor, codes: torch.Tensor, extra_ids=(1, 8192), rep_penalty=10.0,
                   temperature=1.0, top_k=30, top_p=0.8, margin=0.05) -> dict:
    """test, ref: fp32 [steps, B, V] (step s = the logits token s is chosen from); codes int [B, >= steps - 1] = the tokens
    both runs were fed (token s is chosen at step s, so the history at step s is extra_ids + codes[:, :s])."""
    steps, B, V = ref.shape
    dev = ref.device
    test = test.to(dev, torch.float32)
    ref = ref.to(torch.float32)
    codes = codes.to(dev).long()
    extra = torch.tensor(list(extra_ids), dtype=torch.long, device=dev).expand(B, -1)
    d = test - ref
    out = {"steps": int(steps), "rows": int(B), "max_abs": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()),
           "ref_logit_rms": float(ref.pow(2).mean().sqrt())}
    agree = counted = 0
    kl = tv = 0.0
    worst_step = (0.0, 0)
    for s in range(steps):
        hist = torch.cat([extra, codes[:, :s]], 1)
        r = _penalise(ref[s].clone(), hist, rep_penalty) / temperature
        t = _penalise(test[s].clone(), hist, rep_penalty) / temperature
        top2 = r.topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > margin
        agree += int(((r.argmax(-1) == t.argmax(-1)) & sure).sum())
        counted += int(sure.sum())
        lp_r, lp_t = r.log_softmax(-1), t.log_softmax(-1)
        kl += float((lp_r.exp() * (lp_r - lp_t)).sum(-1).mean())
        tv += float(0.5 * (_sampling_dist(r, top_k, top_p) - _sampling_dist(t, top_k, top_p)).abs().sum(-1).mean())
        e = float(d[s].abs().max())
        if e > worst_step[0]:
            worst_step = (e, s)
    out.update(top1_agree=agree / max(counted, 1), top1_counted=counted, top1_margin=margin, kl_softmax=kl / steps,
               tv_sampling=tv / steps, max_abs_first_step=float(d[0].abs().max()), max_abs_last_step=float(d[-1].abs().max()),
               worst_step=int(worst_step[1]))
    return out


@torch.no_grad()
def waveform_accuracy(test: torch.Tensor, ref: torch.Tensor) -> dict:
    """test, ref: waveforms in [-1, 1], same shape [..., T]."""
    t, r = test.float().reshape(-1), ref.float().reshape(-1).to(test.device)
    d = t - r
    return {"samples": int(r.numel()), "rms_err": float(d.pow(2).mean().sqrt()), "max_abs": float(d.abs().max()),
            "ref_rms": float(r.pow(2).mean().sqrt()),
            "snr_db": float(10 * torch.log10(r.pow(2).mean() / d.pow(2).mean().clamp_min(1e-30)))}


@torch.no_grad()
def teacher_forced_logits(engine, emb: torch.Tensor, pad: torch.Tensor, codes: torch.Tensor, steps: int) -> torch.Tensor:
    """Logits [steps, B, V] of `engine` fed codes[:, s-1] at step s (prefill = step 0).  Drives the engine's own decode-step
    launches (GPTEngine._step_transformer) eagerly; the sampling launch only advances the loop state."""
    B = emb.shape[0]
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=1.0, seed=0)
    out = [engine.prefill(emb, pad, steps + 2)[:B].clone()]
    codes = codes.to(engine.device).to(torch.int32)
    engine.force_stop[:B] = -1
    skip, engine.skip_finished = engine.skip_finished, False   # a greedy stop token must not take a row out of the attention
    try:
        for s in range(1, steps):
            engine._sample(B, sp)                   # advances step counter / cache position (lazily, in the next launch)
            engine.tokens[:B] = codes[:, s - 1]
            engine.history[:B, s - 1] = codes[:, s - 1]
            engine.finished[:B] = 0
            engine._step_transformer(B)
            out.append(engine.logits[:B].clone())
    finally:
        engine.skip_finished = skip
    return torch.stack(out, 0)
