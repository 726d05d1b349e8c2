"""Pin the oracle (oracle/*.py) against fixtures produced by running the reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import bigvgan_ref, gpt_ref, sampling_ref

torch.set_grad_enabled(False)
G = os.path.join(os.path.dirname(__file__), "golden")


def gpt_shapes(layers):
    D = 1280
    s = {"text_embedding.weight": (12001, D), "mel_embedding.weight": (8194, D),
         "mel_pos_embedding.emb.weight": (803, D), "text_pos_embedding.emb.weight": (602, D),
         "gpt.ln_f.weight": (D,), "gpt.ln_f.bias": (D,), "final_norm.weight": (D,), "final_norm.bias": (D,),
         "mel_head.weight": (8194, D), "mel_head.bias": (8194,)}
    for i in range(layers):
        p = f"gpt.h.{i}."
        s.update({p + "ln_1.weight": (D,), p + "ln_1.bias": (D,), p + "attn.c_attn.weight": (D, 3 * D),
                  p + "attn.c_attn.bias": (3 * D,), p + "attn.c_proj.weight": (D, D), p + "attn.c_proj.bias": (D,),
                  p + "ln_2.weight": (D,), p + "ln_2.bias": (D,), p + "mlp.c_fc.weight": (D, 4 * D),
                  p + "mlp.c_fc.bias": (4 * D,), p + "mlp.c_proj.weight": (4 * D, D), p + "mlp.c_proj.bias": (D,)})
    return s


def gpt_weights(layers):
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(gpt_shapes(layers), synth.gpt_param).items()}


@pytest.mark.parametrize("tag,layers", [("gpt_small", 2), ("gpt_full", 24)])
def test_gpt_decode_and_latent(tag, layers):
    g = np.load(os.path.join(G, tag + ".npz"))
    W = gpt_weights(layers)
    conds = torch.from_numpy(g["conds"])
    text = torch.from_numpy(g["text"])
    emb, mask, pads = gpt_ref.prepare_gpt_inputs(conds, text, W)
    assert np.array_equal(mask.numpy().astype(np.int64), g["attention_mask"])
    np.testing.assert_allclose(emb.numpy(), g["prefix_emb"], atol=1e-6)
    steps = g["logits"].shape[0]
    logits, past = gpt_ref.decode_prefill(emb, mask, W)
    errs = [np.abs(logits.numpy() - g["logits"][0]).max()]
    codes = torch.from_numpy(g["codes"])
    for s in range(1, steps):
        mask = torch.cat([mask, torch.ones(mask.shape[0], 1, dtype=torch.bool)], dim=1)
        logits, past = gpt_ref.decode_step(codes[:, s - 1], s, mask, past, W)
        errs.append(np.abs(logits.numpy() - g["logits"][s]).max())
    assert max(errs) < 2e-4, errs
    # greedy codes with repetition penalty 10 over the fake prefix + generated history
    hist = g["fake_inputs"]
    for s in range(steps):
        sc = sampling_ref.repetition_penalty(g["logits"][s], hist, 10.0)
        assert np.array_equal(sampling_ref.greedy(sc), g["codes"][:, s])
        hist = np.concatenate([hist, g["codes"][:, s:s + 1]], axis=1)
    # latent pass
    n = int(g["text_lens"][0])
    lat = gpt_ref.latent_pass(conds, text[0, :n], codes[0], W)
    np.testing.assert_allclose(lat.numpy(), g["latent_row0"], atol=2e-4)


def test_left_pad_invariance_small():
    """tests/padding_test.py:69-97: a row decoded alone equals the same row inside a left-padded batch."""
    g = np.load(os.path.join(G, "gpt_small.npz"))
    W = gpt_weights(2)
    conds = torch.from_numpy(g["conds"])
    n = int(g["text_lens"][2])
    emb, mask, _ = gpt_ref.prepare_gpt_inputs(conds, torch.from_numpy(g["text"][2:3, :n]), W)
    logits, _ = gpt_ref.decode_prefill(emb, mask, W)
    np.testing.assert_allclose(logits.numpy(), g["logits_row2_alone_step0"], atol=2e-4)
    np.testing.assert_allclose(logits.numpy()[0], g["logits"][0][2], atol=2e-4)


def test_sampling_processors():
    g = np.load(os.path.join(G, "sampling.npz"))
    s = sampling_ref.repetition_penalty(g["logits"], g["history"], 10.0)
    np.testing.assert_array_equal(s, g["after_penalty"])
    s = sampling_ref.temperature(s, 0.8)
    np.testing.assert_allclose(s, g["after_temperature"], rtol=1e-6)
    s = sampling_ref.top_k(g["after_temperature"], 30)
    np.testing.assert_array_equal(s, g["after_topk"])
    s = sampling_ref.top_p(g["after_topk"], 0.8)
    np.testing.assert_array_equal(np.isfinite(s), np.isfinite(g["after_topp"]))
    np.testing.assert_array_equal(s[np.isfinite(s)], g["after_topp"][np.isfinite(g["after_topp"])])


def test_typical_warper_matches_transformers():
    """oracle/sampling_ref.typical against the installed transformers' TypicalLogitsWarper (the class model.py:704-708 puts in
    the processor list) on peaked and flat rows: the same tokens survive."""
    transformers = pytest.importorskip("transformers")
    from transformers.generation.logits_process import TypicalLogitsWarper
    rng = np.random.default_rng(5)
    for scale, mass, keep in ((1.0, 0.9, 1), (4.0, 0.9, 1), (0.3, 0.5, 2), (8.0, 0.95, 2)):
        x = (rng.standard_normal((6, 257)) * scale).astype(np.float32)
        want = TypicalLogitsWarper(mass=mass, min_tokens_to_keep=keep)(None, torch.from_numpy(x.copy())).numpy()
        got = sampling_ref.typical(x, mass, keep)
        np.testing.assert_array_equal(np.isfinite(got), np.isfinite(want))
        np.testing.assert_array_equal(got[np.isfinite(got)], want[np.isfinite(want)])


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert [int(x) for x in sampling_ref.philox4x32((0, 0, 0, 0), (0, 0))] == \
        [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(x) for x in sampling_ref.philox4x32((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2)] == \
        [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert [int(x) for x in sampling_ref.philox4x32((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344),
                                                    (0xA4093822, 0x299F31D0))] == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_activation1d():
    g = np.load(os.path.join(G, "act1d.npz"))
    np.testing.assert_allclose(bigvgan_ref.kaiser_sinc_filter(), g["up_filter"], atol=2e-8)
    np.testing.assert_allclose(bigvgan_ref.kaiser_sinc_filter(), g["down_filter"], atol=2e-8)
    i = 0
    while f"x{i}" in g:
        y = bigvgan_ref.activation1d(torch.from_numpy(g[f"x{i}"]), torch.from_numpy(g[f"alpha{i}"]),
                                     torch.from_numpy(g[f"beta{i}"]), g["up_filter"], g["down_filter"])
        np.testing.assert_allclose(y.numpy(), g[f"y{i}"], atol=2e-6)
        i += 1
    assert i == 6


def bigvgan_sd():
    import json
    shapes = json.load(open(os.path.join(G, "bigvgan_shapes.json")))
    return synth.fill_state_dict({k: tuple(v) for k, v in shapes.items()}, synth.bigvgan_param)


def test_bigvgan_forward():
    g = np.load(os.path.join(G, "bigvgan.npz"))
    W = bigvgan_ref.Weights(bigvgan_sd())
    taps = {}
    spk = torch.from_numpy(g["spk4"]).transpose(1, 2)
    wav = bigvgan_ref.forward(torch.from_numpy(g["latent4"]), spk, W, taps)
    np.testing.assert_allclose(taps["conv_pre"].numpy(), g["conv_pre4"], atol=1e-5)
    for i in range(6):
        np.testing.assert_allclose(taps[f"up{i}"].numpy(), g[f"up{i}_4"], atol=5e-5)
        np.testing.assert_allclose(taps[f"stage{i}"].numpy(), g[f"stage{i}_4"], atol=5e-5)
    np.testing.assert_allclose(wav.numpy(), g["wav4"], atol=2e-5)
    wav8 = bigvgan_ref.forward(torch.from_numpy(g["latent8"]), spk, W)
    assert np.sqrt(np.mean((wav8.numpy() - g["wav8"]) ** 2)) < 1e-5
    spk2 = torch.from_numpy(g["spk_b2"]).transpose(1, 2)
    wav2 = bigvgan_ref.forward(torch.from_numpy(g["latent_b2"]), spk2, W)
    assert np.sqrt(np.mean((wav2.numpy() - g["wav_b2"]) ** 2)) < 1e-5


def test_beam_oracle_matches_transformers_beam_search():
    """Pins oracle/beam_ref.py (beam search mode: scorer bookkeeping, EOS handling, length penalty, is_done heuristic,
    finalize, repetition penalty on log-probabilities) against the installed transformers' own `generate(num_beams=3,
    do_sample=False)` on a toy GPT-2 with seeded random weights.  (The reference pins transformers 4.44.2; the image has a
    newer release whose beam search is a vectorised rewrite of the same algorithm -- agreement on every case here is the
    evidence that the restatement is the published algorithm.  The sampling variant differs only in how the 2*num_beams
    candidates are drawn, which cannot be matched to torch.multinomial.)"""
    transformers = pytest.importorskip("transformers")
    import warnings

    from oracle import beam_ref
    torch.manual_seed(0)
    V, eos = 24, 23
    cfg = transformers.GPT2Config(vocab_size=V, n_positions=64, n_embd=16, n_layer=2, n_head=2, bos_token_id=0, eos_token_id=eos,
                                  pad_token_id=eos)
    class Boosted(transformers.GPT2LMHeadModel):
        boost = 0.0

        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            out.logits[..., eos] += self.boost   # seen by generate() and by the oracle's drive alike
            return out

    m = Boosted(cfg).eval()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.35)   # flat enough for the beams to compete
    n_eos = n_ranked = n_diff = 0
    for trial in range(24):
        m.boost = [0.0, 1.0, 1.5, 2.0, 2.5, 3.0][trial // 4]  # later cases: EOS is likely, hypotheses close, elements finish
        g = torch.Generator().manual_seed(100 + trial)
        B, P, nb, max_new = 2, 4, 3, 10
        prompt = torch.randint(1, V - 1, (B, P), generator=g)
        # length_penalty 0.0 is what infer.py passes (infer.py:807-814).  Above 1 the two transformers generations differ in
        # the early-stop heuristic (5.x compares the best RUNNING beam, 4.44.2 the best candidate of the step; oracle = 4.44.2)
        lp = [0.0, 1.0, 0.0, 0.5][trial % 4]
        rp = [1.0, 1.3][trial % 2]
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = m.generate(prompt, attention_mask=torch.ones_like(prompt), num_beams=nb, do_sample=False, max_new_tokens=max_new,
                             length_penalty=lp, repetition_penalty=rp, early_stopping=False, num_return_sequences=1,
                             pad_token_id=eos, eos_token_id=eos)
        hf = out[:, P:].numpy()
        sp = dict(do_sample=False, top_k=0, top_p=1.0, temperature=1.0, repetition_penalty=rp)
        for b in range(B):
            bs = beam_ref.BeamSearch(1, nb, sp, prompt[b].tolist(), eos=eos, length_penalty=lp)
            for _ in range(max_new):
                with torch.no_grad():
                    lg = m(torch.tensor(bs.hist[0])).logits[:, -1, :].float().numpy()
                bs.step(lg)
                if bs.all_done():
                    break
            ref, got = bs.finalize()[0].tolist(), hf[b].tolist()
            n = min(len(ref), len(got))
            assert ref[:n] == got[:n] and all(t == eos for t in ref[n:] + got[n:]), (trial, b, got, ref)
            n_eos += int(eos in got)
        # num_return_sequences > 1 (model.py:669, :711-714): the k best hypotheses of every element, best first
        for k in (2, 3):
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                outk = m.generate(prompt, attention_mask=torch.ones_like(prompt), num_beams=nb, do_sample=False, max_new_tokens=max_new,
                                  length_penalty=lp, repetition_penalty=rp, early_stopping=False, num_return_sequences=k,
                                  pad_token_id=eos, eos_token_id=eos)
            hfk = outk[:, P:].numpy()
            assert hfk.shape[0] == B * k
            for b in range(B):
                bs = beam_ref.BeamSearch(1, nb, sp, prompt[b].tolist(), eos=eos, length_penalty=lp)
                for _ in range(max_new):
                    with torch.no_grad():
                        lg = m(torch.tensor(bs.hist[0])).logits[:, -1, :].float().numpy()
                    bs.step(lg)
                    if bs.all_done():
                        break
                refk = bs.finalize(num_return=k)
                for j in range(k):
                    ref, got = refk[j].tolist(), hfk[b * k + j].tolist()
                    n = min(len(ref), len(got))
                    same = ref[:n] == got[:n] and all(t == eos for t in ref[n:] + got[n:])
                    n_ranked += 1
                    if not same:
                        # the one known divergence of the INSTALLED release (5.x) from 4.44.2's scorer: with length_penalty
                        # >= 1 it can keep a short EOS candidate that 4.44.2 never adds (beam_token_rank >= num_beams is
                        # skipped there) -- only ever the LAST kept hypothesis, never the returned best ones at infer.py's 0.0
                        assert lp >= 1.0 and j == k - 1 == nb - 1, (trial, k, b, j, lp, got, ref)
                        n_diff += 1
    assert n_ranked == 240 and n_diff <= 2, (n_ranked, n_diff)
    assert n_eos >= 4, "the EOS-heavy cases must actually close hypotheses"


def test_beam_sample_row_processing_matches_transformers_processors():
    """oracle/beam_ref.process_row (log-softmax -> repetition penalty -> temperature -> top-k -> top-p with
    min_tokens_to_keep = 2, the beam-sample chain) against the installed transformers processor classes."""
    transformers = pytest.importorskip("transformers")
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)

    from oracle import beam_ref
    rng = np.random.default_rng(21)
    V = 8194
    for case in range(6):
        logits = (rng.normal(size=(1, V)) * [1.0, 3.0, 6.0][case % 3]).astype(np.float32)
        hist = rng.integers(0, V, size=(1, 40))
        sp = dict(do_sample=True, top_k=[30, 5, 1][case % 3], top_p=[0.8, 0.3, 0.95][case % 3], temperature=[1.0, 0.7][case % 2],
                  repetition_penalty=[10.0, 1.0][case % 2])
        ids, s = torch.from_numpy(hist), torch.log_softmax(torch.from_numpy(logits), -1)
        if sp["repetition_penalty"] != 1.0:
            s = RepetitionPenaltyLogitsProcessor(sp["repetition_penalty"])(ids, s)
        if sp["temperature"] != 1.0:
            s = TemperatureLogitsWarper(sp["temperature"])(ids, s)
        s = TopKLogitsWarper(sp["top_k"], min_tokens_to_keep=2)(ids, s)
        s = TopPLogitsWarper(sp["top_p"], min_tokens_to_keep=2)(ids, s).numpy()[0]
        got = beam_ref.process_row(logits[0], hist[0], sp, 3)
        assert np.array_equal(np.isfinite(got), np.isfinite(s)), case
        np.testing.assert_allclose(got[np.isfinite(got)], s[np.isfinite(s)], rtol=2e-6, atol=2e-6)


def test_beam_sample_bookkeeping_matches_transformers_with_injected_draws(monkeypatch):
    """Pins the do_sample=True branch of oracle/beam_ref.py (processors with min_tokens_to_keep = 2 on log-probabilities,
    candidate pool over num_beams x V, scorer bookkeeping on DRAWN -- not top -- candidates, EOS handling, finalize) against
    the installed transformers' generate(num_beams=3, do_sample=True): the 2*num_beams candidate draws are INJECTED into both
    sides (torch.multinomial on the HF side, beam_ref.draw_without_replacement on the oracle side are replaced by the same
    rank-based picker over the same Philox numbers), so everything but the random stream itself is compared.
    The injected picks are handed over sorted by score: transformers 4.44.2 (the reference's pin) sorts the drawn
    candidates by score before BeamSearchScorer.process, the installed 5.x keeps draw order -- with sorted picks the two
    orders coincide, so the comparison is valid for the 4.44.2 semantics the oracle restates."""
    transformers = pytest.importorskip("transformers")
    import warnings

    from oracle import beam_ref
    torch.manual_seed(1)
    V, eos, nb = 40, 39, 3
    cfg = transformers.GPT2Config(vocab_size=V, n_positions=64, n_embd=16, n_layer=2, n_head=2, bos_token_id=0, eos_token_id=eos,
                                  pad_token_id=eos)

    class Boosted(transformers.GPT2LMHeadModel):
        boost = 0.0

        def forward(self, *a, **k):
            out = super().forward(*a, **k)
            out.logits[..., eos] += self.boost
            return out

    m = Boosted(cfg).eval()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.35)
    SEED = 4242
    step = {"k": 0}

    class SmallPool(Exception):
        pass

    def ranks(n_alive_start, n_draw, k):
        """positions (in the score-sorted pool) picked by draw 0..n_draw-1: uniform over the not-yet-drawn entries"""
        alive = list(range(n_alive_start))
        out = []
        for i in range(min(n_draw, n_alive_start)):
            u = float(beam_ref.uniform01(SEED, 0, k, i))
            out.append(alive.pop(min(int(u * len(alive)), len(alive) - 1)))
        return sorted(out)   # pool positions ascending = score descending: the order both sides then see

    def fake_multinomial(probs, num_samples, replacement=False, generator=None):
        assert probs.shape[0] == 1 and not replacement
        p = probs[0].double().numpy()
        ids = np.nonzero(p > 0)[0]
        order = ids[np.lexsort((ids, -p[ids]))]
        if order.size < num_samples:
            raise SmallPool()   # real torch.multinomial raises here too (cannot draw 2*num_beams without replacement)
        picks = ranks(order.size, num_samples, step["k"])
        step["k"] += 1
        return torch.from_numpy(order[picks].astype(np.int64))[None]

    def fake_draw(pool_scores, n_draw, seed, b, k):
        # entries whose weight exp(score - max) is 0 in fp32 (rows still at the -1e9 start score) are not candidates on
        # either side: torch's softmax gives them probability exactly 0
        e = np.exp((pool_scores - pool_scores[0]).astype(np.float32)).astype(np.float32)
        return ranks(int((e > 0).sum()), n_draw, k)

    monkeypatch.setattr(beam_ref, "draw_without_replacement", fake_draw)
    n_eos = n_nontop = n_ok = 0
    for trial in range(36):
        m.boost = [0.0, 1.0, 2.0, 3.0][trial // 9]
        g = torch.Generator().manual_seed(500 + trial)
        P, max_new = 4, 10
        prompt = torch.randint(1, V - 1, (1, P), generator=g)
        lp = [0.0, 1.0, 0.0, 0.5][trial % 4]
        rp = [1.0, 1.3][trial % 2]
        tk, tp, temp = [8, 12, 20][trial % 3], [0.8, 0.95, 1.0][trial % 3], [1.0, 0.7][trial % 2]
        step["k"] = 0
        with torch.no_grad(), warnings.catch_warnings(), monkeypatch.context() as mp:
            warnings.simplefilter("ignore")
            mp.setattr(torch, "multinomial", fake_multinomial)
            try:
                out = m.generate(prompt, attention_mask=torch.ones_like(prompt), num_beams=nb, do_sample=True, top_k=tk, top_p=tp,
                                 temperature=temp, max_new_tokens=max_new, length_penalty=lp, repetition_penalty=rp,
                                 early_stopping=False, num_return_sequences=1, pad_token_id=eos, eos_token_id=eos)
            except SmallPool:
                continue   # top-p left fewer than 2*num_beams candidates: generate() cannot run this case at all
        n_ok += 1
        hf = out[0, P:].tolist()
        sp = dict(do_sample=True, top_k=tk, top_p=tp, temperature=temp, repetition_penalty=rp)
        bs = beam_ref.BeamSearch(1, nb, sp, prompt[0].tolist(), eos=eos, length_penalty=lp, seed=SEED)
        for _ in range(max_new):
            with torch.no_grad():
                lg = m(torch.tensor(bs.hist[0])).logits[:, -1, :].float().numpy()
            bs.step(lg)
            if bs.all_done():
                break
        ref = bs.finalize()[0].tolist()
        n = min(len(ref), len(hf))
        assert ref[:n] == hf[:n] and all(t == eos for t in ref[n:] + hf[n:]), (trial, hf, ref)
        n_eos += int(eos in hf)
        # the drawn candidates must not simply be the top ones, or this would only repeat the beam-search test
        with torch.no_grad():
            greedy = m.generate(prompt, attention_mask=torch.ones_like(prompt), num_beams=nb, do_sample=False, max_new_tokens=max_new,
                                length_penalty=lp, repetition_penalty=rp, early_stopping=False, pad_token_id=eos, eos_token_id=eos)
        n_nontop += int(greedy[0, P:].tolist()[:n] != hf[:n])
    assert n_ok >= 20 and n_eos >= 4 and n_nontop >= 8, (n_ok, n_eos, n_nontop)
