"""End-to-end parity of the HIP engines against the reference-run fixtures AND the oracle (same seeded weights).

north_star tolerances: <= 1e-3 max-abs on mel-code logits, <= 1e-4 RMS on the waveform, fp32 (greedy decode)."""
import os

import numpy as np
import pytest
import torch

import synth
import weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def make_gpt(layers, dtype):
    from indextts.gpt.model import UnifiedVoice
    cfg = weights.reference_config()["gpt"]
    cfg = dict(cfg, layers=layers)
    m = UnifiedVoice(**cfg)
    m.load_state_dict(weights.gpt_state_dict(layers))
    m.to(DEV).to(dtype)
    m.post_init_gpt2_config(kv_cache=True)
    return m


@pytest.fixture(scope="module")
def gpt_small_fp32():
    return make_gpt(2, torch.float32)


@pytest.mark.parametrize("tag,layers", [("gpt_small", 2), ("gpt_full", 24)])
def test_gpt_decode_logits_and_latent_fp32(tag, layers, gpt_small_fp32):
    g = np.load(os.path.join(G, tag + ".npz"))
    m = gpt_small_fp32 if layers == 2 else make_gpt(layers, torch.float32)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, torch.tensor([120], device=DEV))
    assert (conds.cpu() - torch.from_numpy(g["conds"])).abs().max().item() < 1e-3
    text = torch.from_numpy(g["text"]).to(DEV)
    fake, emb, mask = m.prepare_gpt_inputs(conds, text)
    assert np.array_equal(mask.cpu().numpy(), g["attention_mask"])
    assert (emb.cpu() - torch.from_numpy(g["prefix_emb"])).abs().max().item() < 1e-3
    # teacher-force the reference's greedy codes and compare raw logits at every step
    steps = g["logits"].shape[0]
    eng = m.engine
    pad = (mask == 0).sum(1).to(torch.int32)
    logits = eng.prefill(emb, pad, steps + 2)
    errs = [(logits.cpu() - torch.from_numpy(g["logits"][0])).abs().max().item()]
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)
    got_codes = []
    for s in range(1, steps):
        eng._sample(3, sp)
        got_codes.append(eng.tokens[:3].cpu().numpy().copy())
        eng.tokens[:3] = torch.from_numpy(g["codes"][:, s - 1]).to(torch.int32).to(DEV)  # teacher forcing
        eng.history[:3, s - 1] = eng.tokens[:3]
        # the sample kernel already advanced state; run the transformer part of the step only
        eng._step_transformer(3)
        errs.append((eng.logits[:3].cpu() - torch.from_numpy(g["logits"][s])).abs().max().item())
    assert max(errs) < 1e-3, errs
    # greedy choices equal the reference's wherever the decision margin (on the repetition-penalised scores the greedy
    # rule sees: ids 1, 8192 and the row's earlier codes divided / multiplied by 10) exceeds twice the logit tolerance
    decided = 0
    for s, c in enumerate(got_codes):
        for b in range(3):
            sc = torch.from_numpy(g["logits"][s][b].copy())
            ids = torch.tensor(sorted({1, 8192} | {int(v) for v in g["codes"][b, :s]}))
            sc[ids] = torch.where(sc[ids] < 0, sc[ids] * 10.0, sc[ids] / 10.0)
            top2 = torch.topk(sc, 2)
            if (top2.values[0] - top2.values[1]).item() > 2e-3:
                assert int(c[b]) == int(g["codes"][b, s]) == int(top2.indices[0]), (s, b, c, g["codes"][:, s])
                decided += 1
    assert decided >= 2 * len(got_codes), decided   # the margin rule must not make the check vacuous
    # latent pass (row 0)
    n = int(g["text_lens"][0])
    lat = m(cond_mel, text[0:1, :n], torch.tensor([n]), torch.from_numpy(g["codes"][0:1]).to(DEV),
            torch.tensor([steps * 1024]), cond_mel_lengths=torch.tensor([120], device=DEV), return_latent=True)
    assert (lat.cpu() - torch.from_numpy(g["latent_row0"])).abs().max().item() < 1e-3


def test_decode_loop_graph_equals_eager_and_pad_invariance(gpt_small_fp32):
    """(i) CUDA-graph replay == eager launches, bit for bit; (ii) tests/padding_test.py:69-97: a row decoded alone gives
    the same greedy codes as inside a left-padded batch (compared where the top-2 margin exceeds 1e-4)."""
    m = gpt_small_fp32
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    kw = dict(do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=24)
    m.engine._graphs.clear()
    codes_a, logits_a = m.inference_speech(cond_mel, text, return_logits=True, **kw)
    eng = m.engine
    conds = m.get_conditioning(cond_mel, None)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    eng.prefill(emb, (mask == 0).sum(1).to(torch.int32), 24)
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)
    codes_b, logits_b = eng.decode(24, sp, use_graph=False, return_logits=True)
    assert torch.equal(codes_a, codes_b)
    assert torch.equal(logits_a, logits_b)
    n2 = int(g["text_lens"][2])
    codes_1, logits_1 = m.inference_speech(cond_mel, text[2:3, :n2], return_logits=True, **kw)
    la, l1 = logits_a[:, 2].cpu(), logits_1[:, 0].cpu()
    same_so_far = True
    for s in range(codes_1.shape[1]):
        if not same_so_far:
            break
        assert (la[s] - l1[s]).abs().max().item() < 1e-3
        top2 = torch.topk(l1[s], 2).values
        if (top2[0] - top2[1]).item() > 1e-3:
            assert codes_a[2, s].item() == codes_1[0, s].item()
        same_so_far = codes_a[2, s].item() == codes_1[0, s].item()


def test_sampling_loop_is_reproducible_and_stops(gpt_small_fp32):
    m = gpt_small_fp32
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, num_beams=1, repetition_penalty=10.0,
              max_generate_length=40, force_stop=[10, 25, 33])
    a = m.inference_speech(cond_mel, text, seed=7, **kw)
    b = m.inference_speech(cond_mel, text, seed=7, **kw)
    c = m.inference_speech(cond_mel, text, seed=8, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    a = a.cpu().numpy()
    assert a.shape[1] <= 40
    for row, stop in zip(a, (10, 25, 33)):
        assert (row[:stop] != 8193).all() and (row[stop:] == 8193).all()


def test_finished_rows_left_out_of_attention_change_nothing_else(gpt_small_fp32):
    """Rows past their stop token are skipped by the decode attention (engine.skip_finished, commit 4cc617f).  With and
    without the skip: identical codes for every row, identical logits for every row up to and including the step that
    produces its stop token; later logits of a finished row are documented as undefined."""
    m = gpt_small_fp32
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    stops = [6, 14, 21]
    kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, num_beams=1, repetition_penalty=10.0,
              max_generate_length=24, force_stop=stops, seed=5, return_logits=True)
    outs = []
    for skip in (True, False):
        m.engine.skip_finished = skip
        m.engine._graphs.clear()
        outs.append(m.inference_speech(cond_mel, text, **kw))
    m.engine.skip_finished = True
    m.engine._graphs.clear()
    (c0, l0), (c1, l1) = outs
    assert torch.equal(c0, c1)
    for r, stop in enumerate(stops):
        assert torch.equal(l0[: stop + 1, r], l1[: stop + 1, r]), f"row {r}: logits before its stop must not depend on the skip"
    assert not torch.equal(l0[stops[0] + 2:, 0], l1[stops[0] + 2:, 0]), "the skip must actually have been taken"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_latent_pass_reuses_the_cached_prompt(dtype):
    """GPTEngine.latent_mel_rows (only the mel rows recomputed, the prompt's keys / values taken from the KV cache the decode
    loop leaves behind) against the full teacher-forced pass over cond | text | mel: the SAME bits, for a left-padded batch of
    mixed text and code lengths, after a real decode (which appends behind the prompt), and through IndexTTS._latents."""
    from indextts.infer import IndexTTS
    cfg = weights.reference_config()
    cfg["gpt"]["layers"] = 2
    tts = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device="cuda:0",
                                precision_config={"gpt": "fp32" if dtype == torch.float32 else "bf16", "vocoder": "fp16"})
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    rng = np.random.default_rng(4)
    texts = [torch.from_numpy(rng.integers(2, 12000, size=int(n))).to(torch.int32) for n in (5, 17, 9, 12)]
    gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
    st = tts._batch_tokens(cond_mel, texts, max_mel_tokens=14, force_stop=[13, 7, 10, 4], seed=3, **gen)
    assert [int(r.numel()) for r in st["rows"]] == [13, 7, 10, 4]
    full = tts._latents(st["conds"], st["texts"], st["rows"], reuse_prefix=False)
    fast = tts._latents(st["conds"], st["texts"], st["rows"], reuse_prefix=True)
    assert len(full) == len(fast) == 4
    for a, b in zip(full, fast):
        assert a.shape == b.shape and torch.equal(a, b)
    # beams: the prompt is cached once per element (row table) or in every beam's row (copy): both are found
    for kv in ("table", "copy"):
        tts.gpt.engine.beam_kv = kv
        stb = tts._batch_tokens(cond_mel, texts[:2], max_mel_tokens=8, seed=3, **dict(gen, num_beams=3))
        fb = tts._latents(stb["conds"], stb["texts"], stb["rows"], reuse_prefix=False)
        rb = tts._latents(stb["conds"], stb["texts"], stb["rows"], reuse_prefix=True, cache_rows=stb["cache_rows"])
        for a, b in zip(fb, rb):
            assert torch.equal(a, b), kv
    tts.gpt.engine.beam_kv = "table"
    # the switch: with reuse off the same call takes the full pass
    tts.reuse_prompt_kv = False
    off = tts._latents(st["conds"], st["texts"], st["rows"], reuse_prefix=True)
    assert all(torch.equal(a, b) for a, b in zip(full, off))


def test_typical_sampling_is_refused(gpt_small_fp32):
    """model.py:704-708's optional warper is never enabled by infer.py / cli.py / api.py (out of
    scope) and is not built: asking for it fails loudly instead of sampling from another distribution."""
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    with pytest.raises(NotImplementedError):
        gpt_small_fp32.inference_speech(cond_mel, torch.from_numpy(g["text"]).to(DEV), typical_sampling=True, max_generate_length=2)


def test_gpt_bf16_tracks_fp32():
    g = np.load(os.path.join(G, "gpt_small.npz"))
    m = make_gpt(2, torch.bfloat16)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    logits = m.engine.prefill(emb, (mask == 0).sum(1).to(torch.int32), 8)
    err = (logits.cpu() - torch.from_numpy(g["logits"][0])).abs().max().item()
    assert err < 0.15, err  # bf16 weights/activations, fp32 accumulation and residual stream


def make_vocoder(dtype):
    from indextts.BigVGAN.models import BigVGAN
    from indextts.utils.config import Config
    v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
    v.load_state_dict(weights.bigvgan_state_dict())
    v.to(DEV).to(dtype).remove_weight_norm()
    return v


def test_bigvgan_fp32_matches_reference():
    g = np.load(os.path.join(G, "bigvgan.npz"))
    v = make_vocoder(torch.float32)
    taps = {}
    wav, _ = v(torch.from_numpy(g["latent4"]).to(DEV), torch.from_numpy(g["melref"]).to(DEV), taps=taps)
    spk = v.speaker_embedding(torch.from_numpy(g["melref"]).to(DEV))
    assert (spk.cpu() - torch.from_numpy(g["spk4"])).abs().max().item() < 1e-4
    for i in range(6):
        ref = torch.from_numpy(g[f"stage{i}_4"]).transpose(1, 2)
        assert (taps[f"stage{i}"].cpu() - ref).abs().max().item() < 2e-4, f"stage {i}"
    assert (wav.cpu() - torch.from_numpy(g["wav4"])).abs().max().item() < 1e-4
    for lat, mel, key in (("latent8", "melref", "wav8"), ("latent_b2", "melref_b2", "wav_b2")):
        w, _ = v(torch.from_numpy(g[lat]).to(DEV), torch.from_numpy(g[mel]).to(DEV))
        rms = (w.cpu() - torch.from_numpy(g[key])).pow(2).mean().sqrt().item()
        assert rms < 1e-4, (key, rms)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
def test_bigvgan_half_tracks_fp32(dtype, tol):
    g = np.load(os.path.join(G, "bigvgan.npz"))
    v = make_vocoder(dtype)
    w, _ = v(torch.from_numpy(g["latent8"]).to(DEV), torch.from_numpy(g["melref"]).to(DEV))
    rms = (w.cpu() - torch.from_numpy(g["wav8"])).pow(2).mean().sqrt().item()
    assert rms < tol, rms
    # the measurement hook of bench.py's per-stage roofline: one entry per stage boundary, the same waveform, and the work
    # it accounts for is the generator's 3.01 GFLOP per frame within the halo rows of the 2-tap upsamplers
    prof = []
    lat = torch.from_numpy(g["latent8"]).to(DEV)
    w2, _ = v(lat, torch.from_numpy(g["melref"]).to(DEV), profile=prof)
    assert torch.equal(w, w2)
    assert [e[0] for e in prof][0] == "start" and len(prof) == 2 + len(v.rates) + 1
    torch.cuda.synchronize()
    assert all(a[1].elapsed_time(b[1]) >= 0.0 for a, b in zip(prof[:-1], prof[1:]))
    frames = lat.shape[0] * lat.shape[1]
    gflop_per_frame = sum(e[2] for e in prof) / frames / 1e9
    assert 2.9 < gflop_per_frame < 3.4, gflop_per_frame
    assert all(e[3] > 0 for e in prof[1:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_bigvgan_ragged_batch_equals_single_runs(dtype):
    """BigVGAN.forward(lens=...): utterances of different lengths in one batch give, sample for sample, what vocoding
    each of them alone gives (the reference vocodes them one by one, infer.py:885-899)."""
    g = np.load(os.path.join(G, "bigvgan.npz"))
    v = make_vocoder(dtype)
    lat = torch.from_numpy(g["latent8"]).to(DEV)             # [B, 8, D]
    lat = torch.cat([lat, lat.flip(1), lat * 0.5], 0)[:5]
    mel = torch.from_numpy(g["melref"]).to(DEV)
    spk = v.speaker_embedding(mel[:1])
    lens = [8, 3, 1, 5, 8][: lat.shape[0]]
    wav, _ = v(lat, speaker_embedding=spk, lens=lens)
    hop = wav.shape[-1] // lat.shape[1]
    for b, n in enumerate(lens):
        w1, _ = v(lat[b:b + 1, :n].contiguous(), speaker_embedding=spk)
        assert torch.equal(wav[b, :, : n * hop], w1[0]), (b, n)
    with pytest.raises(ValueError):
        v(lat, speaker_embedding=spk, lens=[9] * lat.shape[0])


def test_batch_pipeline_equals_serial_infer_batch():
    """BatchPipeline (stage B on a second stream beside the next batch's token loop) returns exactly what
    infer_batch returns batch by batch: same kernels, same inputs, no shared mutable state between the stages."""
    from indextts.infer import BatchPipeline, IndexTTS
    cfg = weights.reference_config()
    cfg["gpt"]["layers"] = 2
    tts = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device="cuda:0",
                                precision_config={"gpt": "bf16", "vocoder": "fp16"})
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    rng = np.random.default_rng(5)
    batches = [[torch.from_numpy(rng.integers(2, 12000, size=int(n))).to(torch.int32) for n in rng.integers(5, 20, size=4)]
               for _ in range(3)]
    gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
    kw = dict(max_mel_tokens=13, force_stop=[12, 9, 12, 7])
    serial = [tts.infer_batch(cond_mel, b, seed=100 + i, return_codes=True, **kw, **gen) for i, b in enumerate(batches)]
    torch.cuda.synchronize()
    pipe = BatchPipeline(tts)
    tickets = [pipe.submit(cond_mel, b, seed=100 + i, **kw, **gen) for i, b in enumerate(batches)]
    for (wavs, rows), t in zip(serial, tickets):
        got = t.result()
        assert [r.tolist() for r in rows] == [r.tolist() for r in t.rows]
        assert len(got) == len(wavs)
        for a, b in zip(wavs, got):
            assert a.shape == b.shape and torch.equal(a, b)
    pipe.drain()
    pipe.close()


@pytest.mark.parametrize("do_sample", [False, True])
def test_beam_decode_matches_host_driven_oracle(gpt_small_fp32, do_sample):
    """engine.decode_beam (device-side scorer, in-place KV permutation, graph replay) against a loop that runs the same
    transformer steps but does the beam bookkeeping with oracle/beam_ref.py on the host and permutes the cache with
    index_select -- same best hypotheses."""
    from oracle import beam_ref
    m = gpt_small_fp32
    eng = m.engine
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, torch.tensor([120], device=DEV))
    text = torch.tensor([[11, 22, 33, 44, 55, 66], [77, 88, 99, 1, 1, 1]], device=DEV)
    B, nb, max_new = 2, 3, 12
    R = B * nb
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    pad = (mask == 0).sum(1).to(torch.int32)
    sp = dict(do_sample=do_sample, top_p=0.8 if do_sample else 1.0, top_k=30 if do_sample else 0, temperature=1.0,
              repetition_penalty=10.0, seed=5, length_penalty=0.0)
    emb_r, pad_r = emb.repeat_interleave(nb, 0), pad.repeat_interleave(nb)
    # host-driven oracle loop (beam search works on whole contiguous cache rows: paged=False)
    eng.prefill(emb_r, pad_r, max_new, paged=False)
    ref = beam_ref.BeamSearch(B, nb, sp, [1] * int(emb.shape[1]) + [8192], eos=8193, length_penalty=0.0, seed=5)
    n = 0
    while True:
        tok, src = ref.step(eng.logits[:R].cpu().numpy())
        n += 1
        if n >= max_new or ref.all_done():
            break
        idx = torch.from_numpy(src).to(DEV)
        eng.kc[:, :R] = eng.kc[:, idx]
        eng.vc[:, :R] = eng.vc[:, idx]
        eng.tokens[:R] = torch.from_numpy(tok).to(torch.int32).to(DEV)
        eng.state[0] += 1   # what the select kernel does: step and cache position advance
        eng.state[1] += 1
        eng._step_transformer(R)
    want = ref.finalize()
    # "table": KV rows follow their beams through the row table the attention kernel reads (no cache bytes move);
    # "copy": the cache rows are permuted in place (itts_beam_reorder_kv, the reference form of _reorder_cache)
    for kv in ("table", "copy"):
        for use_graph in (False, True):
            eng.beam_kv = kv
            eng.prefill(emb_r, pad_r, max_new, paged=False)
            got = eng.decode_beam(max_new, sp, nb, use_graph=use_graph, check_every=4).cpu().numpy()
            w = min(got.shape[1], want.shape[1])
            assert np.array_equal(got[:, :w], want[:, :w]), (kv, use_graph, got, want)
            assert (got[:, w:] == 8193).all() and (want[:, w:] == 8193).all()
    eng.beam_kv = "table"
    # shared prefix: the prompt computed and cached ONCE per batch element, all its beams' table entries point at it
    for use_graph in (False, True):
        eng.prefill(emb, pad, max_new, beams=nb)
        got = eng.decode_beam(max_new, sp, nb, use_graph=use_graph, check_every=4).cpu().numpy()
        w = min(got.shape[1], want.shape[1])
        assert np.array_equal(got[:, :w], want[:, :w]), ("shared", use_graph, got, want)
    with pytest.raises(ValueError):
        eng.prefill(emb, pad, max_new, beams=nb)
        eng.decode_beam(max_new, sp, nb + 1)
    if eng.paged:
        with pytest.raises(ValueError):          # a paged cache cannot be beam-searched: refused, not silently wrong
            eng.prefill(emb_r, pad_r, max_new)
            eng.decode_beam(max_new, sp, nb)
    eng.beam_kv = "copy"
    with pytest.raises(ValueError):
        eng.prefill(emb, pad, max_new, beams=nb)      # sharing the prompt needs the table
    eng.beam_kv = "table"
    # the reference-API entry point takes the same route
    codes = m.inference_speech(cond_mel, text, do_sample=do_sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0,
                               repetition_penalty=10.0, length_penalty=0.0, max_generate_length=max_new, seed=5)
    assert np.array_equal(codes.cpu().numpy()[:, :w], want[:, :w])
    # num_return_sequences > 1 (model.py:669, :711-714): the k best hypotheses of every element, best first, from the device
    # hypothesis store -- against the oracle's finalize(num_return=k), itself pinned against transformers' generate()
    for k in (2, 3):
        wk = ref.finalize(num_return=k)
        ck = m.inference_speech(cond_mel, text, do_sample=do_sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0,
                                repetition_penalty=10.0, length_penalty=0.0, max_generate_length=max_new, seed=5,
                                num_return_sequences=k).cpu().numpy()
        assert ck.shape[0] == B * k
        ww = min(ck.shape[1], wk.shape[1])
        assert np.array_equal(ck[:, :ww], wk[:, :ww]) and (ck[:, ww:] == 8193).all() and (wk[:, ww:] == 8193).all()
        assert np.array_equal(ck[::k, :w], want[:, :w])          # rank 0 is the single-sequence answer
    with pytest.raises(ValueError):
        m.inference_speech(cond_mel, text, num_beams=nb, num_return_sequences=nb + 1, max_generate_length=4)
    with pytest.raises(ValueError):
        m.inference_speech(cond_mel, text, do_sample=False, num_beams=1, num_return_sequences=2, max_generate_length=4)
    if do_sample:   # plain sampling: k independent draws per element (rows expanded before the first forward)
        one = m.inference_speech(cond_mel, text.repeat_interleave(2, 0), do_sample=True, num_beams=1, top_k=30, top_p=0.8,
                                 repetition_penalty=10.0, max_generate_length=8, seed=9)
        two = m.inference_speech(cond_mel, text, do_sample=True, num_beams=1, top_k=30, top_p=0.8, repetition_penalty=10.0,
                                 max_generate_length=8, seed=9, num_return_sequences=2)
        assert two.shape[0] == 4 and torch.equal(one, two) and not torch.equal(two[0], two[1])


def test_public_api_infer_and_infer_fast_write_wavs(tmp_path):
    """BASELINE config 1 ("plumbing"): a 44.1 kHz stereo prompt wav + CJK text through IndexTTS.infer and .infer_fast with
    the reference's default generation settings (beam-sample, num_beams=3) -> a 24 kHz PCM-16 wav of 1024 samples per
    acoustic token (infer.py:892-917); output_path=None returns (24000, int16 [N, 1])."""
    import wave

    from indextts.infer import IndexTTS
    from indextts.utils.audio import write_pcm16
    cfg = weights.reference_config()
    cfg["gpt"]["layers"] = 2
    tts = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device="cuda:0",
                                precision_config={"gpt": "bf16", "vocoder": "fp16"})
    t = np.arange(int(44100 * 1.2)) / 44100.0
    stereo = np.stack([0.3 * np.sin(2 * np.pi * 220 * t), 0.2 * np.sin(2 * np.pi * 330 * t)], 1)
    prompt = str(tmp_path / "prompt.wav")
    write_pcm16(prompt, (stereo * 32767).astype(np.int16), 44100)
    text = "你好世界，今天天气很好。我们去公园散步吧！"
    out = str(tmp_path / "gen.wav")
    with pytest.warns(RuntimeWarning):  # random weights never emit the stop token: the max_mel_tokens warning of infer.py:850
        ret = tts.infer(prompt, text, out, max_mel_tokens=9)
    assert ret == out
    with wave.open(out, "rb") as w:
        assert (w.getframerate(), w.getnchannels(), w.getsampwidth()) == (24000, 1, 2)
        n = w.getnframes()
    assert n > 0 and n % 1024 == 0
    with pytest.warns(RuntimeWarning):
        sr, pcm = tts.infer_fast(prompt, text, None, max_mel_tokens=9, max_text_tokens_per_sentence=8, num_beams=1, do_sample=False)
    assert sr == 24000 and pcm.dtype == np.int16 and pcm.ndim == 2 and pcm.shape[1] == 1 and pcm.shape[0] % 1024 == 0
    with pytest.raises(ValueError):
        tts.infer(prompt, text, None, speaker_id="nobody")


def test_request_pool_matches_serial_infer_batch():
    """RequestPool (two replicas, two threads, two streams, requests overlapping on the GPU) returns what infer_batch
    returns for the same (batch, seed) on one instance."""
    from indextts.infer import IndexTTS, RequestPool
    cfg = weights.reference_config()
    cfg["gpt"]["layers"] = 2
    first = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device="cuda:0",
                                  precision_config={"gpt": "bf16", "vocoder": "fp16"})
    insts = [first, first.replica()]          # shared weights, private KV cache / state / graphs
    e0, e1 = first.gpt.engine, insts[1].gpt.engine
    assert e1 is not e0 and e1.layers is not e0.layers                 # own per-layer dicts (adapter state is per engine) ...
    assert all(a[k] is b[k] for a, b in zip(e0.layers, e1.layers) for k in a)   # ... over the SAME weight tensors
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    rng = np.random.default_rng(9)
    batches = [[torch.from_numpy(rng.integers(2, 12000, size=int(n))).to(torch.int32) for n in rng.integers(5, 20, size=4)]
               for _ in range(5)]
    gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
    kw = dict(max_mel_tokens=11, force_stop=[10, 8, 10, 6])
    serial = [insts[0].infer_batch(cond_mel, b, seed=300 + i, **kw, **gen) for i, b in enumerate(batches)]
    torch.cuda.synchronize()
    pool = RequestPool(insts)
    pool.warm_up(cond_mel, batches[0], seed=1, **kw, **gen)
    jobs = [pool.submit(cond_mel, b, seed=300 + i, **kw, **gen) for i, b in enumerate(batches)]
    for want, job in zip(serial, jobs):
        got = job.result()
        assert len(got) == len(want)
        for a, b in zip(want, got):
            assert a.shape == b.shape and torch.equal(a, b)
    pool.close()


@pytest.mark.parametrize("lens", [[17, 3, 9, 12, 1], [3, 12, 17, 1, 9]])   # element 0 the longest (pad 0) / not (pad 14)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prefill_computes_the_shared_conditioning_rows_once(dtype, lens):
    """prefill(shared_rows=32): the 32 conditioning latents every element of a one-prompt batch starts with go through the
    blocks once, each element contributes only its text rows, their attention reads the shared block out of the same qkv
    buffer.  Logits, the whole KV cache region the decode loop reads, and the decoded tokens are those of the plain packed
    prefill, bit for bit -- mixed text lengths (different left paddings), also through the beam prefill (row table)."""
    m = gpt_small_fp32_or(dtype)
    eng = m.engine
    rng = np.random.default_rng(31)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    L = max(lens)
    text = torch.full((len(lens), L), m.stop_text_token, dtype=torch.int32)
    for i, n in enumerate(lens):
        text[i, :n] = torch.from_numpy(rng.integers(2, 12000, size=n)).to(torch.int32)
    _, emb, mask = m.prepare_gpt_inputs(conds, text.to(DEV))
    pad = (mask == 0).sum(1).to(torch.int32)
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)
    B, S = emb.shape[0], emb.shape[1] + 1
    got = {}
    for shared in (0, 32):
        eng._cap_b = eng._cap_s = 0                       # fresh (zeroed) caches for both runs
        logits = eng.prefill(emb, pad, 12, shared_rows=shared).clone()
        kc, vc = eng.dense_kv(B, S)                       # positions [0, S) of every row, through the block table if paged
        for b_, p_ in enumerate(pad.tolist()):            # (positions in front of a row's window map to the scratch block)
            kc[:, b_, :, :p_] = 0
            vc[:, b_, :, :p_] = 0
        codes = eng.decode(12, sp)
        got[shared] = (logits, kc, vc, codes.clone())
    assert eng.share_prefix and eng.share_kv_reads
    assert int(eng.kv_share.item()) == ((int(pad[0]) << 8) | 32)     # the decode attention was told where row 0 keeps the block
    for a, b in zip(got[0], got[32]):
        assert torch.equal(a, b)
    for b_, p_ in enumerate(pad.tolist()):                # (the comparison above is not vacuous: the prompt region is filled)
        assert got[32][1][:, b_, :, p_:S].abs().sum().item() > 0 and got[32][1][:, b_, :, :p_].abs().sum().item() == 0
    # beam prefill (prompt cached once per element, rows expanded afterwards) on top of it
    outs = []
    for shared in (0, 32):
        eng.prefill(emb[:2], pad[:2], 10, beams=3, shared_rows=shared)
        outs.append(eng.decode_beam(10, dict(sp, do_sample=True, top_k=30, top_p=0.8, length_penalty=0.0, seed=4), 3).clone())
    assert torch.equal(outs[0], outs[1])


def gpt_small_fp32_or(dtype):
    return make_gpt(2, dtype)


def test_decode_refill_gives_every_row_the_codes_it_gets_alone(gpt_small_fp32):
    """Continuous batching (GPTEngine.decode_refill; refills joined at once or staged on a second stream): 9 utterances of different text lengths and stop steps through 3 decode
    slots, greedy.  Every utterance's codes equal those of decoding it ALONE (inference_speech on one row) -- compared up to
    the first step whose top-2 margin in the stand-alone logits is below 1e-3 (a different left padding changes the
    attention's reduction order) -- although it entered a running loop, at a shifted cache position, in a slot another
    utterance had used, with its own clock for mel positions / history / stop step.  Graph replay and eager agree."""
    m = gpt_small_fp32
    eng = m.engine
    rng = np.random.default_rng(21)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    lens = [14, 12, 11, 9, 8, 7, 6, 5, 4]                       # longest first
    texts = [torch.from_numpy(rng.integers(2, 12000, size=n)).to(torch.int32) for n in lens]
    stops = [9, 21, 5, 13, 30, 3, 17, 8, 11]
    max_new = 40
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)

    def prefix(ids):
        L = max(lens[i] for i in ids)
        bh = torch.full((len(ids), L), m.stop_text_token, dtype=torch.int32)
        for j, i in enumerate(ids):
            bh[j, : lens[i]] = texts[i]
        _, emb, mask = m.prepare_gpt_inputs(conds, bh.to(DEV))
        return emb, (mask == 0).sum(1).to(torch.int32)

    alone = []
    for i in range(9):
        c, lg = m.inference_speech(cond_mel, texts[i][None].to(DEV), do_sample=False, num_beams=1, repetition_penalty=10.0,
                                   max_generate_length=max_new, force_stop=[stops[i]], return_logits=True)
        alone.append((c[0].cpu(), lg[:, 0].cpu()))
    got = {}
    for use_graph, staged in ((False, False), (True, False), (True, True)):
        queue = list(range(3, 9))
        emb, pad = prefix([0, 1, 2])
        eng.prefill(emb, pad, 400)

        def feed(k):
            take = [queue.pop(0) for _ in range(min(k, len(queue)))]
            if not take:
                return []
            e, p = prefix(take)
            return [(e[j, int(p[j]):], stops[i]) for j, i in enumerate(take)]

        codes, leftover = eng.decode_refill(max_new, sp, feed, force_stop=stops[:3], use_graph=use_graph, check_every=4,
                                            staged=staged)
        assert not leftover and len(codes) == 9
        assert eng.refill_stats["rows_refilled"] == 6 and eng.refill_stats["staged"] == staged
        got[(use_graph, staged)] = [c.cpu() for c in codes]
    for a, b in zip(got[(False, False)], got[(True, False)]):
        assert torch.equal(a, b)
    # (a staged refill joins one poll later, at another cache position: same utterance ids, same codes)
    for mode in ((True, False), (True, True)):
        for i, c in enumerate(got[mode]):
            want, lg = alone[i]
            assert int(c[-1]) == m.stop_mel_token and c.numel() == stops[i] + 1, (mode, i, c)
            for s_ in range(c.numel()):
                if int(c[s_]) != int(want[s_]):
                    top2 = torch.topk(lg[s_], 2).values
                    assert (top2[0] - top2[1]).item() < 1e-3, (mode, i, s_, c, want)
                    break
    # the cache positions reserved by prefill() bound the loop: what no longer fits comes back as leftover
    queue = list(range(3, 9))
    emb, pad = prefix([0, 1, 2])
    eng.prefill(emb, pad, 400)
    codes, leftover = eng.decode_refill(max_new, sp, feed, force_stop=stops[:3], check_every=4, positions=128)
    assert len(codes) + len(leftover) + len(queue) == 9 and len(leftover) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_paged_kv_cache_gives_the_bits_of_the_contiguous_cache(dtype):
    """The paged KV cache (block pool + per-row block table, the default for num_beams = 1) against contiguous cache rows on
    the same engine: prefill logits, every decode step's logits and codes, the cache contents and the latent pass that reads
    the prompt's keys / values back out of the cache are IDENTICAL BITS -- paging changes where a position lives, nothing
    else.  Mixed text lengths (different left paddings), shared conditioning rows, graph replay."""
    m = make_gpt(2, dtype)
    eng = m.engine
    rng = np.random.default_rng(77)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    lens = [17, 3, 9, 12, 1]
    text = torch.full((len(lens), max(lens)), m.stop_text_token, dtype=torch.int32)
    for i, n in enumerate(lens):
        text[i, :n] = torch.from_numpy(rng.integers(2, 12000, size=n)).to(torch.int32)
    _, emb, mask = m.prepare_gpt_inputs(conds, text.to(DEV))
    pad = (mask == 0).sum(1).to(torch.int32)
    B, S = emb.shape[0], emb.shape[1] + 1
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=21)
    runs = {}
    for paged in (False, True):
        eng.paged = paged
        eng._cap_b = eng._cap_s = 0
        eng._kv_pool = None
        eng._graphs.clear()
        lg0 = eng.prefill(emb, pad, 40, shared_rows=32).clone()
        assert (eng.kv is not None) == paged
        codes, logits = eng.decode(40, sp, return_logits=True)
        eng.prefill(emb, pad, 40, shared_rows=32)
        codes_g = eng.decode(40, sp, use_graph=True)
        kc, vc = eng.dense_kv(B, S + 38)
        for b_, p_ in enumerate(pad.tolist()):
            kc[:, b_, :, :p_] = 0
            vc[:, b_, :, :p_] = 0
        mel = torch.cat([torch.cat([eng.mel_emb[eng.start_mel][None], eng.mel_emb[codes[b, :5]], eng.mel_emb[eng.stop_mel][None]]) +
                         eng.mel_pos[:7] for b in range(B)])
        lat = eng.latent_mel_rows(mel, [7] * B)
        runs[paged] = (lg0, codes, logits, codes_g, kc, vc, lat)
        if paged:
            assert eng.kv.bs == 16 and eng.kv.used_blocks() == sum((S + 40) // 16 - int(p) // 16 + 1 for p in pad.tolist())
    eng.paged = True
    for a, b in zip(runs[False], runs[True]):
        assert torch.equal(a, b)
    assert torch.equal(runs[True][1], runs[True][3])          # graph replay == eager


def test_paged_kv_long_queue_through_few_slots_in_one_loop():
    """Continuous batching on the paged cache: 512 utterances through 32 decode slots in ONE loop (the contiguous cache had to
    drain and restart whenever its position budget was used up) -- the blocks of a row that has stopped go back to the pool and
    serve the utterance that takes its slot, the pool never holds more than the slots' windows, and the position counter runs
    past the block table's ring (64 entries x 16 positions) without a row ever seeing another row's keys: every utterance of a
    sample gets the greedy codes it gets decoded alone."""
    m = make_gpt(2, torch.float32)
    eng = m.engine
    rng = np.random.default_rng(123)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    for N, slots, max_new, hi in ((512, 32, 24, 22), (260, 4, 40, 38)):
        lens = sorted((int(v) for v in rng.integers(2, 15, size=N)), reverse=True)
        texts = [torch.from_numpy(rng.integers(2, 12000, size=n)).to(torch.int32) for n in lens]
        stops = [int(v) for v in rng.integers(3, hi, size=N)]
        sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)

        def prefix(ids):
            L = max(lens[i] for i in ids)
            bh = torch.full((len(ids), L), m.stop_text_token, dtype=torch.int32)
            for j, i in enumerate(ids):
                bh[j, : lens[i]] = texts[i]
            _, emb, mask = m.prepare_gpt_inputs(conds, bh.to(DEV))
            return emb, (mask == 0).sum(1).to(torch.int32)

        queue = list(range(slots, N))
        emb, pad = prefix(list(range(slots)))
        ce = 4
        eng.prefill(emb, pad, max_new + ce + 1, slots_window=emb.shape[1] + 1 + max_new + 2 * ce + 2)
        assert eng.kv is not None
        pool = eng.kv.blocks

        def feed(k):
            take = [queue.pop(0) for _ in range(min(k, len(queue)))]
            if not take:
                return []
            e, p = prefix(take)
            return [(e[j, int(p[j]):], stops[i]) for j, i in enumerate(take)]

        codes, leftover = eng.decode_refill(max_new, sp, feed, force_stop=stops[:slots], check_every=ce, staged=True)
        st = eng.refill_stats
        assert not leftover and len(codes) == N and eng.kv.blocks == pool          # one loop, the pool never grew
        assert st["rows_refilled"] == N - slots and st["peak_blocks"] <= st["blocks"]
        assert all(int(c[-1]) == m.stop_mel_token and c.numel() == stops[i] + 1 for i, c in enumerate(codes))
        if slots == 4:
            assert eng._S + st["steps"] > 64 * eng.kv.bs, "the position counter must have wrapped the block table's ring"
        # every row stopped and left: all blocks are back
        assert eng.kv.used_blocks() == 0
        for i in list(range(0, N, max(1, N // 10)))[:10]:
            c, lg = m.inference_speech(cond_mel, texts[i][None].to(DEV), do_sample=False, num_beams=1, repetition_penalty=10.0,
                                       max_generate_length=max_new, force_stop=[stops[i]], return_logits=True)
            want, got = c[0].cpu(), codes[i].cpu()
            for s_ in range(got.numel()):
                if int(got[s_]) != int(want[s_]):
                    top2 = torch.topk(lg[s_, 0].cpu(), 2).values
                    assert (top2[0] - top2[1]).item() < 1e-3, (N, i, s_, got, want)
                    break


@pytest.mark.parametrize("paged", [True, False])
def test_decode_refill_idle_slot_outlives_the_position_table_and_budget_is_checked(paged):
    """Two findings of the round-3 review.  (1) A slot whose row has stopped and is not refilled keeps stepping formally until the
    last row is done: its clock runs past the mel position table (803 rows) -- the embedding launch clamps the index instead of
    reading past the table.  (2) The loop runs whole blocks of check_every steps, so the cache must hold max_new + check_every
    positions behind the prompt: with contiguous cache rows a smaller reservation is refused (it used to spill K / V into the next
    head's rows); the paged cache deals blocks as the loop advances."""
    m = make_gpt(2, torch.float32)
    eng = m.engine
    eng.paged = paged
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    text = torch.tensor([[11, 22, 33, 44, 55, 66], [77, 88, 99, 1, 1, 1]], dtype=torch.int32, device=DEV)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    pad = (mask == 0).sum(1).to(torch.int32)
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)
    max_new, ce = 806, 16
    assert eng.mel_pos.shape[0] == 803
    if not paged:
        eng.prefill(emb, pad, max_new, paged=False)
        with pytest.raises(ValueError):      # a budget of exactly prompt + max_new + 1 positions (what round 3 accepted)
            eng.decode_refill(max_new, sp, lambda k: [], force_stop=[3, 805], check_every=ce, positions=eng._S + max_new + 1)
    eng.prefill(emb, pad, max_new + ce, paged=paged)
    codes, leftover = eng.decode_refill(max_new, sp, lambda k: [], force_stop=[3, 805], check_every=ce)
    assert not leftover and len(codes) == 2
    assert codes[0].numel() == 4 and codes[1].numel() == 806 and int(codes[1][-1]) == m.stop_mel_token
    assert eng.refill_stats["steps"] >= 806          # slot 0 idled for ~800 steps, its position index far past the table
    torch.cuda.synchronize()
    assert torch.isfinite(eng.h[:2]).all()


def test_infer_queue_equals_utterances_synthesised_one_by_one():
    """IndexTTS.infer_queue (continuous batching through 3 slots) returns, in input order, the waveforms infer_batch gives
    for each utterance alone; a tiny cache budget (several loops) changes nothing."""
    from indextts.infer import IndexTTS
    cfg = weights.reference_config()
    cfg["gpt"]["layers"] = 2
    tts = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device="cuda:0",
                                precision_config={"gpt": "fp32", "vocoder": "fp32"})
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    rng = np.random.default_rng(5)
    lens = [6, 13, 4, 9, 11, 5, 8]
    texts = [torch.from_numpy(rng.integers(2, 12000, size=n)).to(torch.int32) for n in lens]
    stops = [7, 3, 12, 5, 9, 4, 6]
    gen = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    want = [tts.infer_batch(cond_mel, [texts[i]], max_mel_tokens=20, force_stop=[stops[i]], return_codes=True, **gen)
            for i in range(7)]
    for cache in (4096, 160):
        outs, codes = tts.infer_queue(cond_mel, texts, slots=3, max_mel_tokens=20, force_stop=stops, return_codes=True,
                                      cache_positions=cache, **gen)
        for i in range(7):
            w, c = want[i][0][0], want[i][1][0]
            assert torch.equal(codes[i].long().cpu(), c.long().cpu()), (cache, i, codes[i], c)
            assert outs[i].shape == w.shape
            assert (outs[i] - w).abs().max().item() <= 1e-3 * max(1.0, w.abs().max().item()), (cache, i)
    # sampling: the loop's Philox stream is keyed by (slot, loop step) -- reproducible for a seed, different between seeds
    sgen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
    runs = [tts.infer_queue(cond_mel, texts, slots=3, max_mel_tokens=20, force_stop=stops, return_codes=True, seed=sd, **sgen)
            for sd in (5, 5, 6)]
    for i in range(7):
        assert torch.equal(runs[0][1][i], runs[1][1][i]) and torch.equal(runs[0][0][i], runs[1][0][i])
        assert runs[0][1][i].numel() == stops[i] and runs[0][0][i].numel() == stops[i] * 1024
    assert any(not torch.equal(runs[0][1][i], runs[2][1][i]) for i in range(7))
    assert tts.infer_queue(cond_mel, [], slots=3, num_beams=1) == []
    with pytest.raises(NotImplementedError):
        tts.infer_queue(cond_mel, texts, slots=3, num_beams=3)


def test_beam_decode_full_size_bf16_graph_equals_eager():
    """24 layers, bf16, 2 batch elements x 3 beams, beam-sample with the reference's default settings: the captured step
    (171 launches + beam step + in-place KV permutation) reproduces the eagerly launched loop token for token."""
    m = make_gpt(24, torch.bfloat16)
    eng = m.engine
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, torch.tensor([120], device=DEV))
    text = torch.tensor([[11, 22, 33, 44, 55, 66, 77, 88], [99, 111, 222, 1, 1, 1, 1, 1]], device=DEV)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    pad = (mask == 0).sum(1).to(torch.int32)
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=11, length_penalty=0.0)
    outs = []
    for use_graph in (False, True):
        eng.prefill(emb.repeat_interleave(3, 0), pad.repeat_interleave(3), 14, paged=False)
        outs.append(eng.decode_beam(14, sp, 3, use_graph=use_graph, check_every=4).cpu())
    assert outs[0].shape[0] == 2 and torch.equal(outs[0], outs[1])


def test_beam_graphs_follow_their_buffers(gpt_small_fp32):
    """decode_beam with B=2, then B=1, then B=2 on ONE engine: the per-(B, num_beams) buffers are reallocated in between,
    so a captured beam step must not survive them (it would replay over freed memory).  Graph replay == eager each time."""
    m, eng = gpt_small_fp32, gpt_small_fp32.engine
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    conds = m.get_conditioning(cond_mel, torch.tensor([120], device=DEV))
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=3, length_penalty=0.0)
    texts = {2: torch.tensor([[11, 22, 33, 44, 55, 66], [77, 88, 99, 1, 1, 1]], device=DEV),
             1: torch.tensor([[5, 6, 7, 8, 9]], device=DEV)}
    want = {}
    for B in (2, 1):
        _, emb, mask = m.prepare_gpt_inputs(conds, texts[B])
        pad = (mask == 0).sum(1).to(torch.int32)
        eng.prefill(emb.repeat_interleave(3, 0), pad.repeat_interleave(3), 12, paged=False)
        want[B] = eng.decode_beam(12, sp, 3, use_graph=False).cpu()
    for B in (2, 1, 2, 1):
        _, emb, mask = m.prepare_gpt_inputs(conds, texts[B])
        pad = (mask == 0).sum(1).to(torch.int32)
        eng.prefill(emb.repeat_interleave(3, 0), pad.repeat_interleave(3), 12, paged=False)
        got = eng.decode_beam(12, sp, 3, use_graph=True, check_every=4).cpu()
        assert torch.equal(got, want[B]), B


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_packed_and_row_major_activations_give_the_same_bits(dtype):
    """Packed vs row-major activations in the 7-launch decode step are re-arrangements of the same arithmetic: identical
    logits and codes, step for step."""
    m = make_gpt(2, dtype)
    eng = m.engine
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    conds = m.get_conditioning(cond_mel, None)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    pad = (mask == 0).sum(1).to(torch.int32)
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=9)
    outs = []
    eng.decode_mode = "launch"
    for pa in (True, False):
        eng.pa = pa
        eng._graphs.clear()
        eng.prefill(emb, pad, 20)
        outs.append(eng.decode(20, sp, return_logits=True))
    for codes, logits in outs[1:]:
        assert torch.equal(codes, outs[0][0]) and torch.equal(logits, outs[0][1])


def test_folded_decode_step_tracks_the_seven_launch_form():
    """bf16 decode loop, LayerNorm folded into the QKV / FC GEMMs (5 launches per block, the default) against the 7-launch form
    on the same weights, teacher-forced with the 7-launch form's greedy codes: the two differ only in where values are rounded
    to bf16 (raw residual rows vs normalised rows; gamma . W vs W), so the logits agree within the bf16 noise floor of either --
    measured against the fp32 engine on the same inputs, the folded form must not be further away than the 7-launch form by more
    than a small factor.  Also: graph replay == eager launches bit for bit in the folded form, geometry hints do not change
    the codes' validity, and the loop state (step, cache position) advances exactly once per token."""
    steps = 24
    m32, m16 = make_gpt(2, torch.float32), make_gpt(2, torch.bfloat16)
    g = np.load(os.path.join(G, "gpt_small.npz"))
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(g["text"]).to(DEV)
    sp = dict(do_sample=False, top_p=1.0, top_k=0, temperature=1.0, repetition_penalty=10.0, seed=0)

    from indextts.utils.accuracy import teacher_forced_logits

    def run(m, mode, force=None, use_graph=False):
        eng = m.engine
        eng.decode_mode = mode
        eng._graphs.clear()
        conds = m.get_conditioning(cond_mel, None)
        _, emb, mask = m.prepare_gpt_inputs(conds, text)
        pad = (mask == 0).sum(1).to(torch.int32)
        if force is None:
            eng.prefill(emb, pad, steps + 2)
            codes, logits = eng.decode(steps, sp, return_logits=True, use_graph=use_graph)
        else:
            codes, logits = force, teacher_forced_logits(eng, emb, pad, force, steps)
            eng._sample(eng._B, eng._seed_to_state(sp))
        torch.cuda.synchronize()
        return codes, logits, eng.state.tolist()

    c32, l32, _ = run(m32, "launch")
    _, l_launch, st_l = run(m16, "launch", force=c32)
    _, l_fold, st_f = run(m16, "fold", force=c32)
    assert st_l[:2] == st_f[:2]                                   # same step counter and cache position after the same steps
    assert l_launch.shape == l_fold.shape == l32.shape
    e_launch = (l_launch - l32).abs().max().item()
    e_fold = (l_fold - l32).abs().max().item()
    assert e_fold < max(2.0 * e_launch, 0.05), (e_fold, e_launch)
    rms_l = (l_launch - l32).pow(2).mean().sqrt().item()
    rms_f = (l_fold - l32).pow(2).mean().sqrt().item()
    assert rms_f < 1.6 * rms_l + 1e-3, (rms_f, rms_l)
    # free-running: graph replay == eager, both modes agree on the greedy codes wherever the fp32 margin is not tiny
    ce, le, _ = run(m16, "fold")
    cg, lg, stg = run(m16, "fold", use_graph=True)
    assert torch.equal(ce, cg) and torch.equal(le, lg)
    assert stg[0] == steps - 1 and stg[1] == m16.engine._S - 1 + steps - 1
    for rows in ((0, 0), (16, 16), (32, 16)):
        for wide in (False, True):
            m16.engine.fold_rows, m16.engine.fold_wide = list(rows), wide
            _, lv, _ = run(m16, "fold", force=c32)
            assert (lv - l32).abs().max().item() < max(2.0 * e_launch, 0.05)
    m16.engine.fold_rows, m16.engine.fold_wide = [16, 16], True


def test_runtime_lora_equals_merged_checkpoint():
    """Unmerged LoRA adapters attached at run time (GPTEngine.attach_lora: A rides as extra columns of the two output
    projections, B is applied by the reduce launch; c_attn / c_fc adapters are merged at attach time) against the way the
    reference ships fine-tuned models: the adapters merged into the base weights before loading (train.py:802-832).
    fp32, 2 layers: logits of the prefill and of 10 greedy decode steps agree to 1e-3; same greedy codes."""
    from indextts.gpt.model import UnifiedVoice
    cfg = dict(weights.reference_config()["gpt"], layers=2)
    sd = weights.gpt_state_dict(2)
    g = torch.Generator().manual_seed(3)
    r, scaling = 8, 2.0
    adapters = {}
    merged = dict(sd)
    for i in range(2):
        for name in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
            key = f"gpt.h.{i}.{name}"
            k_in, n_out = sd[key + ".weight"].shape
            A = torch.randn(r, k_in, generator=g) * 0.02
            Bm = torch.randn(n_out, r, generator=g) * 0.02
            adapters[key] = (A, Bm)
            merged[key + ".weight"] = sd[key + ".weight"] + (A.t() @ Bm.t()) * scaling
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    text = torch.from_numpy(np.load(os.path.join(G, "gpt_small.npz"))["text"]).to(DEV)
    kw = dict(do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=10, return_logits=True)
    outs = []
    for state, lora in ((merged, None), (sd, adapters)):
        m = UnifiedVoice(**cfg)
        m.load_state_dict(state)
        m.to(DEV).to(torch.float32).post_init_gpt2_config(kv_cache=True)
        if lora is not None:
            m.attach_lora(lora, scaling)
            assert m.engine.lora and "w_o_lora" in m.engine.layers[0] and m.engine.layers[1]["lora_n_pr"] == 1280 + 16
        outs.append(m.inference_speech(cond_mel, text, **kw))
    (c0, l0), (c1, l1) = outs
    assert (l0 - l1).abs().max().item() < 1e-3
    assert torch.equal(c0, c1)
    base = UnifiedVoice(**cfg)
    base.load_state_dict(sd)
    base.to(DEV).to(torch.float32).post_init_gpt2_config(kv_cache=True)
    cb, lbase = base.inference_speech(cond_mel, text, **kw)
    assert (lbase - l0).abs().max().item() > 1e-2, "the adapters must actually change the logits"
    # attach / detach are per engine: a fork taken BEFORE the attach keeps speaking with the base weights, a
    # second attach that names fewer modules starts from the base weights again, detach restores the base bits
    fork = base.replica()
    base.attach_lora(adapters, scaling)
    assert base.engine.lora and not fork.engine.lora and "w_o_lora" not in fork.engine.layers[0]
    assert torch.equal(fork.inference_speech(cond_mel, text, **kw)[1], lbase)
    assert torch.equal(base.inference_speech(cond_mel, text, **kw)[1], l1)
    only_proj = {k: v for k, v in adapters.items() if k.endswith("c_proj")}
    base.attach_lora(only_proj, scaling)
    assert "w_qkv_base" not in base.engine.layers[0] and "w_o_lora" in base.engine.layers[0]
    m2 = UnifiedVoice(**cfg)
    m2.load_state_dict(sd)
    m2.to(DEV).to(torch.float32).post_init_gpt2_config(kv_cache=True)
    m2.attach_lora(only_proj, scaling)
    assert torch.equal(base.inference_speech(cond_mel, text, **kw)[1], m2.inference_speech(cond_mel, text, **kw)[1])
    base.engine.detach_lora()
    assert torch.equal(base.inference_speech(cond_mel, text, **kw)[1], lbase)
