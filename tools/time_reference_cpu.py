#!/usr/bin/env python3
"""Time the REFERENCE's own modules on the benched workload shape, on the host cores of the BUILD CONTAINER
(/root/reference never travels to the GPU box).  Writes profiles/r02_cpu_reference.json; bench.py attaches that file to
its `cpu_baseline` object so that the port's number (timed on the GPU box) has the reference's beside it.

What runs, reference code imported in place through tests/golden/_ref_harness.py with
name-hashed synthetic weights at the real shapes (24 layers, d = 1280; BigVGAN 133.9 M parameters), fp32:
  pass 1  GPT2InferenceModel.forward driven manually with its KV cache (prefill with past=None, then one-token calls:
          transformers 5.15's generate() cannot drive this model), `rows` of the 32 benched utterances as ONE
          left-padded batch, the installed transformers' RepetitionPenalty / TopK / TopP processors + torch.multinomial
          (repetition penalty 10, k = 30, p = 0.8), 140 acoustic tokens each
  pass 2  UnifiedVoice.forward(..., return_latent=True), per row (infer.py:864-874)
  pass 3  BigVGAN.forward(latent, mel_ref) with use_cuda_kernel=False, per row (infer.py:886-890)
RTF formula of infer.py:900: elapsed / audio seconds; reported as audio-seconds per second, median of `repeats` runs
after one warm-up run of the decode part.

    python tools/time_reference_cpu.py [--rows 4] [--tokens 140] [--threads 8] [--repeats 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden as mg  # noqa: E402  (installs the import stand-ins and puts /root/reference on sys.path)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=140)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.set_grad_enabled(False)
    from transformers import RepetitionPenaltyLogitsProcessor, TopKLogitsWarper, TopPLogitsWarper

    t0 = time.time()
    m = mg.build_gpt(24)
    g = mg.build_bigvgan()
    print(f"[ref-cpu] reference modules built in {time.time() - t0:.1f}s", file=sys.stderr)
    # the benched inputs (bench.py make_workload(3, 1)): text U{20..60} seed 2, shared 3.2 s prompt
    gen = torch.Generator().manual_seed(2)
    lens = torch.randint(20, 61, (32,), generator=gen)
    texts = [torch.randint(2, 12000, (int(k),), generator=gen) for k in lens][: args.rows]
    L = max(int(t.numel()) for t in texts)
    text = torch.full((args.rows, L), 1, dtype=torch.long)
    for i, t in enumerate(texts):
        text[i, : t.numel()] = t
    cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0))
    cml = torch.tensor([300])
    procs = [RepetitionPenaltyLogitsProcessor(10.0), TopKLogitsWarper(30), TopPLogitsWarper(0.8)]
    im = m.inference_model

    def decode():
        conds = m.get_conditioning(cond_mel, cml)
        fake, emb, mask = m.prepare_gpt_inputs(conds, text)
        im.store_mel_emb(emb)
        r = im(input_ids=fake, attention_mask=mask, past_key_values=None, use_cache=True, return_dict=True)
        past, history, cur_mask, codes = r.past_key_values, fake.clone(), mask, []
        torch.manual_seed(1234)
        for s in range(args.tokens):
            sc = r.logits[:, -1, :].float()
            for p in procs:
                sc = p(history, sc)
            nxt = torch.multinomial(torch.softmax(sc, -1), 1)[:, 0]
            codes.append(nxt)
            history = torch.cat([history, nxt[:, None]], 1)
            if s + 1 == args.tokens:
                break
            cur_mask = torch.cat([cur_mask, torch.ones(cur_mask.shape[0], 1, dtype=cur_mask.dtype)], 1)
            r = im(input_ids=nxt[:, None], attention_mask=cur_mask, past_key_values=past, use_cache=True, return_dict=True)
            past = r.past_key_values
        return torch.stack(codes, 1)

    def rest(codes):
        n = 0
        for b in range(args.rows):
            tl = int(texts[b].numel())
            lat = m(cond_mel, text[b: b + 1, :tl], torch.tensor([tl]), codes[b: b + 1], torch.tensor([codes.shape[1] * 1024]),
                    cond_mel_lengths=cml, return_latent=True)
            wav, _ = g(lat, cond_mel.transpose(1, 2))
            n += wav.shape[-1]
        return n

    print("[ref-cpu] warm-up ...", file=sys.stderr)
    codes = decode()
    runs = []
    for r in range(args.repeats):
        t1 = time.perf_counter()
        codes = decode()
        t2 = time.perf_counter()
        samples = rest(codes)
        t3 = time.perf_counter()
        runs.append(dict(decode_s=t2 - t1, latent_vocoder_s=t3 - t2, total_s=t3 - t1, audio_s=samples / 24000.0))
        print(f"[ref-cpu] run {r}: {runs[-1]}", file=sys.stderr)
    med = sorted(runs, key=lambda d: d["total_s"])[len(runs) // 2]
    out = {
        "what": "the reference's own modules (indextts.gpt.model.UnifiedVoice / GPT2InferenceModel manual cached drive, "
                "UnifiedVoice.forward(return_latent=True), indextts.BigVGAN.models.BigVGAN.forward use_cuda_kernel=False) "
                "imported from /root/reference in the build container, synthetic weights at the real shapes, fp32",
        "workload": f"{args.rows} of the 32 benched rows as one left-padded batch, HF processors (penalty 10, k=30, p=0.8) + "
                    f"torch.multinomial, {args.tokens} acoustic tokens each; latent pass and vocoder per row",
        "cores": args.threads, "host": "build container (8 vCPU Xeon @ 2.1 GHz)", "torch": torch.__version__,
        "value": round(med["audio_s"] / med["total_s"], 4), "unit": "audio-seconds/sec",
        "rtf": round(med["total_s"] / med["audio_s"], 3),
        "median_run": {k: round(v, 3) for k, v in med.items()}, "runs": [{k: round(v, 3) for k, v in d.items()} for d in runs],
        "ms_per_decode_step": round(1e3 * med["decode_s"] / args.tokens, 1),
        "kind": "reference",
    }
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
