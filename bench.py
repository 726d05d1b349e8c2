#!/usr/bin/env python3
"""bench.py -- IndexTTS inference hot path on MI355X: audio-seconds per second at batch 32 (BASELINE.json metric).

One "step" = one full pass of the hot path over one synthetic utterance batch (BASELINE config 3): 32 utterances that
share a prompt, text lengths U{20..60}, top-k/top-p sampling (k=30, p=0.8, T=1.0, repetition penalty 10, num_beams=1),
every row force-stopped after 140 acoustic tokens (random weights never emit EOS; 140 tokens = 5.97 s of audio):
conditioner -> prefix -> prefill -> 140-step cached decode loop (CUDA-graph replay) -> teacher-forced latent pass ->
BigVGAN vocoder -> PCM clamp.  GPT in bf16, vocoder in fp16, fp32 accumulation.  Weights are name-hashed synthetic
tensors at the real shapes (no checkpoints exist offline), built on rank 0 and broadcast over RCCL to the other ranks;
there is no collective inside the timed region (utterances are independent -> weak scaling, one process per GPU).

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline     -- the dominant kernel of the step: per-launch HIP-event timing of an identical instrumented step
  cpu_baseline -- the oracle (CPU restatement, fp32) timed on the host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

BATCH = 32
MEL_TOKENS = 140
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_weights_rank0():
    import weights
    t0 = time.time()
    gsd = weights.gpt_state_dict(24)
    bsd = weights.bigvgan_state_dict()
    log(f"[bench] synthesised weights in {time.time() - t0:.1f}s")
    return gsd, bsd


def broadcast_state(sd_or_none, rank, world, device):
    """rank 0 -> all: one flat fp32 arena over RCCL (xGMI inside a node)."""
    from indextts.utils.dist import broadcast_state_dict
    if world == 1:
        return sd_or_none
    return broadcast_state_dict(sd_or_none, src=0, device=device)


class KernelTimer:
    """Per-launch HIP-event timing of the C-ABI entry points (events are recorded on the stream the kernels run on)."""
    NAMES = ("gemm_conv", "gemm_skinny", "aa_snake", "attn_decode", "attn_prefill", "layernorm", "ln_reduce", "sample",
             "embed_step", "tanh_pcm")

    def __init__(self, nat):
        self.nat, self.rec, self.orig = nat, [], {}

    def __enter__(self):
        for n in self.NAMES:
            f = getattr(self.nat, n)
            self.orig[n] = f

            def wrapped(*a, _f=f, _n=n, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _f(*a, **k)
                e1.record()
                self.rec.append((_n, e0, e1, self._work(_n, a, k)))
                return r
            setattr(self.nat, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.nat, n, f)

    @staticmethod
    def _work(name, a, k):
        """(flops, algorithmic bytes) of one launch."""
        es = lambda dt: 4 if dt == torch.float32 else 2  # noqa: E731
        if name == "gemm_conv":
            dt, B, Tin, Tout, Cin, N = a[:6]
            taps = k.get("taps", 1)
            fl = 2.0 * B * Tout * N * Cin * taps
            by = B * Tin * Cin * es(dt) + B * Tout * N * (4 if k.get("y_f32") else es(dt)) + taps * Cin * N * es(dt)
            return fl, by
        if name == "gemm_skinny":
            dt, M, N, K = a[:4]
            return 2.0 * M * N * K, K * N * es(dt) + M * K * es(dt) + M * N * 4  # weights once + activations in/out
        if name == "aa_snake":
            x = a[0]
            return 60.0 * x.numel(), 2 * x.numel() * x.element_size()
        if name == "attn_decode":
            q, kc = a[0], a[1]
            return 0.0, 0.0  # context-dependent; priced through decode_step below
        return 0.0, 0.0

    @staticmethod
    def event_overhead_ms(n=200):
        """Median elapsed time of an EMPTY event pair: what a start/stop pair adds on top of the kernel it brackets."""
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in pairs]))

    def summary(self):
        torch.cuda.synchronize()
        ovh = self.event_overhead_ms()
        self.overhead_ms = ovh
        agg = {}
        for n, e0, e1, (fl, by) in self.rec:
            d = agg.setdefault(n, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += max(e0.elapsed_time(e1) - ovh, 0.0)
            d[2] += fl
            d[3] += by
        return agg


def make_inputs(rank, device):
    g = torch.Generator().manual_seed(2 + 1000 * rank)
    lens = torch.randint(20, 61, (BATCH,), generator=g)
    texts = [torch.randint(2, 12000, (int(n),), generator=g).to(torch.int32) for n in lens]
    import synth
    cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(device)
    return cond_mel, texts


def cpu_baseline(gsd, bsd, cond_conds, n_tokens=96):
    """Oracle (oracle/*.py, fp32, torch CPU) on ONE utterance of the same workload, n_tokens acoustic tokens:
    prefill + cached greedy steps + latent pass + vocoder.  Returns (audio_s_per_s, cores, description)."""
    from oracle import bigvgan_ref, gpt_ref
    cores = min(len(os.sched_getaffinity(0)), 16)  # the box's CPU share, not the host's core count
    torch.set_num_threads(cores)
    W = {k: v.float() for k, v in gsd.items() if k.startswith(("gpt.", "final_norm", "mel_", "text_"))}
    g = torch.Generator().manual_seed(2)
    text = torch.randint(2, 12000, (1, 40), generator=g)
    conds = cond_conds.cpu().float()
    VW = bigvgan_ref.Weights({k: v.numpy() for k, v in bsd.items()})
    spk = torch.zeros(1, 512, 1)
    t0 = time.perf_counter()
    emb, mask, _ = gpt_ref.prepare_gpt_inputs(conds, text, W)
    lg, past = gpt_ref.decode_prefill(emb, mask, W)
    codes = []
    for s in range(1, n_tokens + 1):
        tok = lg.argmax(-1)
        codes.append(int(tok))
        if s == n_tokens:
            break
        mask = torch.cat([mask, torch.ones(1, 1, dtype=torch.bool)], 1)
        lg, past = gpt_ref.decode_step(tok, s, mask, past, W)
    lat = gpt_ref.latent_pass(conds, text[0], torch.tensor(codes), W)
    wav = bigvgan_ref.forward(lat, spk, VW)
    dt = time.perf_counter() - t0
    audio_s = wav.shape[-1] / 24000.0
    return audio_s / dt, cores, (f"oracle fp32 on {cores} host threads: 1 utterance, 40 text tokens, {n_tokens} acoustic tokens "
                                 f"(prefill + cached decode + latent pass + BigVGAN), {dt:.1f}s of CPU work")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight for --schedule concurrent")
    ap.add_argument("--no-concurrency", action="store_true",
                    help="skip the extra (reported, never `value`) measurement with two batch-32 requests in flight")
    ap.add_argument("--schedule", choices=["pipelined", "serial", "concurrent"], default="serial",
                    help="pipelined: latent pass + vocoder of batch i on a second HIP stream beside the token loop of batch "
                         "i+1 (BatchPipeline; measured +1.6 %: the workgroup dispatcher serialises the big launches of one "
                         "queue against the small ones of the other); serial: one batch at a time on one stream; "
                         "concurrent: --inflight independent batch-32 requests at a time, each on its own engine instance, "
                         "thread and HIP stream (serving concurrency; the token loops of different requests interleave on "
                         "the CUs, which the latency-bound loop of a single request leaves mostly idle)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    # Rehearsal aid for a one-GPU box (not used by the driver): ITTS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # ITTS_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device; the control flow is the same.
    if os.environ.get("ITTS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("ITTS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    torch.set_grad_enabled(False)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import weights
    from indextts import _native as nat
    from indextts.infer import BatchPipeline, IndexTTS, RequestPool

    gsd = bsd = None
    if rank == 0:
        gsd, bsd = build_weights_rank0()
    gsd_d = broadcast_state(gsd, rank, world, device)
    bsd_d = broadcast_state(bsd, rank, world, device)
    cfg = weights.reference_config()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # keep stdout for the single JSON line
        tts = IndexTTS.from_weights(cfg, gsd_d, bsd_d, device=device,
                                    precision_config={"gpt": "bf16", "vocoder": "fp16"})
    want_conc = args.schedule == "concurrent" or (world == 1 and not args.no_concurrency)
    extra = [tts.replica() for _ in range(max(2, args.inflight) - 1)] if want_conc else []  # shared weights, private state
    del gsd_d, bsd_d
    cond_mel, texts = make_inputs(rank, device)
    force = [MEL_TOKENS] * BATCH
    gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)

    def step(seed, phase_events=None):
        return tts.infer_batch(cond_mel, texts, max_mel_tokens=MEL_TOKENS + 1, force_stop=force, seed=seed,
                               phase_events=phase_events, **gen)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipe = BatchPipeline(tts) if args.schedule == "pipelined" else None
    pool = None

    def make_pool():
        pl = RequestPool([tts] + extra)
        pl.warm_up(cond_mel, texts, max_mel_tokens=MEL_TOKENS + 1, force_stop=force, seed=900, **gen)
        return pl

    if args.schedule == "concurrent":
        pool = make_pool()

    def run_steps(n, seed0):
        """n steps of the hot path; every step's waveforms are complete when this returns."""
        if n <= 0:
            return None
        if pool is not None:
            jobs = [pool.submit(cond_mel, texts, max_mel_tokens=MEL_TOKENS + 1, force_stop=force, seed=seed0 + k, **gen)
                    for k in range(n)]
            return [j.result() for j in jobs][-1]
        if pipe is None:
            o = None
            for k in range(n):
                o = step(seed0 + k)
            return o
        tickets, marks = [], [time.perf_counter()]
        for k in range(n):
            tickets.append(pipe.submit(cond_mel, texts, max_mel_tokens=MEL_TOKENS + 1, force_stop=force, seed=seed0 + k, **gen))
            marks.append(time.perf_counter())
        o = [t.result() for t in tickets][-1]
        marks.append(time.perf_counter())
        log("[bench] pipeline host marks (ms): " + " ".join(f"{1e3 * (b - a):.1f}" for a, b in zip(marks[:-1], marks[1:])))
        return o

    outs = run_steps(args.warmup, 1234)
    barrier()
    t0 = time.perf_counter()
    outs = run_steps(args.steps, 2000)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    log(f"[bench] rank {rank}: {args.steps} steps in {elapsed:.3f}s")
    samples = sum(int(o.numel()) for o in outs)
    audio_s_step = samples / 24000.0
    assert samples == BATCH * MEL_TOKENS * 1024, f"unexpected audio length {samples}"
    value = audio_s_step * world * args.steps / elapsed

    # phase split of one more (un-instrumented, graph-replayed) step, run serially on one stream
    pe = {}
    step(3000, pe)
    torch.cuda.synchronize()
    ts = time.perf_counter()
    step(3001)
    torch.cuda.synchronize()
    serial_ms = 1e3 * (time.perf_counter() - ts)
    names = ["start", "conditioned", "prefilled", "decoded", "latents", "vocoded"]
    phases = {f"{a}->{b}": round(pe[a].elapsed_time(pe[b]), 3) for a, b in zip(names[:-1], names[1:])}
    eng = tts.gpt.engine
    S0 = eng._S
    dec_ms = phases["prefilled->decoded"]
    step_us = 1e3 * dec_ms / MEL_TOKENS
    ctx_mid = S0 + MEL_TOKENS // 2
    dec_bytes = eng.step_bytes(BATCH, ctx_mid)
    log(f"[bench] phases (ms): {phases}; decode step {step_us:.1f} us, {dec_bytes / 1e6:.0f} MB/step -> "
        f"{dec_bytes / (step_us * 1e-6) / 1e9:.0f} GB/s")

    # p50 first-token latency: cached prompt mel -> conditioner + prefix + prefill + first sample
    lat_ms = []
    batch_tokens = torch.full((BATCH, max(int(t.numel()) for t in texts)), 1, dtype=torch.int32, device=device)
    for i, t in enumerate(texts):
        batch_tokens[i, : t.numel()] = t.to(device)
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=1)
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        conds = tts.gpt.get_conditioning(cond_mel, None)
        _, emb, mask = tts.gpt.prepare_gpt_inputs(conds, batch_tokens)
        tts.gpt.engine.prefill(emb, (mask == 0).sum(1).to(torch.int32), 4)
        tts.gpt.engine._sample(BATCH, sp)
        torch.cuda.synchronize()
        lat_ms.append((time.perf_counter() - t1) * 1e3)
    first_token_ms = float(np.median(lat_ms))

    result = {
        "metric": "audio-seconds/sec (RTF^-1) at batch 32, whole pipeline (GPT decode + latent pass + BigVGAN)",
        "value": round(value, 2), "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: batch=32 utterances/GPU, top-k sampling (k=30,p=0.8), 140 acoustic tokens "
                               "(5.97 s) each, shared 3.2 s prompt, text U{20..60} tokens, random-init weights at real shapes",
                   "gpt_dtype": "bf16", "vocoder_dtype": "fp16", "parallelism": f"dp{world} (utterance sharding, no collectives)",
                   "audio_seconds_per_step_per_gpu": round(audio_s_step, 3),
                   "schedule": ("2-stage batch pipeline: latent pass + vocoder of batch i on a second HIP stream beside "
                                "the token loop of batch i+1; all steps complete inside the timed region")
                   if pipe is not None else (f"concurrent: {len(pool.instances)} independent batch-32 requests in flight (one "
                                             "engine instance, thread and HIP stream each)" if pool is not None else
                                             "serial: one batch at a time on one stream")},
        "serial_ms_per_step": round(serial_ms, 3),
        "first_token_ms_p50": round(first_token_ms, 2),
        "phases_ms": phases,
        "decode_step": {"us": round(step_us, 2), "algorithmic_MB": round(dec_bytes / 1e6, 1), "ctx": ctx_mid,
                        "GBps": round(dec_bytes / (step_us * 1e-6) / 1e9, 1),
                        "frac_of_hbm_peak": round(dec_bytes / (step_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)},
    }

    if rank == 0 and world == 1 and want_conc and args.schedule != "concurrent":
        # Reported beside `value`, never as `value`: two independent batch-32 requests in flight (one engine instance,
        # thread and HIP stream each).  A single request's token loop is latency-bound and leaves the CUs mostly idle.
        pl = make_pool()
        nrun = max(4, 2 * len(pl.instances))

        def run_conc(seed0):
            jobs = [pl.submit(cond_mel, texts, max_mel_tokens=MEL_TOKENS + 1, force_stop=force, seed=seed0 + k, **gen)
                    for k in range(nrun)]
            for j in jobs:
                j.result()
        run_conc(5000)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        run_conc(6000)
        torch.cuda.synchronize()
        dtc = time.perf_counter() - tc
        result["concurrent_requests"] = {"inflight": len(pl.instances), "steps": nrun, "value": round(audio_s_step * nrun / dtc, 2),
                                         "unit": "audio-seconds/sec", "ms_per_step": round(1e3 * dtc / nrun, 3),
                                         "note": "serving concurrency (indextts.infer.RequestPool): independent batch-32 requests "
                                                 "overlap; `value` above is one request at a time"}
        pl.close()
        log(f"[bench] {len(pl.instances)} requests in flight: {result['concurrent_requests']['value']} audio-s/s")

    if rank == 0 and not args.no_roofline:
        # one more identical step with per-launch HIP events (eager launches instead of graph replay)
        eng = tts.gpt.engine
        eng.force_eager = True
        with KernelTimer(nat) as kt:
            torch.cuda.synchronize()
            step(4242)
            agg = kt.summary()
        eng.force_eager = False
        tot = sum(v[1] for v in agg.values())
        breakdown = {n: {"launches": v[0], "ms": round(v[1], 3), "share": round(v[1] / tot, 3)} for n, v in
                     sorted(agg.items(), key=lambda kv: -kv[1][1])}
        # Dominant kernel = the decode-step weight-streaming GEMM (gemm_skinny_kernel, ~1/3 of the step, HBM-bound).
        # Its launches are microseconds long, so they are timed the way they run in the timed region: as a replayed
        # graph holding exactly one decode step's 97 GEMM launches, bracketed by one HIP-event pair on that stream.
        n_l, by_l = eng.gemm_launches_of_step(BATCH)  # warm-up
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            eng.gemm_launches_of_step(BATCH)
        gg.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            gg.replay()
        e1.record()
        torch.cuda.synchronize()
        us_launch = 1e3 * e0.elapsed_time(e1) / (reps * n_l)
        ach = (by_l / n_l) / (us_launch * 1e-6) / 1e9
        # HBM traffic per launch: rocprofv3 PMC passes cannot run inside this process; the committed summary of the
        # separate FETCH_SIZE / WRITE_SIZE passes over exactly these launches (tools/pmc_decode_gemm.py) is reported.
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_gemm_skinny.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
            traffic_src = "profiles/r01_pmc_gemm_skinny.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 read correction)"
        roof = {"kernel": "gemm_skinny_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_decode_step": n_l,
                "avg_launch_us": round(us_launch, 2), "algorithmic_MB_per_launch": round(by_l / n_l / 1e6, 3),
                "how": "graph replay of one decode step's GEMM launches, 50 replays between one HIP event pair"}
        if "gemm_conv" in agg:
            c2, ms2, fl2, _ = agg["gemm_conv"]
            roof["gemm_conv_mfma"] = {"achieved_TFLOPs": round(fl2 / (ms2 * 1e-3) / 1e12, 1), "peak": PEAK_MFMA_TFLOPS,
                                      "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS, 4), "launches": c2,
                                      "avg_launch_us": round(1e3 * ms2 / c2, 1)}
        result["roofline"] = roof
        result["kernel_breakdown"] = breakdown
        result["event_pair_overhead_us"] = round(1e3 * kt.overhead_ms, 2)
        log("[bench] kernel breakdown (instrumented eager step):", json.dumps(breakdown))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        conds = tts.gpt.get_conditioning(cond_mel, None)
        v, cores, desc = cpu_baseline(gsd, bsd, conds)
        result["cpu_baseline"] = {"value": round(v, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port", "sample": desc}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
