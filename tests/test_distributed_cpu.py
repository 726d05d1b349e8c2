"""world_size-2 gloo rehearsal of the N>1 path: one weight broadcast, utterance sharding, no data-path collective."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    """A port the kernel just handed out (a pid-derived number can be taken, which made the rendezvous fail now and then)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indextts.utils.dist import broadcast_state_dict, shard_utterances
    sd = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        sd = {"a.weight": torch.randn(5, 7, generator=g), "b.bias": torch.randn(3, generator=g).half(),
              "bn.num_batches_tracked": torch.tensor(12), "c": torch.randn(2, 2, 2, generator=g),
              "big.counter": torch.tensor([2 ** 24 + 1, 2 ** 40 + 3]), "w.bf16": torch.randn(4, 8, generator=g).bfloat16(),
              "flag": torch.tensor([True, False, True])}
    out = broadcast_state_dict(sd, src=0, device="cpu")
    lens = [50, 10, 40, 40, 5, 90, 20, 20]
    mine = shard_utterances(lens, world)[rank]
    # each rank "processes" only its own shard; results meet on rank 0 through a gather of small int tensors
    done = torch.zeros(len(lens), dtype=torch.int64)
    done[mine] = rank + 1
    dist.all_reduce(done)  # bookkeeping only (not on the data path)
    # plain bytes through the queue: tensors would travel as shared-memory handles that die with this process (a race
    # that failed the test now and then)
    wire = {k: (str(v.dtype).replace("torch.", ""), tuple(v.shape), v.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()
                if v.dtype != torch.bool else v.to(torch.uint8).numpy().tobytes()) for k, v in out.items()}
    q.put((rank, wire, done.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    def unwire(w):
        out = {}
        for k, (dt, shape, raw) in w.items():
            tdt = getattr(torch, dt)
            t = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            out[k] = (t.to(torch.bool) if tdt == torch.bool else t.view(tdt)).reshape(shape)
        return out

    (_, sd0, done0), (_, sd1, done1) = res
    sd0, sd1 = unwire(sd0), unwire(sd1)
    # the source tensors, rebuilt here: every rank must hold them bit for bit, in their own dtypes
    g = torch.Generator().manual_seed(0)
    src = {"a.weight": torch.randn(5, 7, generator=g), "b.bias": torch.randn(3, generator=g).half(),
           "bn.num_batches_tracked": torch.tensor(12), "c": torch.randn(2, 2, 2, generator=g),
           "big.counter": torch.tensor([2 ** 24 + 1, 2 ** 40 + 3]), "w.bf16": torch.randn(4, 8, generator=g).bfloat16(),
           "flag": torch.tensor([True, False, True])}
    assert list(sd0) == list(sd1) == list(src)
    for k in src:
        for got in (sd0[k], sd1[k]):
            assert got.dtype == src[k].dtype and got.shape == src[k].shape and torch.equal(got, src[k]), k
    assert sd1["big.counter"].tolist() == [2 ** 24 + 1, 2 ** 40 + 3]     # would be rounded through an fp32 arena
    assert done0 == done1 and all(d in (1, 2) for d in done0)


def test_compact_state_dicts_keep_engine_bits():
    """What bench.py broadcasts: GEMM weights pre-cast to the compute dtype / weight norm pre-folded.  Casting before the
    broadcast must give the very bits the loaders would produce from the original state dict."""
    from indextts.BigVGAN.models import fold_weight_norm
    from indextts.utils.dist import arena_bytes, compact_bigvgan_state_dict, compact_gpt_state_dict
    g = torch.Generator().manual_seed(1)
    gsd = {"gpt.h.0.attn.c_attn.weight": torch.randn(8, 24, generator=g), "gpt.h.0.attn.c_attn.bias": torch.randn(24, generator=g),
           "gpt.h.0.ln_1.weight": torch.randn(8, generator=g), "mel_head.weight": torch.randn(10, 8, generator=g),
           "mel_embedding.weight": torch.randn(10, 8, generator=g), "conditioning_encoder.x.weight": torch.randn(3, 3, generator=g)}
    c = compact_gpt_state_dict(gsd, torch.bfloat16)
    assert c["gpt.h.0.attn.c_attn.weight"].dtype == torch.bfloat16 and c["mel_head.weight"].dtype == torch.bfloat16
    assert torch.equal(c["gpt.h.0.attn.c_attn.weight"], gsd["gpt.h.0.attn.c_attn.weight"].to(torch.bfloat16))
    for k in ("gpt.h.0.attn.c_attn.bias", "gpt.h.0.ln_1.weight", "mel_embedding.weight", "conditioning_encoder.x.weight"):
        assert c[k].dtype == torch.float32 and torch.equal(c[k], gsd[k])
    assert arena_bytes(c) < arena_bytes(gsd)
    assert compact_gpt_state_dict(gsd, torch.float32)["mel_head.weight"].dtype == torch.float32
    bsd = {"conv_pre.weight_g": torch.rand(6, 1, 1, generator=g) + 0.5, "conv_pre.weight_v": torch.randn(6, 4, 7, generator=g),
           "conv_pre.bias": torch.randn(6, generator=g), "speaker_encoder.fc.weight": torch.randn(3, 3, generator=g)}
    cb = compact_bigvgan_state_dict(bsd, torch.float16)
    assert set(cb) == {"conv_pre.weight", "conv_pre.bias", "speaker_encoder.fc.weight"}
    want = fold_weight_norm(bsd["conv_pre.weight_g"], bsd["conv_pre.weight_v"]).to(torch.float16)
    assert torch.equal(cb["conv_pre.weight"], want) and cb["conv_pre.bias"].dtype == torch.float32


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra_args, env_extra=None, launcher=False):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra_args
    if launcher:  # the driver's command line for N > 1
        port = _free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + extra_args
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def _check_two_rank_line(js):
    assert js["n_gpus"] == 2 and js["scaling"] == "weak"
    assert sorted(r["rank"] for r in js["ranks_seen"]) == [0, 1]
    assert len({r["uuid"] for r in js["ranks_seen"]}) == 2, "two distinct processes must have reported"
    rows = [set(p["rows"]) for p in js["per_rank"]]
    assert len(rows) == 2 and not (rows[0] & rows[1]) and rows[0] | rows[1] == set(range(64)), "one global list, disjoint shards"
    assert all(len(r) == 32 for r in rows)
    want = sum(p["audio_s_per_step"] for p in js["per_rank"])
    assert abs(js["audio_seconds_per_step_job"] - want) < 1e-6
    assert abs(js["value"] - want * js["steps"] / (js["ms_per_step"] * 1e-3 * js["steps"])) / js["value"] < 1e-2


def test_bench_self_launch_two_ranks_stub():
    """`python bench.py --gpus 2` with NO launcher must start two ranks itself (gloo, stub engine): the control flow the
    driver's single-process command line exercises."""
    rc, js, err = _run_bench(["--gpus", "2", "--stub", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    _check_two_rank_line(js)
    # config 4: mixed lengths from ONE global list -> shards carry different amounts of audio
    rc, js, err = _run_bench(["--gpus", "2", "--stub", "--steps", "1", "--warmup", "0", "--config", "4"])
    assert rc == 0, err[-2000:]
    a = [p["audio_s_per_step"] for p in js["per_rank"]]
    assert a[0] != a[1] and all(32 * 40 * 1024 / 24000 <= x <= 32 * 400 * 1024 / 24000 for x in a)


def test_bench_eight_ranks_config4_stub():
    """BASELINE config 4's real shape, rehearsed on CPU so that the first 8-GPU run is not also the first 8-rank run:
    `bench.py --gpus 8 --config 4` (self-launch, gloo, stub engine) -- 256 mixed-length utterances from ONE global list, eight
    disjoint shards of 32 (the serpentine longest-first deal of indextts/utils/dist.py), every rank reports, the shards carry
    different amounts of audio, the aggregate is their sum and the tail imbalance is in the line."""
    rc, js, err = _run_bench(["--gpus", "8", "--stub", "--steps", "1", "--warmup", "0", "--config", "4"])
    assert rc == 0, err[-2000:]
    assert js["n_gpus"] == 8 and js["scaling"] == "weak"
    assert sorted(r["rank"] for r in js["ranks_seen"]) == list(range(8))
    assert len({r["uuid"] for r in js["ranks_seen"]}) == 8
    rows = [set(p["rows"]) for p in sorted(js["per_rank"], key=lambda p: p["rank"])]
    assert all(len(r) == 32 for r in rows)
    assert set().union(*rows) == set(range(256)) and sum(len(r) for r in rows) == 256, "eight disjoint shards of one list"
    audio = [p["audio_s_per_step"] for p in js["per_rank"]]
    assert len(set(audio)) > 1 and all(32 * 40 * 1024 / 24000 <= a <= 32 * 400 * 1024 / 24000 for a in audio)
    assert abs(js["audio_seconds_per_step_job"] - sum(audio)) < 1e-6
    assert js["tail_imbalance"] >= 1.0
    # the deal balances what is known up front (text length): no shard holds more than a few per cent more text than another
    import bench
    texts, _ = bench.make_workload(4, 8)
    load = [sum(int(texts[i].numel()) for i in r) for r in rows]
    assert max(load) / min(load) < 1.05, load


def test_bench_under_torchrun_two_ranks_stub():
    rc, js, err = _run_bench(["--gpus", "2", "--stub", "--steps", "1", "--warmup", "0"], launcher=True)
    assert rc == 0, err[-2000:]
    _check_two_rank_line(js)


def test_bench_child_failure_is_fatal():
    rc, js, err = _run_bench(["--gpus", "2", "--stub", "--steps", "1", "--warmup", "0"], {"ITTS_BENCH_STUB_FAIL_RANK": "1"})
    assert rc != 0 and js is None
    # a world size that does not match --gpus is refused instead of silently running one rank
    rc, js, err = _run_bench(["--gpus", "2", "--stub"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and js is None
