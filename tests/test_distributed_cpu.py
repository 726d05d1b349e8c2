"""world_size-2 gloo rehearsal of the N>1 path: one weight broadcast, utterance sharding, no data-path collective."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indextts.utils.dist import broadcast_state_dict, shard_utterances
    sd = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        sd = {"a.weight": torch.randn(5, 7, generator=g), "b.bias": torch.randn(3, generator=g).half(),
              "bn.num_batches_tracked": torch.tensor(12), "c": torch.randn(2, 2, 2, generator=g)}
    out = broadcast_state_dict(sd, src=0, device="cpu")
    lens = [50, 10, 40, 40, 5, 90, 20, 20]
    mine = shard_utterances(lens, world)[rank]
    # each rank "processes" only its own shard; results meet on rank 0 through a gather of small int tensors
    done = torch.zeros(len(lens), dtype=torch.int64)
    done[mine] = rank + 1
    dist.all_reduce(done)  # bookkeeping only (not on the data path)
    q.put((rank, {k: v.clone() for k, v in out.items()}, done.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sd0, done0), (_, sd1, done1) = res
    assert set(sd0) == set(sd1) == {"a.weight", "b.bias", "bn.num_batches_tracked", "c"}
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]) and sd0[k].shape == sd1[k].shape
    assert sd1["bn.num_batches_tracked"].item() == 12 and sd1["bn.num_batches_tracked"].dtype == torch.int64
    assert done0 == done1 and all(d in (1, 2) for d in done0)
