"""Multi-GPU data parallelism for the inference path (new design; the reference has no multi-GPU inference code,
SURVEY.md §2.1/§8e): one process per GPU, every rank holds a full weight replica received ONCE from rank 0 through a
single RCCL broadcast of a flat arena (xGMI inside a node), utterances are sharded statically longest-first, and the
decode loop contains no collective at all."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def broadcast_state_dict(sd: Dict[str, torch.Tensor] | None, src: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Rank `src` passes a state dict, the others pass None; everyone returns the same tensors on `device`.
    Floating tensors travel as one flat fp32 arena (one collective), integer buffers as int64 views of it."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: v.to(device) for k, v in sd.items()}
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), "int" if not v.is_floating_point() else "float") for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    sizes = [int(np.prod(s)) if len(s) else 1 for _, s, _ in meta]
    arena = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if rank == src:
        off = 0
        for (k, _, _), n in zip(meta, sizes):
            arena[off:off + n] = sd[k].reshape(-1).to(device, torch.float32)
            off += n
    dist.broadcast(arena, src=src)
    out, off = {}, 0
    for (k, s, kind), n in zip(meta, sizes):
        t = arena[off:off + n].view(s)
        out[k] = t.to(torch.int64) if kind == "int" else t
        off += n
    return out


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Static longest-first assignment of utterance indices to `world` ranks (greedy onto the least-loaded rank;
    ties -> lowest rank).  The only length known before decoding is the text length."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        shards[r].append(i)
        load[r] += int(lengths[i])
    return shards
