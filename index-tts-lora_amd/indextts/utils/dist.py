"""Multi-GPU data parallelism for the inference path (new design; the reference has no multi-GPU inference code,
SURVEY.md §2.1/§8e): one process per GPU, every rank holds a full weight replica received ONCE from rank 0 through
RCCL broadcasts of flat arenas (xGMI inside a node), utterances are sharded statically longest-first, and the decode
loop contains no collective at all."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

# GEMM operands of the decode path: these are stored in the compute dtype on every rank anyway (GPTEngine packs them),
# so rank 0 casts them BEFORE the broadcast and the arena carries 2 bytes per weight instead of 4.
_GPT_GEMM_SUFFIXES = ("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")


def compact_gpt_state_dict(sd: Dict[str, torch.Tensor], dtype: torch.dtype) -> Dict[str, torch.Tensor]:
    """The GPT state dict with its transformer / head GEMM weights cast to `dtype` (what GPTEngine would do at load);
    LayerNorm parameters, biases, embedding tables and the conditioner stay as they are.  Same engine bits as loading
    the original, half the broadcast bytes."""
    if dtype == torch.float32:
        return dict(sd)
    out = {}
    for k, v in sd.items():
        gemm = (k.startswith("gpt.h.") and k.endswith(_GPT_GEMM_SUFFIXES)) or k == "mel_head.weight"
        out[k] = v.to(dtype) if gemm and v.is_floating_point() else v
    return out


def compact_bigvgan_state_dict(sd: Dict[str, torch.Tensor], dtype: torch.dtype) -> Dict[str, torch.Tensor]:
    """The vocoder state dict with weight norm folded in fp32 (w = g v / ||v||, what remove_weight_norm leaves) and the
    folded convolution weights cast to `dtype`; BigVGAN.load_state_dict accepts the plain `.weight` form.  The speaker
    encoder, biases, snake parameters and filters are untouched."""
    from indextts.BigVGAN.models import fold_weight_norm
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_v"):
            continue
        if k.endswith(".weight_g"):
            p = k[: -len(".weight_g")]
            out[p + ".weight"] = fold_weight_norm(v, sd[p + ".weight_v"]).to(dtype)
        else:
            out[k] = v
    return out


def broadcast_state_dict(sd: Dict[str, torch.Tensor] | None, src: int = 0, device="cpu",
                         force_collectives: bool = False) -> Dict[str, torch.Tensor]:
    """Rank `src` passes a state dict, the others pass None; everyone returns the same tensors, dtypes preserved, on
    `device`.  One flat arena and one collective PER DTYPE present (typically fp32 + bf16/fp16 + int64): nothing is
    widened for the wire, integers travel as integers."""
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collectives):
        return {k: v.to(device) for k, v in sd.items()}   # (force_collectives: run the collectives with one rank, a rehearsal)
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    by_dtype: Dict[str, List[int]] = {}
    for i, (_, _, dt) in enumerate(meta):
        by_dtype.setdefault(dt, []).append(i)
    out: Dict[str, torch.Tensor] = {}
    for dt in sorted(by_dtype):
        tdt = getattr(torch, dt)
        wire = torch.uint8 if tdt == torch.bool else tdt
        idx = by_dtype[dt]
        sizes = [int(np.prod(meta[i][1])) if len(meta[i][1]) else 1 for i in idx]
        arena = torch.empty(sum(sizes), dtype=wire, device=device)
        if rank == src:
            off = 0
            for i, n in zip(idx, sizes):
                arena[off:off + n] = sd[meta[i][0]].reshape(-1).to(device, wire)
                off += n
        dist.broadcast(arena.view(torch.uint8), src=src)  # bytes on the wire: every backend moves uint8
        off = 0
        for i, n in zip(idx, sizes):
            t = arena[off:off + n].view(meta[i][1])
            out[meta[i][0]] = t.to(torch.bool) if tdt == torch.bool else t
            off += n
    return {k: out[k] for k, _, _ in meta}


def arena_bytes(sd: Dict[str, torch.Tensor]) -> int:
    return sum(v.numel() * v.element_size() for v in sd.values())


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


def shard_utterances_even(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first assignment with EQUAL shard sizes (len(lengths) must divide by world): the sorted utterances are
    dealt in a serpentine (0..w-1, w-1..0, ...), so every rank decodes the same number of rows (one fixed-size batch
    per rank, BASELINE config 4: 256 utterances -> 32 per GPU) while the total text length per rank stays balanced."""
    n = len(lengths)
    if n % world != 0:
        raise ValueError(f"{n} utterances do not divide over {world} ranks")
    order = sorted(range(n), key=lambda i: (-int(lengths[i]), i))
    shards: List[List[int]] = [[] for _ in range(world)]
    for j, i in enumerate(order):
        rnd, k = divmod(j, world)
        shards[k if rnd % 2 == 0 else world - 1 - k].append(i)
    return shards
