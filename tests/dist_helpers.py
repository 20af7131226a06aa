"""torch.distributed (gloo) restatement of the library's score merge for the CPU tests (tests/_dist_worker.py, world size 2):
what csrc/comm.cpp does with RCCL on the device.  Test infrastructure only - the product package imports no torch."""
from __future__ import annotations


def all_gather_scores(local, sizes=None, group=None):
    """Merge per-rank score shards (1-D tensors) into the full list on every rank.
    sizes: number of scores of every rank (exchanged first when not given)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if sizes is None:
        mine = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        allsz = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(allsz, mine, group=group)
        sizes = [int(x) for x in allsz.tolist()]
    if len(set(sizes)) == 1:
        out = torch.empty(sum(sizes), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    parts = [torch.empty(pad, dtype=local.dtype, device=local.device) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def all_gather_padded(buf, chunk: int, group=None):
    """In-place merge of an item-sharded run: `buf` (1-D, >= world * chunk elements) already holds this
    rank's scores in buf[rank * chunk : (rank + 1) * chunk]; afterwards it holds every rank's slice."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = buf[rank * chunk:(rank + 1) * chunk].clone()  # no aliasing between the collective's input and output
    dist.all_gather_into_tensor(buf[:world * chunk], mine, group=group)
    return buf
