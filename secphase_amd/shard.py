"""Sharding of read groups over ranks + the single gather of decision records.

Groups are independent (no state flows between reads: secphase.c:230-351 dispatches one
job per group), so rank r scores a contiguous shard and rank 0 gathers fixed-size 8-byte
decision records with ONE collective (RCCL over xGMI on GPUs, gloo in CPU tests).
Record layout = spx_pack_decisions (include/spx.h).
"""
import numpy as np


def shard_range(n_groups, rank, world):
    """contiguous, balanced: rank r owns groups [lo, hi)"""
    lo = n_groups * rank // world
    hi = n_groups * (rank + 1) // world
    return lo, hi


def pack_record(group, prim_idx, max_idx, tie_mask, passed):
    r = (group & 0xffffffff) | ((prim_idx & 0xff) << 32) | ((max_idx & 0xff) << 40) | ((tie_mask & 0x7fff) << 48)
    r |= (1 if passed else 0) << 63
    return r - (1 << 64) if r >= (1 << 63) else r  # as int64


def unpack_records(arr):
    a = np.asarray(arr).astype(np.int64).view(np.uint64)
    return dict(group=(a & np.uint64(0xffffffff)).astype(np.int64),
                prim_idx=((a >> np.uint64(32)) & np.uint64(0xff)).astype(np.int8),
                max_idx=((a >> np.uint64(40)) & np.uint64(0xff)).astype(np.int8),
                tie_mask=((a >> np.uint64(48)) & np.uint64(0x7fff)).astype(np.uint16),
                passed=((a >> np.uint64(63)) & np.uint64(1)).astype(bool))


def gather_records(local, dist, dst=0):
    """local: int64 torch tensor of this rank's records (any length). Returns on dst the
    concatenation ordered by global group index, else None.  One all_gather of the sizes (tiny)
    + one gather of padded records."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    pad = torch.full((cap,), -1, dtype=torch.int64, device=local.device)
    pad[: local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    parts = [b[: int(s.item())] for b, s in zip(bufs, sizes)]
    allr = torch.cat(parts).cpu().numpy()
    order = np.argsort(unpack_records(allr)["group"], kind="stable")
    return allr[order]
