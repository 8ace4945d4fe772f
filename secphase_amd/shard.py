"""Sharding of read groups over ranks and the collectives of a multi-GPU run.

Groups are independent (src/secphase.c:230-351 dispatches one job per group), so rank r scores its own shard with no
data-path collective.  The relabel list is a property of the whole file -- records in file order, tie-breaking draws
consumed in file order (the reference at -@1) -- so two things travel to rank 0 (RCCL over xGMI on GPUs, gloo in the
CPU tests):
  * one 16-byte spx_decision per dispatched group (ONE gather of fixed-size records), and
  * one spx_relabel_rec per candidate group (the few whose best alignment can be a secondary).
Rank 0 merges, replays the draws in global group order and writes the list (spx_gather.cpp).
"""
import ctypes as C

import numpy as np

DECISION_BYTES = 16


def shard_range(n_groups, rank, world):
    """contiguous, balanced by COUNT: rank r owns groups [lo, hi)"""
    lo = n_groups * rank // world
    hi = n_groups * (rank + 1) // world
    return lo, hi


def shard_by_cost(costs, world):
    """contiguous shards balanced by COST (e.g. DP cells per group): boundaries b[0..world] such that rank r
    owns groups [b[r], b[r+1]); returns (boundaries, imbalance = heaviest shard / mean shard)"""
    c = np.asarray(costs, np.float64)
    n = len(c)
    if n == 0:
        return [0] * (world + 1), 1.0
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(cum, total * r / world, side="left")))
    b.append(n)
    for r in range(1, world + 1):
        b[r] = max(b[r], b[r - 1])
    loads = [cum[b[r + 1]] - cum[b[r]] for r in range(world)]
    mean = total / world if total > 0 else 1.0
    return b, (max(loads) / mean if mean > 0 else 1.0)


def _as_bytes_tensor(arr, torch, device):
    """ctypes array / numpy array -> uint8 torch tensor on `device`"""
    a = np.frombuffer(memoryview(arr), np.uint8) if not isinstance(arr, np.ndarray) else arr.view(np.uint8).reshape(-1)
    return torch.from_numpy(a.copy()).to(device)


def gather_bytes(local, dist, torch, dst=0):
    """local: uint8 tensor (any length, same device on every rank).  Returns on dst the list of every rank's bytes
    (rank order), else None.  One all_gather of the lengths (8 bytes each) + one gather of padded buffers."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    pad = torch.zeros(cap, dtype=torch.uint8, device=local.device)
    pad[: local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:s].cpu().numpy() for b, s in zip(bufs, sizes)]


def merge_and_write(api, params, fin, ref, dec_parts, cand_parts, log_path, mode="a"):
    """rank 0: dec_parts / cand_parts = lists (one entry per rank) of raw bytes holding spx_decision / spx_relabel_rec
    arrays.  Replays the draws of every group in global group order on finalizer `fin` (a c_void_p from
    spx_finalizer_create) and appends the relabelled candidates to log_path.  Returns (#decisions, #records written)."""
    L = api.lib()
    dec = np.concatenate([np.frombuffer(p.tobytes(), np.uint8) for p in dec_parts]) if dec_parts else np.zeros(0, np.uint8)
    nd = len(dec) // DECISION_BYTES
    dt = np.dtype([("group", "<u4"), ("n_aln", "i1"), ("prim_idx", "i1"), ("max_idx", "i1"), ("pass_", "u1"),
                   ("tie_mask", "<u2"), ("reserved", "<u2"), ("absdiff", "<i4")])
    d = np.frombuffer(dec.tobytes(), dt, count=nd)
    d = d[d["n_aln"] >= 2]  # device records keep a slot for rejected groups (n_aln 0): they draw nothing
    order = np.argsort(d["group"], kind="stable")
    d = np.ascontiguousarray(d[order])
    nd = len(d)
    best = (C.c_int8 * max(nd, 1))()
    rel = (C.c_int8 * max(nd, 1))()
    darr = (api.Decision * max(nd, 1)).from_buffer_copy(d.tobytes() if nd else bytes(DECISION_BYTES))
    api._chk(L.spx_finalizer_apply_decisions(fin, C.byref(params), darr, nd, best, rel), "spx_finalizer_apply_decisions")
    best_of = dict(zip(d["group"].tolist(), [best[k] for k in range(nd)]))
    rsz = C.sizeof(api.RelabelRec)
    cand = np.concatenate([np.frombuffer(p.tobytes(), np.uint8) for p in cand_parts]) if cand_parts else np.zeros(0, np.uint8)
    nc = len(cand) // rsz
    recs = (api.RelabelRec * max(nc, 1)).from_buffer_copy(cand.tobytes() if nc else bytes(rsz))
    idx = sorted(range(nc), key=lambda k: recs[k].group)
    srt = (api.RelabelRec * max(nc, 1))()
    bsel = (C.c_int8 * max(nc, 1))()
    for j, k in enumerate(idx):
        srt[j] = recs[k]
        bsel[j] = best_of.get(recs[k].group, -1)
    n = L.spx_write_relabel_records(log_path.encode(), mode.encode(), ref, srt, nc, bsel)
    if n < 0:
        raise api.SpxError(n, "spx_write_relabel_records")
    return nd, n
