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


def decide_locally(api, params, fin, out, n, dist, torch, device):
    """Round 3: every rank decides its own groups.  out[0..n) = this rank's collected results of the step (in its shard's
    file order).  The ranks exchange the number of rand() values their groups consume (one all_gather of 8 bytes per
    rank), every rank moves its copy of the stream `fin` over the draws of the lower ranks, finalizes its groups, and moves
    over the higher ranks' draws -- all copies stay at the global position of the stream (the reference at -@1 over the
    groups in (step, rank, group) order)."""
    L = api.lib()
    mine = int(L.spx_count_draws(out, n))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        api._chk(L.spx_finalizer_apply(fin, C.byref(params), out, n), "spx_finalizer_apply")
        return mine
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.tensor([mine], dtype=torch.int64, device=device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    draws = [int(p.item()) for p in parts]
    api._chk(L.spx_finalizer_skip(fin, sum(draws[:rank])), "spx_finalizer_skip")
    api._chk(L.spx_finalizer_apply(fin, C.byref(params), out, n), "spx_finalizer_apply")
    api._chk(L.spx_finalizer_skip(fin, sum(draws[rank + 1:])), "spx_finalizer_skip")
    return mine


def relabel_text(api, batches, ref, out):
    """the fragment of the relabel list that the finalized groups of `batches` (record blocks in order, results out[]
    concatenated the same way) contribute, as a uint8 numpy array"""
    L = api.lib()
    parts = []
    base = 0
    gsz = C.sizeof(api.GroupOut)
    for bp in batches:
        txt, ln = C.c_void_p(), C.c_int64()
        sub = C.cast(C.byref(out, base * gsz), C.POINTER(api.GroupOut))
        api._chk(L.spx_format_relabel_text(bp, ref, sub, C.byref(txt), C.byref(ln)), "spx_format_relabel_text")
        if ln.value:
            parts.append(np.frombuffer(C.string_at(txt, ln.value), np.uint8))
        L.spx_free_text(txt)
        base += bp.contents.n_groups
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def append_fragments(parts, log_path):
    """rank 0: the ranks' fragments of one step, in rank order"""
    with open(log_path, "ab") as f:
        for p in parts:
            if len(p):
                f.write(memoryview(np.ascontiguousarray(p)))


def merge_and_write(api, params, fin, ref, dec_parts, cand_parts, log_path, mode="a"):
    """rank 0: dec_parts / cand_parts = lists (one entry per rank) of raw bytes holding spx_decision / spx_relabel_rec
    arrays.  Replays the draws of every group in global group order on finalizer `fin` (a c_void_p from
    spx_finalizer_create) and appends the relabelled candidates to log_path.  Returns (#decisions, #records written).
    No per-group Python work: at 8 ranks x 131 072 groups per step this runs once per step on rank 0."""
    L = api.lib()
    dec = np.concatenate([np.asarray(p, np.uint8).reshape(-1) for p in dec_parts]) if dec_parts else np.zeros(0, np.uint8)
    nd = len(dec) // DECISION_BYTES
    dt = np.dtype([("group", "<u4"), ("n_aln", "i1"), ("prim_idx", "i1"), ("max_idx", "i1"), ("pass_", "u1"),
                   ("tie_mask", "<u2"), ("reserved", "<u2"), ("absdiff", "<i4")])
    d = dec[: nd * DECISION_BYTES].view(dt)
    d = d[d["n_aln"] >= 2]  # device records keep a slot for rejected groups (n_aln 0): they draw nothing
    grp = d["group"]
    if len(grp) > 1 and np.any(grp[1:] < grp[:-1]):  # rank order = group order already, unless shards interleave
        d = d[np.argsort(grp, kind="stable")]
    d = np.ascontiguousarray(d)
    nd = len(d)
    best = np.zeros(max(nd, 1), np.int8)
    rel = np.zeros(max(nd, 1), np.int8)
    dbuf = d.view(np.uint8) if nd else np.zeros(DECISION_BYTES, np.uint8)
    api._chk(L.spx_finalizer_apply_decisions(fin, C.byref(params), C.cast(dbuf.ctypes.data, C.POINTER(api.Decision)), nd,
                                             C.cast(best.ctypes.data, C.POINTER(C.c_int8)),
                                             C.cast(rel.ctypes.data, C.POINTER(C.c_int8))), "spx_finalizer_apply_decisions")
    rsz = C.sizeof(api.RelabelRec)
    dgrp = d["group"]

    def best_of(cgrp):
        if not nd:
            return np.full(len(cgrp), -1, np.int8)
        at = np.minimum(np.searchsorted(dgrp, cgrp), nd - 1)
        return np.where(dgrp[at] == cgrp, best[at], -1).astype(np.int8)

    def write(rows, cgrp, md):
        nc_ = len(cgrp)
        if nc_ == 0:
            return 0
        bsel = best_of(cgrp)
        n_ = L.spx_write_relabel_records(log_path.encode(), md.encode(), ref, C.cast(rows.ctypes.data, C.POINTER(api.RelabelRec)), nc_,
                                         C.cast(bsel.ctypes.data, C.POINTER(C.c_int8)))
        if n_ < 0:
            raise api.SpxError(n_, "spx_write_relabel_records")
        return n_

    parts = []
    for p in (cand_parts or []):
        a_ = np.ascontiguousarray(np.asarray(p, np.uint8).reshape(-1))
        k = len(a_) // rsz
        if k:
            rows = a_[: k * rsz].reshape(k, rsz)
            parts.append((rows, np.ascontiguousarray(rows[:, :4]).view("<u4").reshape(-1)))
    # contiguous shards in rank order arrive sorted already: written part by part, nothing is copied or re-ordered
    in_order = all(len(g) < 2 or not np.any(g[1:] < g[:-1]) for _, g in parts) and \
        all(parts[i][1][-1] <= parts[i + 1][1][0] for i in range(len(parts) - 1))
    if mode == "w":
        open(log_path, "w").close()
    n = 0
    if in_order:
        for rows, cgrp in parts:
            n += write(rows, cgrp, "a")
    elif parts:
        rows = np.concatenate([r for r, _ in parts])
        cgrp = np.concatenate([g for _, g in parts])
        order = np.argsort(cgrp, kind="stable")
        n += write(np.ascontiguousarray(rows[order]), cgrp[order], "a")
    return nd, n
