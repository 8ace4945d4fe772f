"""CPU, world_size 2 over gloo: reads shard across ranks, one gather of the 8-byte decision
records, rank 0 replays the tie-breaking draws in file order -> same relabel list as one process."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from secphase_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_groups, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import small_genome
    from oracle import orc
    from secphase_amd import records, synth
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=2, read_len=4000)
    p = records.preset("hifi")
    lo, hi = shard.shard_range(n_groups, rank, world)
    r = g.reads(lo, hi - lo)          # this rank's shard only
    _, res = orc.run_batch(r.batch, g.ref, p, threads=1, seed=1)   # stands in for the device path on CPU
    recs = []
    for i, e in enumerate(res):
        if e.n_aln <= 0:
            continue
        sec = [a for a in range(e.n_aln) if a != e.prim_idx]
        mxs = max(e.score[a] for a in sec)
        mx = next(a for a in sec if e.score[a] == mxs)
        tie = sum(1 << a for a in sec if e.score[a] >= mxs)
        ok = not (mxs <= e.score[e.prim_idx] + p.prim_margin_score or mxs < p.min_score)
        recs.append(shard.pack_record(lo + i, e.prim_idx, mx, tie, ok))
    out = shard.gather_records(torch.tensor(recs, dtype=torch.int64), dist, dst=0)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(built, tmp_path):
    n_groups = 24
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, n_groups, str(tmp_path)), nprocs=2, join=True)
    got = shard.unpack_records(np.load(str(tmp_path / "gathered.npy")))
    from common import small_genome
    from oracle import orc
    from secphase_amd import records, synth
    g = small_genome(synth.HIFI, max_secondaries=3, n_paralogs=2, read_len=4000)
    p = records.preset("hifi")
    r = g.reads(0, n_groups)
    _, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1)
    exp = [(i, e) for i, e in enumerate(res) if e.n_aln > 0]
    assert got["group"].tolist() == [i for i, _ in exp]
    assert got["prim_idx"].tolist() == [e.prim_idx for _, e in exp]
    # decisions that do not depend on a draw must agree; (ties are replayed on rank 0 in file order)
    for k, (i, e) in enumerate(exp):
        if bin(int(got["tie_mask"][k])).count("1") == 1:
            best = int(got["max_idx"][k]) if got["passed"][k] else int(got["prim_idx"][k])
            assert best == e.best_idx, i


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
