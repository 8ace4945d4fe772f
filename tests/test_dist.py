"""CPU, world_size 2 over gloo: the multi-GPU path that EMITS the relabel list.  Reads shard across ranks by cost
(unequal shard sizes), every rank turns its results into spx_decision + spx_relabel_rec records through the C ABI, two
gathers bring them to rank 0, which replays the tie-breaking draws in global file order and writes out.log -- byte for
byte what one process writes, tie groups included (/root/reference/programs/src/secphase.c:194-217 at -@1).
There is no GPU here: the oracle stands in for the device path (test infrastructure only), everything after the scores
is the product's own code."""
import ctypes as C
import filecmp
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from secphase_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GROUPS = 60


def _setup():
    from common import small_genome
    from secphase_amd import records, synth
    g = small_genome(synth.HIFI, max_secondaries=4, n_paralogs=3, read_len=4000, min_secondaries=0, paralog_snv_rate=0.0002)  # near-identical paralogs: tied secondaries
    p = records.preset("hifi")
    p.prim_margin_score = 5.0  # more relabelled reads
    return g, p


def _results_from_oracle(api, res, params):
    """oracle GroupResult -> the spx_group_out array spx_collect would hand back (stand-in for the device path)"""
    out = (api.GroupOut * max(len(res), 1))()
    for i, e in enumerate(res):
        o = out[i]
        o.n_aln = e.n_aln if e.n_aln > 0 else (e.n_aln if e.n_aln < 0 else 0)
        o.prim_idx = o.max_idx = o.best_idx = -1
        if e.n_aln < 2:
            continue
        for a in range(e.n_aln):
            o.score[a] = e.score[a]
            o.rfe[a] = e.rfe[a]
        o.prim_idx = e.prim_idx
        sec = [a for a in range(e.n_aln) if a != e.prim_idx]
        mxs = max(e.score[a] for a in sec)
        o.max_idx = next(a for a in sec if e.score[a] == mxs)
        o.tie_mask = sum(1 << a for a in sec if e.score[a] >= mxs)
        o.pass_ = 0 if (mxs <= e.score[e.prim_idx] + params.prim_margin_score or mxs < params.min_score) else 1
    return out


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from secphase_amd import api
    g, p = _setup()
    # shards by cost: every rank computes the same boundaries from the same cost vector (here: read length as a proxy)
    whole = g.reads(0, N_GROUPS)
    b = whole.batch.contents
    cost = [sum(b.l_qseq[a] for a in range(b.grp_first[k], b.grp_first[k + 1])) ** 1.0 * (1 + 7 * (k < N_GROUPS // 3))
            for k in range(N_GROUPS)]
    bounds, imb = shard.shard_by_cost(cost, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    r = g.reads(lo, hi - lo)  # this rank's shard only
    _, res = orc.run_batch(r.batch, g.ref, p, threads=1, seed=1)
    out = _results_from_oracle(api, res, p)
    n = hi - lo
    L = api.lib()
    dec = (api.Decision * max(n, 1))()
    nd = L.spx_decisions_from_results(out, n, lo, dec, n)
    assert nd >= 0
    nc = L.spx_relabel_candidates(r.batch, lo, out, C.byref(p), None, 0)
    cand = (api.RelabelRec * max(nc, 1))()
    assert L.spx_relabel_candidates(r.batch, lo, out, C.byref(p), cand, nc) == nc
    dev = torch.device("cpu")
    dparts = shard.gather_bytes(torch.from_numpy(np.frombuffer(memoryview(dec), np.uint8)[: nd * 16].copy()), dist, torch)
    cparts = shard.gather_bytes(torch.from_numpy(np.frombuffer(memoryview(cand), np.uint8)[: nc * C.sizeof(api.RelabelRec)].copy()),
                                dist, torch)
    if rank == 0:
        fin = C.c_void_p()
        api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
        log = os.path.join(tmp, "dist.out.log")
        open(log, "w").close()
        ndec, nrec = shard.merge_and_write(api, p, fin, g.ref, dparts, cparts, log)
        L.spx_finalizer_free(fin)
        with open(os.path.join(tmp, "info.txt"), "w") as f:
            f.write(f"{ndec} {nrec} {bounds[1]} {imb}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_emit_the_same_relabel_list_as_one_process(built, tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import orc
    g, p = _setup()
    r = g.reads(0, N_GROUPS)
    log_o = str(tmp_path / "one.out.log")
    nre, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    ndec, nrec, cut, imb = open(str(tmp_path / "info.txt")).read().split()
    assert int(cut) != N_GROUPS // 2  # the two shards are NOT equally long
    assert int(ndec) == sum(1 for e in res if e.n_aln >= 2)
    assert nre > 3 and int(nrec) == nre
    # tie groups are part of the comparison: at least one group draws twice
    ties = 0
    for e in res:
        if e.n_aln >= 2:
            sec = [a for a in range(e.n_aln) if a != e.prim_idx]
            mxs = max(e.score[a] for a in sec)
            ties += sum(1 for a in sec if e.score[a] >= mxs) > 1
    assert ties > 0
    assert filecmp.cmp(log_o, str(tmp_path / "dist.out.log"), shallow=False)


def test_decisions_replay_equals_result_replay(built):
    """spx_finalizer_apply_decisions over packed 16-byte records == spx_finalizer_apply over full results, also with a
    non-zero prim_margin_random (coin flips) -- the two share one rule, this pins the record layout"""
    from oracle import orc
    from secphase_amd import api
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    g, p = _setup()
    r = g.reads(0, 40)
    L = api.lib()
    for pmr in (0.0, 30.0):
        p.prim_margin_random = pmr
        _, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1)
        out = _results_from_oracle(api, res, p)
        dec = (api.Decision * 40)()
        nd = L.spx_decisions_from_results(out, 40, 0, dec, 40)
        f1, f2 = C.c_void_p(), C.c_void_p()
        L.spx_finalizer_create(7, C.byref(f1))
        L.spx_finalizer_create(7, C.byref(f2))
        best = (C.c_int8 * 40)()
        rel = (C.c_int8 * 40)()
        api._chk(L.spx_finalizer_apply_decisions(f1, C.byref(p), dec, nd, best, rel), "decisions")
        api._chk(L.spx_finalizer_apply(f2, C.byref(p), out, 40), "results")
        k = 0
        for i in range(40):
            if out[i].n_aln >= 2:
                assert dec[k].group == i and best[k] == out[i].best_idx and rel[k] == out[i].relabel, i
                k += 1
        assert k == nd
        L.spx_finalizer_free(f1)
        L.spx_finalizer_free(f2)
    p.prim_margin_random = 0.0


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    rng = np.random.default_rng(5)
    cost = rng.pareto(1.2, 5000) + 1
    for w in (2, 4, 8):
        b, imb = shard.shard_by_cost(cost, w)
        assert b[0] == 0 and b[-1] == 5000 and all(b[i] <= b[i + 1] for i in range(w))
        assert imb < 1.25  # heaviest shard within 25 % of the mean on a heavy-tailed cost vector
    assert shard.shard_by_cost([], 4)[0] == [0, 0, 0, 0, 0]


def _gpu_worker(rank, world, port, tmp):
    """two ranks sharing ONE MI355X (the pool's boxes have one): each scores its shard on the device through the
    pipeline, packs its 16-byte decision records with the device kernel, builds its candidate records; the collectives
    run over gloo here (RCCL needs one GPU per rank) -- everything else is the multi-GPU path of bench.py"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
    torch.cuda.is_available()  # torch's HIP runtime first (see conftest.py)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from secphase_amd import api
    g, p = _setup()
    whole = g.reads(0, N_GROUPS)
    b = whole.batch.contents
    cost = [sum(b.l_qseq[a] for a in range(b.grp_first[k], b.grp_first[k + 1])) * (1 + 7 * (k < N_GROUPS // 3)) for k in range(N_GROUPS)]
    bounds, _ = shard.shard_by_cost(cost, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    n = hi - lo
    r = g.reads(lo, n)
    ctx = api.Context(0)
    ctx.set_reference(g.ref)
    w = ctx.stage(r.batch, p)
    pipe = api.Pipe(ctx, p, depth=2, host_threads=4)
    pipe.submit(staged=w)
    out, got = pipe.next()
    assert got == n
    dev = torch.zeros(max(n, 1) * 16, dtype=torch.uint8, device="cuda")
    nd = w.pack_decisions(lo, dev.data_ptr(), max(n, 1))
    L = api.lib()
    nc = L.spx_relabel_candidates(r.batch, lo, out, C.byref(p), None, 0)
    cand = (api.RelabelRec * max(nc, 1))()
    assert L.spx_relabel_candidates(r.batch, lo, out, C.byref(p), cand, nc) == nc
    dparts = shard.gather_bytes(dev[: nd * 16].cpu(), dist, torch)
    cparts = shard.gather_bytes(torch.from_numpy(np.frombuffer(memoryview(cand), np.uint8)[: nc * C.sizeof(api.RelabelRec)].copy()), dist, torch)
    if rank == 0:
        fin = C.c_void_p()
        api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
        log = os.path.join(tmp, "dist_gpu.out.log")
        open(log, "w").close()
        ndec, nrec = shard.merge_and_write(api, p, fin, g.ref, dparts, cparts, log)
        L.spx_finalizer_free(fin)
        open(os.path.join(tmp, "info_gpu.txt"), "w").write(f"{ndec} {nrec} {bounds[1]}\n")
    pipe.close()
    w.free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_device_emit_the_same_relabel_list(built, tmp_path):
    """the multi-GPU path with the scores coming from the HIP kernels: unequal shards, device-packed decision records,
    tie groups -- rank 0's out.log is byte-identical to the oracle's single-process list"""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import orc
    g, p = _setup()
    r = g.reads(0, N_GROUPS)
    log_o = str(tmp_path / "one.out.log")
    nre, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    ndec, nrec, cut = open(str(tmp_path / "info_gpu.txt")).read().split()
    assert int(cut) != N_GROUPS // 2 and int(ndec) == sum(1 for e in res if e.n_aln >= 2) and int(nrec) == nre
    assert filecmp.cmp(log_o, str(tmp_path / "dist_gpu.out.log"), shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("world,platform,gps", [(2, "mixed", 2048), (8, "hifi", 1024)], ids=["2ranks-mixed", "8ranks-hifi"])
def test_bench_ranks_write_the_list_one_process_writes(built, tmp_path, world, platform, gps):
    """`python bench.py --gpus N` itself -- the self-launch (a child started before anything touches the GPU), cost-cut
    shards of the mixed workload, every rank deciding and formatting its own groups, the gather of the fragments, rank 0's
    writer thread -- on a one-GPU box (ranks share the device, collectives over gloo: a rig, never a measurement).  The list
    rank 0 wrote over ALL its steps must equal what ONE process writes for the same groups in (step, rank, group) order with
    one rand() stream; bench's own checks against the oracle (timed groups, list prefix) must have run on rank 0."""
    import json
    import subprocess
    from secphase_amd import api, records, synth
    keep = str(tmp_path / "bench2.out.log")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dist-backend", "gloo", "--platform", platform, "--steps", "2", "--warmup", "1",
           "--groups-per-step", str(gps), "--keep-log", keep, "--no-build", "--no-host-input-leg"]
    if world == 8:
        cmd += ["--distinct", "2"]  # fewer distinct batches than the pipeline is deep (what 8 ranks under one memory limit get): batches staged twice
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, env=dict(os.environ, GPU_MAX_HW_QUEUES="14"))
    assert p.returncode == 0, (p.stdout[-400:], p.stderr[-1200:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    assert line["n_gpus"] == world and line["config"]["verified_timed_groups"] >= 256
    assert line["config"]["verified_own_relabel_list"]["oracle_list_is_byte_prefix_of_this_runs_list"] is True
    if platform == "mixed":
        assert line["config"]["sharding"]["imbalance_by_cost"] < line["config"]["sharding"]["imbalance_by_count"] + 1e-9
    # the N > 1 line carries SURVEY 8(d)'s metric too: the command line on the job's devices, its list checked against the oracle
    assert line["from_bam"]["rc"] == 0 and line["from_bam"]["out_log_identical_to_oracle"] is True and line["metric_8d"]["groups_per_s"] > 0
    man = [json.load(open(f"{keep}.rank{r}.json")) for r in range(world)]
    assert all(m["sequence"] == man[0]["sequence"] for m in man) and len(man[0]["sequence"]) >= 3 + 2
    g = synth.Genome(synth.default_cfg(synth.MIXED if platform == "mixed" else synth.HIFI))
    par = records.preset("hifi")
    ctx = api.Context(0)
    ctx.set_reference(g.ref)
    L = api.lib()
    fin = C.c_void_p()
    api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    one = str(tmp_path / "one_process.out.log")
    open(one, "w").close()
    cache = {}
    n_tot = 0
    for i in man[0]["sequence"]:
        for r in range(world):
            for start, n in man[r]["ranges"][str(i)]:
                if (start, n) not in cache:
                    reads = g.reads(start, n)
                    out, _ = ctx.score_batch(reads.batch, par, finalize_seed=None)
                    cache[(start, n)] = (reads, out)
                reads, out = cache[(start, n)]
                api._chk(L.spx_finalizer_apply(fin, C.byref(par), out, n), "spx_finalizer_apply")
                api.write_relabel_log(one, reads.batch, g.ref, out, mode="a")
                n_tot += n
    L.spx_finalizer_free(fin)
    ctx.close()
    assert n_tot == world * gps * len(man[0]["sequence"])
    assert os.path.getsize(one) > 1000 and filecmp.cmp(one, keep, shallow=False)


def _nccl_worker(rank, world, port, tmp):
    """world size 1 over backend "nccl" (= RCCL): the collectives bench.py issues at N > 1 -- all_reduce of the step statistics,
    all_gather of the draw counts, gather of byte fragments and of the device-packed decision records -- on DEVICE tensors"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from oracle import orc
    from secphase_amd import api
    g, p = _setup()
    r = g.reads(0, N_GROUPS)
    L = api.lib()
    ctx = api.Context(0)
    ctx.set_reference(g.ref)
    w = ctx.stage(r.batch, p)
    pipe = api.Pipe(ctx, p, depth=2, host_threads=4)
    pipe.submit(staged=w)
    out, got = pipe.next()
    assert got == N_GROUPS
    # (1) the statistics reductions of bench.py
    t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [1.5, 2.0]
    # (2) device-packed decision records through the gather
    dev = torch.zeros(N_GROUPS * 16, dtype=torch.uint8, device="cuda")
    nd = w.pack_decisions(0, dev.data_ptr(), N_GROUPS)
    dparts = shard.gather_bytes(dev[: nd * 16], dist, torch)
    assert len(dparts) == 1 and len(dparts[0]) == nd * 16
    nc = L.spx_relabel_candidates(r.batch, 0, out, C.byref(p), None, 0)
    cand = (api.RelabelRec * max(nc, 1))()
    assert L.spx_relabel_candidates(r.batch, 0, out, C.byref(p), cand, nc) == nc
    cparts = shard.gather_bytes(torch.from_numpy(np.frombuffer(memoryview(cand), np.uint8)[: nc * C.sizeof(api.RelabelRec)].copy()).to("cuda"), dist, torch)
    fin = C.c_void_p()
    api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    log = os.path.join(tmp, "nccl_merge.out.log")
    open(log, "w").close()
    shard.merge_and_write(api, p, fin, g.ref, dparts, cparts, log)
    L.spx_finalizer_free(fin)
    # (3) the round-3 path: every rank decides its own groups (draw counts all-gathered on the device), one gather of the text
    api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    shard.decide_locally(api, p, fin, out, N_GROUPS, dist, torch, "cuda")
    frag = shard.relabel_text(api, [r.batch], g.ref, out)
    parts = shard.gather_bytes(torch.from_numpy(frag.copy()).to("cuda"), dist, torch)
    log2 = os.path.join(tmp, "nccl_local.out.log")
    open(log2, "w").close()
    shard.append_fragments(parts, log2)
    L.spx_finalizer_free(fin)
    pipe.close()
    w.free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_branch_runs_at_world_size_one(built, tmp_path):
    """backend "nccl" IS RCCL on ROCm: the box has one GPU, so the process group has one rank -- but every collective bench.py and
    secphase_amd/shard.py issue at N > 1 runs through RCCL on device tensors here (the driver's 8-GPU run is the first time they
    see a peer); both ways of producing the list equal the oracle's"""
    port = 35500 + os.getpid() % 2000
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    from oracle import orc
    g, p = _setup()
    r = g.reads(0, N_GROUPS)
    log_o = str(tmp_path / "one.out.log")
    nre, _ = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    assert nre > 3
    assert filecmp.cmp(log_o, str(tmp_path / "nccl_merge.out.log"), shallow=False)
    assert filecmp.cmp(log_o, str(tmp_path / "nccl_local.out.log"), shallow=False)


def test_relabel_record_writer_on_threads_keeps_order_and_format(built, tmp_path):
    """spx_write_relabel_records formats slices of >= 4 096 records on threads: the file must be what one pass of
    print_alignment_scores (src/secphase.c:32-57) over the records in order would write"""
    import ctypes as C

    import numpy as np

    from common import small_genome
    from secphase_amd import api, synth
    L = api.lib()
    g = small_genome(synth.HIFI)
    ref = g.ref.contents
    names = [C.string_at(ref.names + ref.name_off[i]).decode() for i in range(ref.n_contigs)]
    rng = np.random.default_rng(5)
    n = 9000
    recs = (api.RelabelRec * n)()
    best = (C.c_int8 * n)()
    want = []
    for k in range(n):
        r = recs[k]
        r.group = k
        na = int(rng.integers(2, 6))
        r.n_aln, r.prim_idx = na, int(rng.integers(0, na))
        r.qname = f"read_{k}_{int(rng.integers(0, 10 ** 9))}".encode()
        for i in range(na):
            r.score[i] = float(np.round(rng.normal(-50, 40), 3))
            r.rfe[i] = int(rng.integers(0, 10 ** 6))
            r.pos[i] = int(rng.integers(0, 10 ** 6))
            r.tid[i] = int(rng.integers(0, len(names)))
            r.flag[i] = 0 if i == r.prim_idx else 256
        b = int(rng.integers(-1, na))
        best[k] = b
        if b < 0 or b == r.prim_idx:
            continue
        want.append("#MARKER SCORE\n$\t%s\n" % r.qname.decode())
        for i in range(na):
            tag = "*" if not (r.flag[i] & 256) else ("@" if i == b else "!")
            want.append("%s\t%.2f\t%s\t%d\t%d\n" % (tag, r.score[i], names[r.tid[i]], r.pos[i], r.rfe[i]))
        want.append("\n")
    path = str(tmp_path / "records.log")
    nw = L.spx_write_relabel_records(path.encode(), b"w", g.ref, recs, n, best)
    assert nw == sum(1 for x in want if x.startswith("#MARKER"))
    assert open(path).read() == "".join(want)


def test_relabel_log_writer_on_threads_equals_the_oracles_log(built, tmp_path):
    """spx_write_relabel_log formats batches of >= 4 096 groups on threads: fed with the oracle's results it must write
    the oracle's own log byte for byte"""
    import filecmp

    from common import small_genome
    from oracle import orc
    from secphase_amd import api, records, synth
    g = small_genome(synth.HIFI, read_len=300, max_secondaries=3, min_secondaries=1, n_paralogs=3)
    n = 5000
    r = g.reads(0, n)
    par = records.preset("hifi")
    log_o, log_s = str(tmp_path / "oracle.log"), str(tmp_path / "spx.log")
    _, res = orc.run_batch(r.batch, g.ref, par, threads=8, seed=1, log_path=log_o)
    out = (api.GroupOut * n)()
    nrel = 0
    for k in range(n):
        e, o = res[k], out[k]
        o.n_aln, o.prim_idx, o.best_idx, o.relabel = e.n_aln, e.prim_idx, e.best_idx, int(bool(e.relabel))
        nrel += int(bool(e.relabel))
        for a in range(max(e.n_aln, 0)):
            o.score[a], o.rfe[a] = e.score[a], e.rfe[a]
    assert nrel > 50
    api.write_relabel_log(log_s, r.batch, g.ref, out)
    assert filecmp.cmp(log_o, log_s, shallow=False)


def test_merge_and_write_with_interleaved_shards(built, tmp_path):
    """shards that are NOT contiguous in rank order (round-robin blocks): merge_and_write has to sort decisions and
    candidate records by group before the replay; the list equals the single-process one"""
    import filecmp

    import numpy as np

    from oracle import orc
    from secphase_amd import api, shard
    g, p = _setup()
    n = N_GROUPS
    r = g.reads(0, n)
    L = api.lib()
    log_o, log_s = str(tmp_path / "oracle.log"), str(tmp_path / "merged.log")
    _, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    out = _results_from_oracle(api, res, p)
    dparts, cparts = [], []
    for rank in range(3):  # blocks of 7 groups dealt round-robin to three "ranks"
        dec_bytes, cand_bytes = [], []
        for lo in range(rank * 7, n, 21):
            hi = min(lo + 7, n)
            sub = g.reads(lo, hi - lo)
            sub_out = (api.GroupOut * (hi - lo))(*[out[k] for k in range(lo, hi)])
            dec = (api.Decision * (hi - lo))()
            nd = L.spx_decisions_from_results(sub_out, hi - lo, lo, dec, hi - lo)
            dec_bytes.append(np.frombuffer(memoryview(dec), np.uint8)[: nd * shard.DECISION_BYTES].copy())
            nc = L.spx_relabel_candidates(sub.batch, lo, sub_out, C.byref(p), None, 0)
            if nc > 0:
                arr = (api.RelabelRec * nc)()
                L.spx_relabel_candidates(sub.batch, lo, sub_out, C.byref(p), arr, nc)
                cand_bytes.append(np.frombuffer(memoryview(arr), np.uint8).copy())
        dparts.append(np.concatenate(dec_bytes) if dec_bytes else np.zeros(0, np.uint8))
        cparts.append(np.concatenate(cand_bytes) if cand_bytes else np.zeros(0, np.uint8))
    fin = C.c_void_p()
    api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    nd, nw = shard.merge_and_write(api, p, fin, g.ref, dparts, cparts, log_s, mode="w")
    L.spx_finalizer_free(fin)
    assert nd == sum(1 for e in res if e.n_aln >= 2)
    assert filecmp.cmp(log_o, log_s, shallow=False)


# ---------------------------------------------------------------- round 3: decisions made where the groups are
def _local_worker(rank, world, port, tmp):
    """what bench.py does per step at N > 1: draw counts exchanged, every rank decides and formats its own groups, ONE
    gather of the text fragments, rank 0 appends them.  Three steps of unequal shards: the stream position carries over."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import orc
    from secphase_amd import api
    g, p = _setup()
    L = api.lib()
    fin = C.c_void_p()
    api._chk(L.spx_finalizer_create(1, C.byref(fin)), "spx_finalizer_create")
    log = os.path.join(tmp, "local.out.log")
    if rank == 0:
        open(log, "w").close()
    # global order = (step, rank, group): step s covers groups [s*20, s*20+20), cut unevenly between the ranks
    cuts = [(0, 7, 20), (20, 33, 40), (40, 41, 60)]
    for lo, mid, hi in cuts:
        a, b = (lo, mid) if rank == 0 else (mid, hi)
        r = g.reads(a, b - a)
        _, res = orc.run_batch(r.batch, g.ref, p, threads=1, seed=1)
        out = _results_from_oracle(api, res, p)
        shard.decide_locally(api, p, fin, out, b - a, dist, torch, torch.device("cpu"))
        frag = shard.relabel_text(api, [r.batch], g.ref, out)
        parts = shard.gather_bytes(torch.from_numpy(frag.copy()), dist, torch)
        if rank == 0:
            shard.append_fragments(parts, log)
    L.spx_finalizer_free(fin)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_deciding_their_own_groups_emit_the_same_list(built, tmp_path):
    port = 33500 + os.getpid() % 2000
    mp.spawn(_local_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import orc
    g, p = _setup()
    r = g.reads(0, N_GROUPS)
    log_o = str(tmp_path / "one.out.log")
    nre, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1, log_path=log_o)
    ties = 0
    for e in res:
        if e.n_aln >= 2:
            sec = [a for a in range(e.n_aln) if a != e.prim_idx]
            mxs = max(e.score[a] for a in sec)
            ties += sum(1 for a in sec if e.score[a] >= mxs) > 1
    assert ties > 0 and nre > 3
    assert filecmp.cmp(log_o, str(tmp_path / "local.out.log"), shallow=False)


def test_draw_count_and_skip(built):
    """spx_count_draws = the number of values spx_finalizer_apply consumes; spx_finalizer_skip moves a second copy of the
    stream by exactly that much (checked through the next draw of both)"""
    from oracle import orc
    from secphase_amd import api
    g, p = _setup()
    r = g.reads(0, 40)
    L = api.lib()
    L.spx_finalizer_draw.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    _, res = orc.run_batch(r.batch, g.ref, p, threads=2, seed=1)
    out = _results_from_oracle(api, res, p)
    n = L.spx_count_draws(out, 40)
    assert n >= sum(1 for e in res if e.n_aln >= 2)
    f1, f2 = C.c_void_p(), C.c_void_p()
    L.spx_finalizer_create(1, C.byref(f1))
    L.spx_finalizer_create(1, C.byref(f2))
    api._chk(L.spx_finalizer_apply(f1, C.byref(p), out, 40), "apply")
    api._chk(L.spx_finalizer_skip(f2, n), "skip")
    a, b = C.c_int32(), C.c_int32()
    L.spx_finalizer_draw(f1, C.byref(a))
    L.spx_finalizer_draw(f2, C.byref(b))
    assert a.value == b.value
    L.spx_finalizer_free(f1)
    L.spx_finalizer_free(f2)
