#!/usr/bin/env python3
"""bench.py -- secphase hot path (marker -> banded-HMM BAQ -> marker-consistency score ->
decision) on N MI355X GPUs of one node.

A "step" is one pass of the device path over one batch of synthetic alignment groups whose
work list (packed query windows, wanted rows, marker tables) and reference are ALREADY resident
in HBM: every banded DP problem of the batch + the scoring/decision kernel + (N>1) one RCCL
gather of the 8-byte decision records to rank 0.  `value` = groups/s over all ranks.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_VECTOR_TFLOPS = 78.6  # MI355X vector FP64, FMA counted as 2 flops (datasheet)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
FLOPS_PER_CELL = 45             # forward 19 + backward 20 + MAP 6 per band cell
FWD_FLOPS_PER_CELL = 19


def pmc_traffic(kernel, gps):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_counters.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this very command, KB units).  The guide's x2
    correction of FETCH_SIZE on gfx950 applies to 16 B/lane streaming reads; for the access pattern of these kernels
    the counters were calibrated on the backward kernel's known read / write volume: factor 1.
    None when the profile does not match."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_counters.json")))
        meta = prof.get("_meta", {})
        if meta.get("groups_per_step") != gps:
            return None
        k = prof.get("void " + kernel) or prof.get(kernel)
        return int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)
    except Exception:
        return None


def gen_parallel(genome, first, n, chunk, threads):
    """generate n groups starting at `first` in `chunk`-sized batches on `threads` threads"""
    from secphase_amd import synth  # noqa: F401
    starts = list(range(first, first + n, chunk))
    out = [None] * len(starts)
    it = iter(range(len(starts)))
    lock = threading.Lock()

    def run():
        while True:
            with lock:
                k = next(it, None)
            if k is None:
                return
            out[k] = genome.reads(starts[k], min(chunk, first + n - starts[k]))

    ths = [threading.Thread(target=run) for _ in range(max(1, threads))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--platform", default="hifi", choices=["hifi", "ont", "mixed"])
    ap.add_argument("--kernel-only", action="store_true", help="(profiling runs) only the device-resident replay")
    ap.add_argument("--groups-per-step", type=int, default=0, help="groups per rank per step (0: preset)")
    ap.add_argument("--chunk", type=int, default=0, help="groups per prepared work list (0: one list per step)")
    ap.add_argument("--gen-chunk", type=int, default=1024, help="groups per generator call (parallel generation)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="groups in the CPU baseline sample (0: preset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=256, help="groups checked against the oracle before timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per stream of the context (before HIP initialises)
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the one-rank-per-GPU job as a CHILD process -- nothing has touched the
        # GPU yet (torch is not even imported), and the launcher is never exec'd -- and relay its JSON line
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from secphase_amd import api, records, synth

    if not os.path.exists(api.LIB_PATH):
        ge.build()
    ge.build_cpu_helpers()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the scoring path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ont = args.platform == "ont"
    mixed = args.platform == "mixed"
    gps = args.groups_per_step or (4096 if ont else 8192 if mixed else 32768)  # HiFi: 32 768 groups per launch amortise the tails of the rare wide classes (+2.4 % over 16 384)
    # config 5 (mixed HiFi+ONT, power-law lengths, <= 8 secondaries) is run as --hifi over the whole mix, SURVEY 8(d)
    params = records.preset("ont", bandwidth=50) if ont else records.preset("hifi")
    cfg = synth.default_cfg(synth.ONT if ont else synth.MIXED if mixed else synth.HIFI)
    # SURVEY section 8(d) assembly: 2 haplotypes x 10 contigs x 5 Mbp (+ paralog copies); resident in HBM as 4-bit codes
    t0 = time.time()
    genome = synth.Genome(cfg)
    t_genome = time.time() - t0
    ctx = api.Context(local_rank)
    ctx.set_reference(genome.ref)

    ncpu = os.cpu_count() or 1
    gen_threads = max(1, min(64, ncpu // max(1, world)))
    first = rank * gps  # every rank scores its own shard of the read groups (weak scaling)
    t0 = time.time()
    chunk = args.chunk or gps
    chunks = gen_parallel(genome, first, gps, args.gen_chunk, gen_threads)
    t_gen = time.time() - t0
    t0 = time.time()
    per = max(1, chunk // args.gen_chunk)
    works = [ctx.prepare([ch.batch for ch in chunks[k:k + per]], params, host_threads=gen_threads)
             for k in range(0, len(chunks), per)]
    t_prep = time.time() - t0
    stats = [w.stats() for w in works]
    n_disp = sum(s.n_dispatched for s in stats)
    n_prob = sum(s.n_problems for s in stats)
    n_rows = sum(s.n_rows for s in stats)
    cells = sum(s.dp_cells for s in stats)
    bytes_in = sum(s.bytes_h2d for s in stats)

    # ---- parity gate on a sample of the very workload being timed (oracle = checker only) ----
    verified = 0
    if rank == 0 and args.verify > 0:
        from oracle import orc
        nchk = min(args.verify, chunks[0].batch.contents.n_groups)
        sub = genome.reads(first, nchk)
        w = ctx.prepare(sub.batch, params, host_threads=gen_threads)
        w.launch()
        out = w.collect(finalize_seed=1)
        _, res = orc.run_batch(sub.batch, genome.ref, params, threads=min(ncpu, 64), seed=1)
        for i in range(nchk):
            o, e = out[i], res[i]
            ok = o.n_aln == e.n_aln and o.best_idx == e.best_idx and all(o.score[a] == e.score[a] for a in range(max(e.n_aln, 0)))
            if not ok:
                sys.exit(f"parity check failed on group {i}: GPU result differs from the oracle")
        verified = nchk
        w.free()

    # decision records land in a torch tensor so that RCCL can gather them
    # (every rank sizes its buffer by the largest dispatched count of any rank: ranks whose dispatch filter drops a
    # different number of groups would otherwise disagree about the gather's message size)
    cap = n_disp
    if world > 1:
        t = torch.tensor([n_disp], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap = int(t.item())
    dec = torch.full((max(cap, 1),), -1, dtype=torch.int64, device="cuda")
    gathered = [torch.zeros_like(dec) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        off = 0
        for k, w in enumerate(works):
            w.launch()
            off += w.pack_decisions(first + k * chunk, dec.data_ptr() + 8 * off, dec.numel() - off)
        if world > 1:
            ctx_sync()
            dist.gather(dec, gathered, dst=0)

    def ctx_sync():
        api._chk(api.lib().spx_sync(ctx.h), "spx_sync")

    for _ in range(args.warmup):
        step()
    ctx_sync()
    if args.warmup > 0:
        works[0].collect(finalize_seed=None)  # closes the event window: the averages below cover the TIMED launches only
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    baq_ms = 0.0
    score_ms = 0.0
    for _ in range(args.steps):
        step()
    ctx_sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n_disp, n_prob, cells], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_disp_all, n_prob_all, cells_all = [int(x) for x in tot.tolist()]
    else:
        n_disp_all, n_prob_all, cells_all = n_disp, n_prob, cells

    # dominant kernel: HIP events recorded by spx_launch on the launch stream around the BAQ kernels of the
    # timed region (hipEventRecord on the ctx stream, not torch's current stream)
    per_launch = []
    for w in works:
        if len(works) > 1:
            w.launch()  # the events belong to the context: with several work lists per step re-run each alone
        w.collect(finalize_seed=None)  # one list per step: averages over the launches of the timed region
        per_launch.append(w.stats())
    baq_ms = sum(p.baq_kernel_ms for p in per_launch)
    score_ms = sum(p.score_kernel_ms for p in per_launch)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_disp_all * args.steps / elapsed
        # ---- roofline of the dominant kernel: the forward kernel of the band class holding most cells.
        # FP64 vector-ALU bound (not HBM, not MFMA): 19 of the 45 flops per band cell are forward flops.
        st0 = per_launch[0]
        G, slots = st0.main_class_lanes, st0.main_class_slots
        kname = f"baq_fwd1_kernel<{slots - 1}>" if G == 1 else f"baq_fwd_kernel<{G}, {slots // G}, 0, false>"
        fwd_ms = sum(p.main_fwd_ms for p in per_launch) / len(per_launch)
        bwd_ms = sum(p.main_bwd_ms for p in per_launch) / len(per_launch)
        cls_cells = sum(p.main_class_cells for p in per_launch) / len(per_launch)
        achieved_tf = FWD_FLOPS_PER_CELL * cls_cells / (fwd_ms * 1e-3) / 1e12
        # whole BAQ phase (forward + backward + MAP, all classes) at the algorithm's 45 flop per cell
        phase_tf = FLOPS_PER_CELL * (cells / len(works)) / (baq_ms / len(works) * 1e-3) / 1e12
        compulsory = (bytes_in + n_rows * 13) / len(works)  # inputs + per-row outputs, per launch
        roofline = {
            "bound": "mfma",  # the compute roof of the two the contract names; see "compute_unit" and "note"
            "compute_unit": "valu_fp64",
            "achieved": round(achieved_tf, 3),
            "peak": PEAK_FP64_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tf / PEAK_FP64_VECTOR_TFLOPS, 4),
            "traffic": pmc_traffic(kname, gps),
            "kernel": kname,
            "avg_launch_ms": round(fwd_ms, 4),
            "launches_averaged": int(st0.n_launches_averaged),
            "cells_per_launch": int(cls_cells),
            "flops_per_cell": FWD_FLOPS_PER_CELL,
            "note": "compute-bound, but on the FP64 VECTOR ALU, not on the matrix cores: the DP has sequential dependences "
                    "inside every row and MFMA's fused rounding would break bit-exactness; MI355X's FP64 MFMA peak equals "
                    "its FP64 vector peak (78.6 TFLOP/s), so the roof is the same number. "
                    "rows live in VGPRs/LDS: FP64 vector-ALU work under the package power limit (the big kernels run at "
                    "1.8-2.0 GHz, ~85% of the VALU issue slots), not HBM/MFMA. peak = vector FP64 at 2.4 GHz with FMA counted "
                    "as 2; the bit-exact path may not fuse mul+add, so 39.3 is the attainable ceiling. algorithmic flops: "
                    "19 (forward) of 45 per band cell, SURVEY 8(d); avg_launch_ms = HIP events on the launch stream, "
                    "averaged over the timed launches (the other band classes run beside it on their own streams)",
            "phase": {"what": "forward + backward + MAP kernels, all band classes, 45 flop per band cell",
                      "achieved": round(phase_tf, 3), "frac": round(phase_tf / PEAK_FP64_VECTOR_TFLOPS, 4),
                      "ms_per_launch": round(baq_ms / len(works), 4),
                      "backward_kernel_ms": round(bwd_ms, 4)},
            "hbm": {"bound": "hbm", "achieved": round(compulsory / (baq_ms / len(works) * 1e-3) / 1e9, 2),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(compulsory / (baq_ms / len(works) * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                    "algorithmic_bytes_per_launch": int(compulsory)},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only: the other ranks would sit in a barrier
            from oracle import orc
            ns = args.cpu_sample or max(256 if ont else 1024, (4 if ont else 16) * ncpu)
            ns = min(ns, gps)
            sample = genome.reads(first, ns)
            # the oracle keeps the reference's memory pattern (two ~0.8 MB matrices zeroed per BAQ call), which
            # saturates the host memory system well before all hardware threads are busy: time it at all
            # threads and at 32, report the better one with the thread count actually used
            best = None
            for cores in sorted({ncpu, min(ncpu, 32)}, reverse=True):
                t0 = time.perf_counter()
                nre, res = orc.run_batch(sample.batch, genome.ref, params, threads=cores, seed=1, reuse_scratch=True)
                dt = time.perf_counter() - t0
                ndis = sum(1 for r in res if r.n_aln > 0)
                cand = {"value": round(ndis / dt, 2), "unit": "groups/s", "cores": cores, "kind": "port",
                        "sample": f"first {ns} groups of the same workload, oracle (C restatement, -O2 -ffp-contract=off, "
                                  f"pthread pool over groups, per-thread DP scratch instead of calloc/free per call), "
                                  f"{dt:.2f} s wall; host has {ncpu} hardware threads",
                        "cells_per_s": round(sum(r.dp_cells for r in res) / dt, 1)}
                if best is None or cand["value"] > best["value"]:
                    best = cand
            cpu = best
        line = {
            "metric": "reads/sec (primary+secondary groups scored)",
            "value": round(value, 2),
            "unit": "groups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("1M ONT reads, 30 kb, <=4 secondaries, band=50, ONT gap params" if ont else
                             "Mixed HiFi+ONT reads, length 2-100 kb power-law, <=8 secondaries, run as --hifi" if mixed else
                             "1M HiFi reads, 15 kb, <=2 secondaries, band=20, BAQ window 500 bp") +
                            f" (streamed as batches of {gps} groups per GPU per step; inputs resident in HBM)",
                "groups_per_step_per_gpu": gps,
                "dp_problems_per_step": n_prob_all,
                "dp_cells_per_step": cells_all,
                "wanted_rows_per_step_rank0": n_rows,
                "parallelism": f"reads sharded over {world} GPU(s); RCCL gather of 8-byte decision records" if world > 1
                               else "single GPU",
                "verified_groups_vs_oracle": verified,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": {"baq": round(baq_ms, 3), "score": round(score_ms, 3)},
            "dp_cells_per_s": round(cells_all * args.steps / elapsed, 1),
            "dp_problems_per_s": round(n_prob_all * args.steps / elapsed, 1),
            "setup_s": {"genome": round(t_genome, 2), "generate": round(t_gen, 2), "host_prepare+h2d": round(t_prep, 2)},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
