#!/usr/bin/env python3
"""round 6: one prepared work list replayed -- DP kernel times of the two-tier path (tools/scratch/kbench6.py [hifi|ont|mixed] [groups] [reps])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from secphase_amd import api, records, synth
plat = sys.argv[1] if len(sys.argv) > 1 else "hifi"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = synth.default_cfg(dict(hifi=synth.HIFI, ont=synth.ONT, mixed=synth.MIXED)[plat], n_contigs=4, contig_len=1000000)
g = synth.Genome(cfg)
par = records.preset("ont", bandwidth=50) if plat == "ont" else records.preset("hifi")
ctx = api.Context(0)
ctx.set_reference(g.ref)
r = g.reads(0, n)
for tiers in ([int(x) for x in os.environ.get("KB_TIERS", "1,0").split(",")]):
    api.set_dp_tiers(tiers)
    w = ctx.prepare(r.batch, par)
    for k in range(2):
        w.launch(); w.sync()
    w.collect()
    t0 = time.time()
    for k in range(reps):
        w.launch()
    w.sync()
    dt = (time.time() - t0) / reps
    out = w.collect()
    st = w.stats()
    print(f"{plat} tiers={tiers}: {n} groups, {st.n_problems} problems, {st.dp_cells/1e9:.2f} Gcells, slices {st.dp_slices}: wall/launch {dt*1e3:.2f} ms, baq {st.baq_kernel_ms:.2f} ms, "
          f"main cls {st.main_class} fwd {st.main_fwd_ms:.2f} bwd {st.main_bwd_ms:.2f} ms ({st.main_class_cells/1e9:.2f} Gcells -> fwd {19*st.main_class_cells/max(st.main_fwd_ms,1e-9)/1e9:.1f} TFLOP/s), score {st.score_kernel_ms:.2f}; "
          f"tiers fast {st.tier_fast_problems} rerun cert/model/range {st.tier_rerun_certificate}/{st.tier_rerun_model}/{st.tier_rerun_range} rows {st.tier_rows_uncertified}", flush=True)
    print("   problems per class:", [int(x) for x in st.problems_per_class][:14])
    w.free()
