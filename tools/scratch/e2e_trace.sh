#!/bin/bash
# kernel trace of the command line on a synthetic BAM: what the inflate kernels do beside the DP kernels
G=${1:-131072}; W=${2:-10}
python tools/e2e_cli.py --groups $G --batch 16384 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
rm -rf $D/out
cd /tmp; export TMPDIR=/tmp
SPX_TIDY_EXIT=1 SPX_TIMING=1 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/e2e_trace -o cli --output-format csv -- $GRAFT_REPO_ROOT/secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 16384 -@ 16 --gpuInflate $W 2> /tmp/e2e.err > /dev/null
grep -h "scoring loop\|inflate chunks" /tmp/e2e.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/e2e_trace/**/cli_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
inf = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0))) for r in rows if "bgzf_inflate" in r["Kernel_Name"]]
dp = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows if "baq_" in r["Kernel_Name"]]
cb = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "copyBuffer" in r["Kernel_Name"]]
print("copy KERNELS:", len(cb), "total ms", round(sum(cb), 1), "of them > 1 ms:", sum(1 for x in cb if x > 1))
import statistics as st
d = [(b - a) / 1e6 for a, b, _ in inf]
print("inflate kernels", len(inf), "duration ms: median", round(st.median(d), 2), "mean", round(st.mean(d), 2), "max", round(max(d), 2), "grid", inf[0][2])
span = (max(b for _, b, _ in inf) - min(a for a, _, _ in inf)) / 1e6
print("span ms", round(span, 1), "sum of durations ms", round(sum(d), 1), "=> average concurrency", round(sum(d) / span, 2))
# union of DP kernel time
ev = sorted(dp); cur_a, cur_b, tot = ev[0][0], ev[0][1], 0
for a, b in ev[1:]:
    if a > cur_b: tot += cur_b - cur_a; cur_a, cur_b = a, b
    else: cur_b = max(cur_b, b)
tot += cur_b - cur_a
print("DP kernels: union of their time ms", round(tot / 1e6, 1))
# inflate durations when no DP kernel overlaps vs overlapping
def overlaps(a, b): return any(x < b and y > a for x, y in dp)
alone = [(b - a) / 1e6 for a, b, _ in inf if not overlaps(a, b)]
both = [(b - a) / 1e6 for a, b, _ in inf if overlaps(a, b)]
print("inflate kernel alone:", len(alone), round(st.mean(alone), 2) if alone else None, "beside DP:", len(both), round(st.mean(both), 2) if both else None)
PY
rm -rf $D gpurun_out/e2e_trace
