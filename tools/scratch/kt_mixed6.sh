#!/bin/bash
# kernel trace of the mixed bench (steady state), pulled back for a timeline of one list's DP kernels
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/kt6 -o run --output-format csv -- python3 $ROOT/bench.py --platform mixed --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 10 --warmup 3 --verify 0 --guard-exposure 0 > $OUT/mixed6_under_rocprof.json 2> $OUT/mixed6_under_rocprof.err)
python3 - $OUT <<'PY'
import csv, sys, gzip
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/kt6/run_kernel_trace.csv")))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Workgroup_Size", "Grid_Size"]
keep = [k for k in keep if k in rows[0]]
t1 = max(int(r["End_Timestamp"]) for r in rows)
t0 = min(int(r["Start_Timestamp"]) for r in rows)
lo = t0 + int(0.55 * (t1 - t0))
with gzip.open(f"{out}/mixed6_trace_tail.csv.gz", "wt") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        if int(r["Start_Timestamp"]) >= lo:
            w.writerow([r[k][:70] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $OUT/kt6
tail -1 $OUT/mixed6_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
ls -la $OUT/mixed6_trace_tail.csv.gz
