#!/bin/bash
# kernel trace + stats of the mixed bench in its final round-5 configuration (six side streams, six lanes, eight lists in flight)
export TMPDIR=/tmp SPX_PREP_LANES=6
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt7 -o run --output-format csv -- python3 $ROOT/bench.py --platform mixed --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 16 --warmup 3 --verify 0 --guard-exposure 0 > $OUT/mixed7_under_rocprof.json 2> $OUT/mixed7_under_rocprof.err)
cp $OUT/kt7/run_kernel_stats.csv $OUT/mixed7_kernel_stats.csv
python3 - $OUT <<'PY'
import csv, sys, gzip
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/kt7/run_kernel_trace.csv")))
keep = [k for k in ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"] if k in rows[0]]
t1 = max(int(r["End_Timestamp"]) for r in rows); t0 = min(int(r["Start_Timestamp"]) for r in rows)
lo = t0 + int(0.6 * (t1 - t0))
with gzip.open(f"{out}/mixed7_trace_tail.csv.gz", "wt") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows:
        if int(r["Start_Timestamp"]) >= lo:
            w.writerow([r[k][:70] if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf $OUT/kt7
tail -1 $OUT/mixed7_under_rocprof.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
