#!/bin/bash
# refresh of the round's profile files that depend on the final build (bench lines of the ONT / mixed workloads, kernel trace of the default bench)
set -u
ROOT=$(pwd); OUT=gpurun_out/profiles; TAG=r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python3 bench.py --platform ont --no-from-bam --no-build > "$OUT/${TAG}_bench_ont.json" 2> "$OUT/${TAG}_bench2.err"
python3 bench.py --platform mixed --no-from-bam --no-build > "$OUT/${TAG}_bench_mixed.json" 2>> "$OUT/${TAG}_bench2.err"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2>> "$ROOT/$OUT/${TAG}_bench2.err")
cp "$OUT/kt/run_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp $OUT/kt/*/run_kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys, glob
out, tag = sys.argv[1], sys.argv[2]
b = json.loads(open(f"{out}/{tag}_bench_under_rocprof.json").read().strip().splitlines()[-1])
k = b["roofline"]["kernel"]
f = (glob.glob(f"{out}/kt/run_kernel_trace.csv") + glob.glob(f"{out}/kt/*/run_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void " + k)]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = b["steps"] * b["roofline"].get("kernel_launches_per_step", 1)
json.dump({"kernel": k, "launches_in_trace": len(d), "all_launches_avg_ms": round(sum(d) / max(len(d), 1), 3),
           "timed_launches_avg_ms": round(sum(d[-n:]) / max(len(d[-n:]), 1), 3), "timed_launches_ms": [round(x, 2) for x in d[-n:]],
           "bench_avg_launch_ms_hip_events": b["roofline"]["avg_launch_ms"], "bench_value": b["value"],
           "note": "rocprofv3 --kernel-trace of the default bench command: durations of the dominant kernel; the last `steps` launches "
                   "are the timed steps, the ones before are set-up and warm-up (other concurrency)"},
          open(f"{out}/{tag}_kernel_trace_dominant.json", "w"), indent=1)
PY
rm -rf "$OUT/kt"
python3 tools/pmc_collect.py --platform mixed --out "$OUT/${TAG}_counters_mixed.json" >> "$OUT/${TAG}_bench2.err" 2>&1
for f in ont mixed; do python3 -c "
import json;d=json.loads(open('$OUT/${TAG}_bench_$f.json').read().strip().splitlines()[-1]);print('$f',d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline'].get('alone_frac'),d['cpu_baseline']['value'])"; done
cat $OUT/${TAG}_kernel_trace_dominant.json | head -8
