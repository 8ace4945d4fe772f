#!/bin/bash
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ktc
rocprofv3 --kernel-trace -d /tmp/ktc -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 4 --warmup 2 > /tmp/ktc.json 2>/tmp/ktc.err
cat /tmp/ktc.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SPX_RESULT_COPY', '$SPX_RESULT_COPY', d[\"value\"], d[\"ms_per_step\"])"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktc/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1e6]
print("copy kernels longer than 1 ms:", len(big), [round(x, 1) for x in big[-6:]])
PY
