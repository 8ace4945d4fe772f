#!/bin/bash
cd /tmp; export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=${Q:-24} rocprofv3 --kernel-trace -d /tmp/ktc -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 4 --warmup 2 > /tmp/ktc.json 2>/tmp/ktc.err
cat /tmp/ktc.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"])"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktc/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print(list(rows[0].keys()))
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if "copyBuffer" in r["Kernel_Name"] and d > 1.0:
        q = r.get("Queue_Id"); prev = next((rows[j]["Kernel_Name"][:50] for j in range(i - 1, -1, -1) if rows[j].get("Queue_Id") == q), None)
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.1f} ms  {d:7.2f} ms  queue {q} stream {r.get("Stream_Id")} grid {r.get("Grid_Size_X", r.get("Grid_Size"))} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size"))}  prev on queue: {prev}')
PY
