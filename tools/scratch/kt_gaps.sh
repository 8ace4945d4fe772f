#!/bin/bash
P=${1:-ont}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktm
rocprofv3 --kernel-trace -d /tmp/ktm -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --platform $P --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 6 --warmup 3 > /tmp/ktm.json 2>/tmp/ktm.err
python3 - <<'PY'
import csv, glob, json
b = json.loads(open("/tmp/ktm.json").read().strip().splitlines()[-1])
f = glob.glob("/tmp/ktm/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows); lo = t_end - b["ms_per_step"] * 4 * 1e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
t0 = int(sel[0]["Start_Timestamp"])
big = [r for r in sel if "baq_fwd_kernel<4" in r["Kernel_Name"] or "baq_bwd_kernel<4" in r["Kernel_Name"]]
print("ms/step", b["ms_per_step"])
for r in big: print(f'{(int(r["Start_Timestamp"])-t0)/1e6:8.1f} -> {(int(r["End_Timestamp"])-t0)/1e6:8.1f}  stream {r["Stream_Id"]}  {r["Kernel_Name"][:40]}')
# what runs in the gaps of the stream of the main class
main = [r for r in big if "<4, 28" in r["Kernel_Name"]]
sid = main[0]["Stream_Id"]
ms = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if r["Stream_Id"] == sid)
gaps = [(ms[i][1], ms[i + 1][0]) for i in range(len(ms) - 1) if ms[i + 1][0] - ms[i][1] > 5e6]
for a, e in gaps[:4]:
    print(f"gap on the main stream {(a-t0)/1e6:.1f} -> {(e-t0)/1e6:.1f} ms ({(e-a)/1e6:.1f} ms): kernels overlapping it:")
    for r in sel:
        x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if x < e and y > a and (y - x) > 2e6: print(f'      {(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} stream {r["Stream_Id"]} {r["Kernel_Name"][:50]}')
PY
