#!/bin/bash
# timeline of the MAIN stream over the last steps of the default bench: what runs, and what the gaps wait for
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktm
rocprofv3 --kernel-trace -d /tmp/ktm -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --verify 0 --steps 6 --warmup 3 "$@" > /tmp/ktm.json 2>/tmp/ktm.err
python3 - <<'PY'
import csv, glob, json, collections
b = json.loads(open("/tmp/ktm.json").read().strip().splitlines()[-1])
f = glob.glob("/tmp/ktm/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows); lo = t_end - b["ms_per_step"] * 3 * 1e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
t0 = int(sel[0]["Start_Timestamp"])
print("ms/step", b["ms_per_step"], "value", b["value"])
main = [r for r in sel if "fast_fwd_kernel<2, 21, 41" in r["Kernel_Name"] or "baq_fwd1_kernel<41>" in r["Kernel_Name"]]
sid = collections.Counter(r["Stream_Id"] for r in main if "fast" in r["Kernel_Name"]).most_common(1)[0][0] if any("fast" in r["Kernel_Name"] for r in main) else main[0]["Stream_Id"]
print("main stream", sid)
prev = None
for r in sel:
    if r["Stream_Id"] != sid: continue
    x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (x - prev) / 1e6 if prev else 0
    print(f'{(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f} ms; gap before {gap:6.2f})  {r["Kernel_Name"][:60]}')
    prev = y
ms = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if r["Stream_Id"] == sid)
gaps = [(ms[i][1], ms[i + 1][0]) for i in range(len(ms) - 1) if ms[i + 1][0] - ms[i][1] > 8e6]
for a, e in gaps[:2]:
    print(f"gap on the main stream {(a-t0)/1e6:.1f} -> {(e-t0)/1e6:.1f} ms ({(e-a)/1e6:.1f} ms): kernels overlapping [gap - 30 ms, gap end]:")
    for r in sel:
        x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if x < e and y > a - 30e6 and r["Stream_Id"] != sid: print(f'      {(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f}) stream {r["Stream_Id"]:>3} {r["Kernel_Name"][:70]}')
# one preparation lane's kernels over the window
prep = collections.Counter(r["Stream_Id"] for r in sel if "aln_emit" in r["Kernel_Name"])
if prep:
    ps = prep.most_common(1)[0][0]
    print("preparation stream", ps)
    prev = None
    for r in sel:
        if r["Stream_Id"] != ps: continue
        x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (y - x) > 0.3e6 or (prev and x - prev > 1e6):
            print(f'   {(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f} ms; gap before {((x-prev)/1e6 if prev else 0):6.2f})  {r["Kernel_Name"][:70]}')
        prev = y
# per-stream busy time
busy = collections.defaultdict(float)
for r in sel: busy[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("busy ms per stream over", (t_end - lo) / 1e6, "ms:", dict(sorted(busy.items(), key=lambda kv: -kv[1])))
PY
