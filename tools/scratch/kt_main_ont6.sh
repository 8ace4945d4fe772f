#!/bin/bash
# ONT: timeline of the MAIN stream over the last steps of the pipelined bench (gaps = what the DP kernels wait for)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktm
rocprofv3 --kernel-trace -d /tmp/ktm -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --platform ont --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --verify 0 --guard-exposure 0 --steps 6 --warmup 3 "$@" > /tmp/ktm.json 2>/tmp/ktm.err
python3 - <<'PY'
import csv, glob, json, collections
b = json.loads(open("/tmp/ktm.json").read().strip().splitlines()[-1])
f = glob.glob("/tmp/ktm/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows); lo = t_end - b["ms_per_step"] * 2.2 * 1e6
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
t0 = int(sel[0]["Start_Timestamp"])
print("ms/step", b["ms_per_step"], "value", b["value"], b["kernel_ms_per_step"])
tot = collections.Counter()
for r in sel:
    if "fast_fwd_kernel<4, 28" in r["Kernel_Name"]: tot[r["Stream_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
sid = tot.most_common(1)[0][0]
print("main stream", sid)
prev = None
for r in sel:
    if r["Stream_Id"] != sid: continue
    x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (x - prev) / 1e6 if prev else 0
    print(f'{(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f} ms; gap before {gap:6.2f})  {r["Kernel_Name"][:60]}')
    prev = y
busy = collections.defaultdict(float)
for r in sel: busy[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("busy ms per stream over", (t_end - lo) / 1e6, "ms:", dict(sorted(busy.items(), key=lambda kv: -kv[1])))
# the longest kernels on other streams
oth = sorted((r for r in sel if r["Stream_Id"] != sid), key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:25]
for r in sorted(oth, key=lambda r: int(r["Start_Timestamp"])):
    x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'   other {(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f}) stream {r["Stream_Id"]:>3} {r["Kernel_Name"][:60]}')
PY
