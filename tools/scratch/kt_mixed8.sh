#!/bin/bash
# mixed, lists of 65 536 groups: every DP kernel of the last list by stream (what the main stream waits for)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ktm
SPX_PREP_LANES=4 rocprofv3 --kernel-trace -d /tmp/ktm -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --platform mixed --groups-per-step 65536 --distinct 3 --depth 4 --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --verify 0 --guard-exposure 0 --steps 4 --warmup 2 "$@" > /tmp/ktm.json 2>/tmp/ktm.err
python3 - <<'PY'
import csv, glob, json, collections
b = json.loads(open("/tmp/ktm.json").read().strip().splitlines()[-1])
f = glob.glob("/tmp/ktm/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows); lo = t_end - b["ms_per_step"] * 1.3 * 1e6
print("ms/step", b["ms_per_step"], "value", b["value"], b["kernel_ms_per_step"], b["dp_tiers"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo and any(k in r["Kernel_Name"] for k in ("fast_", "baq_", "map_kernel", "score_kernel"))]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(x-t0)/1e6:8.1f} -> {(y-t0)/1e6:8.1f} ({(y-x)/1e6:6.2f}) stream {r["Stream_Id"]:>3} {r["Kernel_Name"][:64]}')
busy = collections.defaultdict(float)
allsel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
for r in allsel: busy[r["Kernel_Name"][:50]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("kernel ms over the window of", (t_end - lo) / 1e6, "ms:")
for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:25]: print(f"   {v:8.1f}  {k}")
PY
