#!/bin/bash
P=${1:-mixed}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/ktm -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --platform $P --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 8 --warmup 4 > /tmp/ktm.json 2>/tmp/ktm.err
python3 - <<'PY'
import csv, glob, json, collections
b = json.loads(open("/tmp/ktm.json").read().strip().splitlines()[-1])
print("value", b["value"], "ms/step", b["ms_per_step"], b["kernel_ms_per_step"])
f = glob.glob("/tmp/ktm/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
win = b["ms_per_step"] * 6 * 1e6   # the last 6 steps
lo = t_end - win
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo]
tot = collections.defaultdict(float); cnt = collections.Counter(); mx = collections.defaultdict(float)
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[n] += d; cnt[n] += 1; mx[n] = max(mx[n], d)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
a, e, busy = ev[0][0], ev[0][1], 0
for x, y in ev[1:]:
    if x > e: busy += e - a; a, e = x, y
    else: e = max(e, y)
busy += e - a
print(f"window {win/1e6:.0f} ms (6 steps): some kernel running {busy/1e6:.0f} ms; sum of kernel durations {sum(tot.values()):.0f} ms")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {v/6:8.2f} ms/step  {cnt[n]/6:5.1f} launches/step  longest {mx[n]:7.2f} ms  {n}")
# per stream busy
st = collections.defaultdict(float)
for r in sel: st[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("per stream ms/step:", {k: round(v / 6, 1) for k, v in sorted(st.items(), key=lambda kv: -kv[1])})
PY
