#!/bin/bash
# kernel trace of tools/scratch/kbench6.py, every kernel of the LAST replay by stream:  tools/scratch/kt_kbench_all6.sh ont 16384 3
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/ktk6; mkdir -p $OUT
(cd /tmp && KB_TIERS=${KB_TIERS:-1} rocprofv3 --kernel-trace -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/tools/scratch/kbench6.py" "$@" > "$ROOT/$OUT/kbench.txt" 2> "$ROOT/$OUT/kbench.err")
cat $OUT/kbench.txt
python3 - "$OUT" <<'PY'
import csv, sys, glob, re
out = sys.argv[1]
f = (glob.glob(f"{out}/kt/run_kernel_trace.csv") + glob.glob(f"{out}/kt/*/run_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ms = float(re.search(r"wall/launch ([0-9.]+) ms", open(f"{out}/kbench.txt").read()).group(1))
t_end = max(int(r["End_Timestamp"]) for r in rows)
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - ms * 1e6 * 0.5]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    x, y = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if y - x > 200000: print(f'{(x-t0)/1e6:8.2f} -> {(y-t0)/1e6:8.2f} ({(y-x)/1e6:6.2f}) stream {r["Stream_Id"]:>3} {r["Kernel_Name"][:70]}')
PY
rm -rf $OUT/kt
