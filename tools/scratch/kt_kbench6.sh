#!/bin/bash
# kernel trace of tools/scratch/kbench6.py: tools/scratch/kt_kbench6.sh hifi 32768 5
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/ktk6; mkdir -p $OUT
(cd /tmp && KB_TIERS=${KB_TIERS:-1} rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/tools/scratch/kbench6.py" "$@" > "$ROOT/$OUT/kbench.txt" 2> "$ROOT/$OUT/kbench.err")
cat $OUT/kbench.txt
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/kt/run_kernel_trace.csv")))
# the last launch: kernels after the last fast_fwd of class 41 start
t_last = max(int(r["Start_Timestamp"]) for r in rows if "fast_fwd_kernel<2, 21" in r["Kernel_Name"] or "fast_fwd_kernel<1, 41" in r["Kernel_Name"] or "baq_fwd1_kernel<41>" in r["Kernel_Name"])
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t_last - 2000000]
t0 = min(int(r["Start_Timestamp"]) for r in sel)
for r in sorted(sel, key=lambda r: int(r["Start_Timestamp"])):
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.3f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:9.3f} ms  {r["Kernel_Name"][:90]}')
PY
rm -rf $OUT/kt
