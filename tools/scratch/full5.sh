#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
T0=$(date +%s.%N)
python3 bench.py --steps 20 --warmup 5 > $OUT/full_$1.json 2> $OUT/full_$1.err
echo "rc=$? wall_s=$(echo "$(date +%s.%N) - $T0" | bc)"
python3 - $OUT/full_$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["distinct_batches_per_gpu"], d["setup_s"], d["kernel_ms_per_step"])
print(d.get("metric_8d"))
print({k:(v.get("value"), v.get("wall_s"), (v.get("from_bam") or {}).get("loop_groups_per_s")) for k,v in d["also"].items()})
r=d["roofline"]; print(r["frac"], r.get("alone_frac"), r.get("lane_instr_per_cell"), r["phase"]["frac"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["threads"], d["cpu_baseline"]["effective_cpus"])
print(d.get("pipelined_from_host"))
PY
tail -3 $OUT/full_$1.err
