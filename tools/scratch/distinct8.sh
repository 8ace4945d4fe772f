#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
( for i in $(seq 1 60); do rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; free -g | sed -n 2p; sleep 5; done ) > $OUT/distinct8_mem.txt 2>&1 &
MON=$!
timeout 900 python3 bench.py --distinct 8 --no-also --no-from-bam --no-cpu-baseline --no-host-leg --steps 16 --warmup 4 > $OUT/distinct8.json 2> $OUT/distinct8.err
echo rc=$?
kill $MON 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/distinct8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['distinct_batches_per_gpu'], d['setup_s'])"
sort -u $OUT/distinct8_mem.txt | tail -8; tail -3 $OUT/distinct8.err
