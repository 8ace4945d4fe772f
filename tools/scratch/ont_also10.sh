#!/bin/bash
mkdir -p gpurun_out/r05
SPX_TIMING=1 SPX_BENCH_KEEP_GUARD_BLOCKS=1 python3 bench.py --platform ont --steps 8 --warmup 2 --no-also --no-host-leg --no-build --verify 64 --no-cpu-baseline --no-host-input-leg --no-from-bam --distinct 8 --depth 4 2>gpurun_out/r05/ont10.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont keep blocks:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
grep -c . gpurun_out/r05/ont10.err; grep "hipMalloc\|pool of" gpurun_out/r05/ont10.err | tail -60
