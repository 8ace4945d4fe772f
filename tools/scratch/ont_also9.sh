#!/bin/bash
SPX_BENCH_KEEP_GUARD_BLOCKS=1 python3 bench.py --platform ont --steps 8 --warmup 2 --no-also --no-host-leg --no-build --verify 64 --no-cpu-baseline --no-host-input-leg --no-from-bam --distinct 8 --depth 4 2>/tmp/ont9.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont, blocks of the guard-exposure lists left waiting, library decides:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
timeout 600 python3 -m pytest tests -x -q -m gpu -k "pipeline or batch_ or memory or trim or slice or command_line" 2>&1 | tail -3
