#!/bin/bash
python3 bench.py --platform ont --steps 8 --warmup 2 --no-also --no-host-leg --no-build --verify 64 --no-cpu-baseline --no-host-input-leg --no-from-bam --distinct 8 --depth 4 2>/tmp/ont8.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont also-leg flags (guard exposure on, trim after it):', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d.get('guard_exposure',{}).get('hip_path'))"
grep -c "hipMalloc" /tmp/ont8.err; grep "hipMalloc" /tmp/ont8.err | tail -5
