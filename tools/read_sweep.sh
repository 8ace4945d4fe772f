#!/bin/bash
# reader-only experiments on the GPU box's host
G=${1:-65536}
python tools/read_bench.py --groups $G --threads 64 --batch 32768 --repeat 1 > /dev/null 2>&1
run() { echo "$1: $(env $2 python tools/read_bench.py --groups $G --threads ${3:-64} --batch 32768 --repeat 2 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print([r["groups_per_s"] for r in d["runs"]])')"; }
run "base 64" "A=1"
run "base 32" "A=1" 32
run "pread 64" "SPX_BAM_PREAD=1"
run "populate-all 64" "SPX_BAM_POPULATE_ALL=1"
run "pretouch 12GB 64" "SPX_BAM_PRETOUCH_GB=12"
run "pretouch 12GB no-thp 64" "SPX_BAM_PRETOUCH_GB=12 SPX_BAM_NO_THP=1"
run "populate-all + pretouch 64" "SPX_BAM_POPULATE_ALL=1 SPX_BAM_PRETOUCH_GB=12"
run "populate-all + pretouch 128" "SPX_BAM_POPULATE_ALL=1 SPX_BAM_PRETOUCH_GB=12" 128
run "pread + pretouch 64" "SPX_BAM_PREAD=1 SPX_BAM_PRETOUCH_GB=12"
echo; SPX_TIMING=1 python tools/read_bench.py --groups $G --threads 64 --batch 32768 --repeat 1 2>&1 | grep timing
echo; SPX_BAM_POPULATE_ALL=1 SPX_BAM_PRETOUCH_GB=12 SPX_TIMING=1 python tools/read_bench.py --groups $G --threads 64 --batch 32768 --repeat 1 2>&1 | grep timing
