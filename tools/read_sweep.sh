#!/bin/bash
# reader-only experiments on the GPU box's host (no device): groups/s of spx_bam_next_batch over thread counts and chunk sizes
G=${1:-65536}
python tools/read_bench.py --groups $G --threads 16 --batch 32768 --repeat 1 > /dev/null 2>&1
run() { echo "$1: $(env $2 python tools/read_bench.py --groups $G --threads ${3:-16} --batch 32768 --repeat 2 2>/dev/null | python -c 'import sys,json; d=json.load(sys.stdin); print([r["groups_per_s"] for r in d["runs"]])')"; }
for t in 8 16 32 64; do run "threads $t" "A=1" $t; done
run "chunks of 8 MB" "SPX_BAM_CHUNK_KB=8192"
run "chunks of 64 MB" "SPX_BAM_CHUNK_KB=65536"
run "no CRC check" "SPX_BAM_NOCRC=1"
run "zlib instead of libdeflate" "SPX_NO_LIBDEFLATE=1"
run "no transparent huge pages" "SPX_BAM_NO_THP=1"
echo; SPX_TIMING=1 python tools/read_bench.py --groups $G --threads 16 --batch 32768 --repeat 1 2>&1 | grep timing | tail -5
