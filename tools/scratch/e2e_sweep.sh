#!/bin/bash
# end-to-end command line on one synthetic BAM (tmpfs), host / device inflate split and thread counts swept
G=${1:-262144}
python tools/e2e_cli.py --groups $G --batch 16384 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
run() {  # label, env, args
  for rep in 1 2; do
    rm -rf $D/out; S=$(date +%s.%N)
    env SPX_TIMING=1 $2 secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 16384 $3 2> /tmp/e2e.err > /dev/null
    E=$(date +%s.%N)
    echo "$1: wall $(python -c "print(round($E-$S,3))") s; $(grep -o 'time in the scoring loop[^;]*' /tmp/e2e.err | head -1); $(grep -o 'inflate chunks[^;]*' /tmp/e2e.err | head -1)"
  done
}
run "default (-@64, gpuInflate 5)" "A=1" "-@ 64"
run "-@64 gpuInflate 0" "A=1" "-@ 64 --gpuInflate 0"
run "-@64 gpuInflate 8" "A=1" "-@ 64 --gpuInflate 8"
run "-@64 gpuInflate 12" "A=1" "-@ 64 --gpuInflate 12"
run "-@32 gpuInflate 8" "A=1" "-@ 32 --gpuInflate 8"
run "-@16 gpuInflate 8" "A=1" "-@ 16 --gpuInflate 8"
run "-@16 gpuInflate 8 device-all" "SPX_BAM_DEVICE_ALL=1" "-@ 16 --gpuInflate 8"
run "-@16 gpuInflate 16 device-all" "SPX_BAM_DEVICE_ALL=1" "-@ 16 --gpuInflate 16"
grep -h "spx timing\|secphase\]" /tmp/e2e.err | tail -12
rm -rf $D
