#!/bin/bash
# end-to-end command line: the host pool's share of the inflate, now that the DP kernels are 1.6 x faster (round 5's sweep: nothing to choose between 15 and 39 %)
G=${1:-524288}
OUT=gpurun_out/r06; mkdir -p $OUT
python tools/e2e_cli.py --groups $G --batch 16384 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
run() {  # label, env
  for rep in 1 2 3; do
    rm -rf $D/out; S=$(date +%s.%N)
    env SPX_TIMING=1 $2 secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 16384 -@ 16 2> /tmp/e2e.err > /dev/null
    E=$(date +%s.%N)
    echo "$1: wall $(python -c "print(round($E-$S,3))") s; $(grep -o 'time in the scoring loop[^)]*)' /tmp/e2e.err | head -1); $(grep -o 'CPU time of the process [0-9.]* core-s' /tmp/e2e.err | head -1); md5 $(md5sum < $D/out/e2e.out.log | cut -c1-8)" | tee -a $OUT/e2e_hostpct6.txt
  done
}
run "default (17 %)" "A=1"
run "host 25 %" "SPX_DIN_HOST_PCT=25"
run "host 33 %" "SPX_DIN_HOST_PCT=33"
run "host 40 %" "SPX_DIN_HOST_PCT=40"
run "host 10 %" "SPX_DIN_HOST_PCT=10"
rm -rf $D
