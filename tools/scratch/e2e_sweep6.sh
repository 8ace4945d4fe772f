#!/bin/bash
# end-to-end command line on one synthetic BAM (tmpfs): which classes take the fast tier on 18 k-group lists, segment size
G=${1:-524288}
OUT=gpurun_out/r06; mkdir -p $OUT
python tools/e2e_cli.py --groups $G --batch 16384 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
run() {  # label, env, args
  for rep in 1 2 3; do
    rm -rf $D/out; S=$(date +%s.%N)
    env SPX_TIMING=1 $2 secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 16384 -@ 16 $3 2> /tmp/e2e.err > /dev/null
    E=$(date +%s.%N)
    echo "$1: wall $(python -c "print(round($E-$S,3))") s; $(grep -o 'time in the scoring loop[^)]*)' /tmp/e2e.err | head -1); md5 $(md5sum < $D/out/e2e.out.log | cut -c1-8)" | tee -a $OUT/e2e_sweep6.txt
  done
}
run "default" "A=1" ""
run "share0" "SPX_FAST_MIN_SHARE=0" ""
run "share1" "SPX_FAST_MIN_SHARE=1" ""
run "seg2048" "SPX_DIN_SEG_MB=2048" ""
run "seg2048 share0" "SPX_DIN_SEG_MB=2048 SPX_FAST_MIN_SHARE=0" ""
run "seg4096 share0" "SPX_DIN_SEG_MB=4096 SPX_FAST_MIN_SHARE=0" ""
run "tiers off" "SPX_DP_TIERS=0" ""
rm -rf $D
