#!/bin/bash
# one synthetic BAM, the command line under SPX_TIMING in several configurations: wall times (incl. the parent's view) + core-seconds
G=${1:-262144}
python tools/e2e_cli.py --groups $G --batch 16384 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
run() {
  rm -rf $D/out
  S=$(date +%s.%N)
  env SPX_TIMING=1 $2 secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 16384 $3 2> /tmp/e2e.err > /dev/null
  E=$(date +%s.%N)
  echo "== $1: parent sees $(python3 -c "print(round($E-$S,3))") s"; grep -h "scoring loop\|whole process\|inflate chunks" /tmp/e2e.err | sed 's/^\[[0-9: -]*\] //' | cut -c1-200
}
while read -r line; do
  [ -z "$line" ] && continue
  run "$line" "${line%%|*}" "${line#*|}"
done <<CFG
A=1|-@ 16 --gpuInflate 8
A=1|-@ 16 --gpuInflate 12
A=1|-@ 16 --gpuInflate 16
SPX_BAM_HOST_WINDOW=2|-@ 16 --gpuInflate 12
A=1|-@ 16 --gpuInflate 8
A=1|-@ 16 --gpuInflate 12
A=1|-@ 16 --gpuInflate 16
SPX_BAM_HOST_WINDOW=2|-@ 16 --gpuInflate 12
CFG
rm -rf $D
