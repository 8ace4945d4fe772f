#!/bin/bash
# the command line built with ThreadSanitizer (build/tsan/: secphase_tsan + its libspx.so) on a small synthetic BAM.
# Build first, in the CPU container (build/ is git-ignored but travels to the GPU box with the snapshot):
#   tools/sanitize_cpu.sh tsan                      # leaves /tmp/spx_tsan/secphase_amd/{libspx.so,csrc}
#   mkdir -p build/tsan && cp /tmp/spx_tsan/secphase_amd/libspx.so build/tsan/
#   (cd /tmp/spx_tsan/secphase_amd/csrc && hipcc -O1 -g -std=c++17 -fsanitize=thread -o $OLDPWD/build/tsan/secphase_tsan \
#        secphase_main.cpp -L.. -lspx -lpthread -Wl,-rpath,'$ORIGIN')
# then:  gpurun -- 'bash tools/scratch/cli_tsan.sh 8192'
python tools/e2e_cli.py --groups ${1:-8192} --batch 2048 --check-groups 0 --keep > /tmp/e2e_first.json 2>/tmp/e2e_first.err || { tail -5 /tmp/e2e_first.err; exit 1; }
D=$(ls -d /dev/shm/spx_e2e_* | head -1)
rm -rf $D/out $D/out2
TSAN_OPTIONS="halt_on_error=0:log_path=gpurun_out/tsan_cli_log:report_signal_unsafe=0" SPX_TIDY_EXIT=1 build/tsan/secphase_tsan --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out2 --prefix e2e --groupsPerBatch 2048 -@ 8 --devices 0,0 > /dev/null 2> /tmp/cli_tsan.err
echo "rc $?"; tail -2 /tmp/cli_tsan.err | cut -c1-200
secphase_amd/bin/secphase --hifi -i $D/reads.bam -f $D/asm.fa --outDir $D/out --prefix e2e --groupsPerBatch 2048 -@ 8 > /dev/null 2>&1
cmp $D/out/e2e.out.log $D/out2/e2e.out.log && echo "out.log identical to the plain build's"
grep -h SUMMARY gpurun_out/tsan_cli_log* 2>/dev/null | grep -v "libamdhip64\|libhsa-runtime" | sort | uniq -c | sort -rn | head
rm -rf $D
