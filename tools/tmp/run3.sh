cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r04f; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/r04_bench.json 2> $OUT/r04_bench.err
python3 tools/e2e_cli.py --groups 524288 --threads 64 --rocprof $OUT/cli_kt > $OUT/r04_e2e_cli.json 2>> $OUT/r04_bench.err
cp $OUT/cli_kt/*/cli_kernel_stats.csv $OUT/r04_e2e_cli_kernel_stats.csv 2>/dev/null || cp $OUT/cli_kt/cli_kernel_stats.csv $OUT/r04_e2e_cli_kernel_stats.csv 2>/dev/null
rm -rf $OUT/cli_kt
timeout 400 python3 tools/fuzz_reader.py 41 150 gpu > $OUT/fuzz_reader_gpu.txt 2>&1
timeout 300 python3 tools/fuzz.py 42 120 gpu > $OUT/fuzz_gpu.txt 2>&1
timeout 200 python3 tools/fuzz_inflate.py 14 90 > $OUT/fuzz_inflate.txt 2>&1
tail -2 $OUT/fuzz_reader_gpu.txt $OUT/fuzz_gpu.txt $OUT/fuzz_inflate.txt
