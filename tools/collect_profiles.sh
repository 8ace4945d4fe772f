#!/bin/bash
# Regenerates profiles/rNN_* on an MI355X box: bench lines (default workload with CPU bracket and BAM leg, ONT, mixed),
# the rocprofv3 kernel trace of the default bench command, the PMC passes (tools/pmc_collect.py) and the command line end
# to end.  Usage (from the repo root):  tools/collect_profiles.sh r02 [outdir]     (outdir defaults to gpurun_out/profiles)
# Every rocprofv3 call has the program itself after `--`; --pmc is never combined with a trace option.
set -u
TAG=${1:-r02}
OUT=${2:-gpurun_out/profiles}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --cpu-bracket --from-bam 65536 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python3 bench.py --platform ont > "$OUT/${TAG}_bench_ont.json" 2>> "$OUT/${TAG}_bench.err"
python3 bench.py --platform mixed > "$OUT/${TAG}_bench_mixed.json" 2>> "$OUT/${TAG}_bench.err"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-leg > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2>> "$ROOT/$OUT/${TAG}_bench.err")
cp "$OUT/kt/run_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/kt"
python3 tools/pmc_collect.py --platform hifi --out "$OUT/${TAG}_counters_hifi.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform ont --out "$OUT/${TAG}_counters_ont.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform hifi --full --out "$OUT/${TAG}_counters_prep.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/e2e_cli.py --groups 131072 --threads 64 --batch 32768 > "$OUT/${TAG}_e2e_cli.json" 2>> "$OUT/${TAG}_bench.err"
rm -rf "$ROOT/gpurun_out/pmc_tmp"
ls -la "$OUT"
