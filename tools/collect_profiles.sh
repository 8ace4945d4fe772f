#!/bin/bash
# Regenerates profiles/rNN_* on an MI355X box: bench lines (default workload with CPU bracket and BAM leg, ONT, mixed),
# the rocprofv3 kernel trace of the default bench command, the PMC passes (tools/pmc_collect.py) and the command line end
# to end.  Usage (from the repo root):  tools/collect_profiles.sh r02 [outdir]     (outdir defaults to gpurun_out/profiles)
# Every rocprofv3 call has the program itself after `--`; --pmc is never combined with a trace option.
set -u
TAG=${1:-r05}
OUT=${2:-gpurun_out/profiles}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/$OUT"
export TMPDIR=/tmp
cd "$ROOT"
# build BEFORE any profiler run: a process started under rocprofv3 must not spawn compilers (it never builds: --no-build)
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
python3 bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python3 bench.py --platform ont --no-from-bam > "$OUT/${TAG}_bench_ont.json" 2>> "$OUT/${TAG}_bench.err"
python3 bench.py --platform mixed --no-from-bam > "$OUT/${TAG}_bench_mixed.json" 2>> "$OUT/${TAG}_bench.err"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also > "$ROOT/$OUT/${TAG}_bench_under_rocprof.json" 2>> "$ROOT/$OUT/${TAG}_bench.err")
cp "$OUT/kt/run_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
# the dominant kernel's TIMED launches in the trace (the stats file averages set-up and warm-up launches in as well)
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
b = json.loads(open(f"{out}/{tag}_bench_under_rocprof.json").read().strip().splitlines()[-1])
k = b["roofline"]["kernel"]
rows = [r for r in csv.DictReader(open(f"{out}/kt/run_kernel_trace.csv")) if r["Kernel_Name"].startswith("void " + k)]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = b["steps"] * b["roofline"].get("kernel_launches_per_step", 1)  # (a sliced list launches the kernel once per DP slice)
json.dump({"kernel": k, "launches_in_trace": len(d), "all_launches_avg_ms": round(sum(d) / max(len(d), 1), 3),
           "timed_launches_avg_ms": round(sum(d[-n:]) / max(len(d[-n:]), 1), 3), "timed_launches_ms": [round(x, 2) for x in d[-n:]],
           "bench_avg_launch_ms_hip_events": b["roofline"]["avg_launch_ms"], "bench_value": b["value"],
           "note": "rocprofv3 --kernel-trace of the default bench command: durations of the dominant kernel; the last `steps` launches "
                   "are the timed steps, the ones before are set-up and warm-up (other concurrency)"},
          open(f"{out}/{tag}_kernel_trace_dominant.json", "w"), indent=1)
PY
rm -rf "$OUT/kt"
if [ -z "${SKIP_PMC:-}" ]; then
python3 tools/pmc_collect.py --platform hifi --out "$OUT/${TAG}_counters_hifi.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform ont --out "$OUT/${TAG}_counters_ont.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform mixed --out "$OUT/${TAG}_counters_mixed.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform hifi --full --out "$OUT/${TAG}_counters_prep.json" >> "$OUT/${TAG}_bench.err" 2>&1
python3 tools/pmc_collect.py --platform mixed --full --out "$OUT/${TAG}_counters_prep_mixed.json" >> "$OUT/${TAG}_bench.err" 2>&1
fi
# the command line end to end (device-resident input: the default), its kernel trace, and the round-3 host reader for comparison
python3 tools/e2e_cli.py --groups 524288 --threads 64 --rocprof "$OUT/cli_kt" > "$OUT/${TAG}_e2e_cli.json" 2>> "$OUT/${TAG}_bench.err"
cp "$OUT/cli_kt"/*/cli_kernel_stats.csv "$OUT/${TAG}_e2e_cli_kernel_stats.csv" 2>/dev/null || cp "$OUT/cli_kt"/cli_kernel_stats.csv "$OUT/${TAG}_e2e_cli_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/cli_kt"
python3 tools/e2e_cli.py --groups 262144 --threads 64 --host-input --batch 16384 > "$OUT/${TAG}_e2e_cli_host_input.json" 2>> "$OUT/${TAG}_bench.err"
# the BGZF inflate kernels alone (decode + copy + CRC: the default; the decode kernel by itself; round 4's first kernel; round 3's): GB/s, and the
# default's counters
{
  echo '{"default_decode_copy_crc":'; python3 tools/inflate_bench.py --groups 49152 2>> "$OUT/${TAG}_bench.err"
  echo ',"decode_kernel_only":'; SPX_INFLATE_TOK_STAGE=1 python3 tools/inflate_bench.py --groups 49152 2>> "$OUT/${TAG}_bench.err"
  echo ',"sixteen_lanes_per_block":'; SPX_INFLATE_TOK=16 python3 tools/inflate_bench.py --groups 49152 2>> "$OUT/${TAG}_bench.err"
  echo ',"round4_first_kernel_two_blocks_per_wave":'; SPX_INFLATE_TOK=0 python3 tools/inflate_bench.py --groups 49152 2>> "$OUT/${TAG}_bench.err"
  echo ',"round3_kernel_one_block_per_wave":'; SPX_INFLATE_TOK=0 SPX_INFLATE_LANES=64 python3 tools/inflate_bench.py --groups 49152 2>> "$OUT/${TAG}_bench.err"
  echo ',"default_on_ont_blocks":'; python3 tools/inflate_bench.py --groups 8192 --platform ont 2>> "$OUT/${TAG}_bench.err"
  echo '}'
} > "$OUT/${TAG}_inflate_bench.json"
if [ -z "${SKIP_PMC:-}" ]; then
python3 tools/pmc_collect.py --inflate --more --platform hifi --groups-per-step 32768 --steps 2 --out "$OUT/${TAG}_counters_inflate.json" >> "$OUT/${TAG}_bench.err" 2>&1
fi
rm -rf "$ROOT/gpurun_out/pmc_tmp"
ls -la "$OUT"
