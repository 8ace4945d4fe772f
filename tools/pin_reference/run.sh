#!/bin/bash
# Pins the in-tree logic (CIGAR/cs walk, markers, consensus windows, BAQ driver, filter, score, decision, relabel list,
# BED side outputs -- SURVEY 8 rows A1-A14, N1) against a REAL secphase build:
#     tools/pin_reference/run.sh /path/to/secphase
# writes the deterministic synthetic fixtures (tools/pin_reference/make_fixtures.py), runs the reference on each with -@1
# (one worker: file order, one rand() stream; src/secphase.c:194-217,713-732) and stores
#     tests/golden/ref_<name>.out.log, ref_<name>.modified.bed, ref_<name>.markers.bed, ref_manifest.json.
# tests/test_reference_pin.py then checks the CPU oracle (and with -m gpu the HIP path) against them.  Together with
# tools/pin_htslib/run.sh (probaln_glocal alone) this turns DESIGN.md's "PARITY UNPINNED" into test results.
# Needs: a secphase binary built from the reference (htslib 1.17, sonLib), python3 + this repo built (make -C synth).
set -eu
BIN=${1:?usage: run.sh /path/to/secphase}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
python3 "$ROOT/tools/pin_reference/make_fixtures.py" "$WORK"
GOLD="$ROOT/tests/golden"
for name in $(python3 -c "import json;print(' '.join(json.load(open('$WORK/manifest.json'))['flags']))"); do
    flags=$(python3 -c "import json;print(' '.join(json.load(open('$WORK/manifest.json'))['flags']['$name']))")
    "$BIN" $flags -@1 -i "$WORK/$name.bam" -f "$WORK/$name.fa" --outDir "$WORK/out_$name" --prefix ref
    cp "$WORK/out_$name/ref.out.log" "$GOLD/ref_$name.out.log"
    cp "$WORK/out_$name/ref.modified_read_blocks.markers.bed" "$GOLD/ref_$name.modified.bed"
    cp "$WORK/out_$name/ref.marker_blocks.bed" "$GOLD/ref_$name.markers.bed"
done
"$BIN" --version > "$WORK/version.txt" 2>&1 || true
python3 - "$WORK" "$GOLD" <<'PY'
import json, sys
work, gold = sys.argv[1], sys.argv[2]
m = json.load(open(work + "/manifest.json"))
m["secphase_version_output"] = open(work + "/version.txt").read()[:400]
json.dump(m, open(gold + "/ref_manifest.json", "w"), indent=1, sort_keys=True)
PY
echo "goldens written to $GOLD/ref_*; now run: python -m pytest tests/test_reference_pin.py"
