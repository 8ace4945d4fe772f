#!/bin/sh
# Pins the oracle against the real htslib:  tools/pin_htslib/run.sh /path/to/htslib-prefix
# (a prefix holding include/htslib/hts.h and lib/libhts.*; htslib 1.17 is what the reference's Dockerfile:17-25 builds).
# Produces tests/golden/htslib_probaln_vectors.json; commit it and run  python -m pytest tests/test_htslib_pin.py
set -e
PREFIX=${1:?usage: run.sh HTSLIB_PREFIX}
HERE=$(cd "$(dirname "$0")" && pwd)
TMP=$(mktemp -d)
cc -O2 -o "$TMP/pin_probaln" "$HERE/pin_probaln.c" -I"$PREFIX/include" -L"$PREFIX/lib" -Wl,-rpath,"$PREFIX/lib" -lhts -lm
python3 "$HERE/make_problems.py" > "$TMP/problems.txt"
"$TMP/pin_probaln" < "$TMP/problems.txt" > "$TMP/answers.txt"
VERSION=$(grep -h 'define HTS_VERSION_TEXT' "$PREFIX"/include/htslib/*.h 2>/dev/null | head -1 | sed 's/.*"\(.*\)".*/\1/')
python3 "$HERE/make_vectors.py" "$TMP/problems.txt" "$TMP/answers.txt" "${VERSION:-unknown}"
rm -rf "$TMP"
