#!/bin/bash
# ONE command that turns "PARITY UNPINNED" into test results, on any machine with docker and network access (this
# repository's build container has neither; nothing of the reference is copied into this repository):
#
#     tools/pin_all.sh /path/to/a/checkout/of/mobinasri/secphase   [--dry-run]
#
# 1. builds the reference's OWN image from its OWN Dockerfile (htslib 1.17 from the release tarball, sonLib, `make` in
#    programs/: Dockerfile:10-32 of the checkout);
# 2. inside that image, with this repository mounted: tools/pin_htslib/run.sh against the image's htslib
#    (/usr/local: `make install`, Dockerfile:17-25)  ->  tests/golden/htslib_probaln_vectors.json
#    and tools/pin_reference/run.sh against the image's secphase (/home/programs/bin/secphase, Dockerfile:31-32)
#    ->  tests/golden/ref_*.out.log, ref_*.modified.bed, ref_*.markers.bed, ref_manifest.json;
# 3. tells you which tests stop skipping: tests/test_htslib_pin.py (oracle: CPU; kernels: -m gpu) and
#    tests/test_reference_pin.py (oracle: CPU; command line on the HIP path: -m gpu).
# The generator the fixtures need (synth/libspxsynth.so) is built inside the image with its gcc.
set -euo pipefail
usage() { echo "usage: $0 REF_CHECKOUT [--dry-run]" >&2; exit 2; }
[ $# -ge 1 ] || usage
REF=$1
DRY=0
[ "${2:-}" = "--dry-run" ] && DRY=1
[ $# -le 2 ] || usage
[ $# -eq 1 ] || [ $DRY -eq 1 ] || usage
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$REF/Dockerfile" ] || { echo "$REF has no Dockerfile: not a checkout of mobinasri/secphase" >&2; exit 2; }
[ -f "$REF/programs/src/secphase.c" ] || { echo "$REF/programs/src/secphase.c is missing: not a checkout of mobinasri/secphase" >&2; exit 2; }
grep -q 'htslib-1.17' "$REF/Dockerfile" || echo "warning: $REF/Dockerfile does not mention htslib-1.17 (the version SURVEY.md pins)" >&2
IMAGE=secphase-reference-pin
INNER='set -e; cd /spx; make -s -C synth; make -s -C oracle; tools/pin_htslib/run.sh /usr/local; tools/pin_reference/run.sh "$(command -v secphase)"'
CMDS=(
  "docker build -t $IMAGE $REF"
  "docker run --rm -v $ROOT:/spx -w /spx $IMAGE bash -c '$INNER'"
)
if [ $DRY -eq 1 ]; then
    printf '%s\n' "${CMDS[@]}"
    exit 0
fi
command -v docker >/dev/null || { echo "docker is not installed" >&2; exit 3; }
docker build -t "$IMAGE" "$REF"
docker run --rm -v "$ROOT":/spx -w /spx "$IMAGE" bash -c "$INNER"
echo "pinned.  Now:  git add tests/golden/htslib_probaln_vectors.json tests/golden/ref_*  &&  python -m pytest tests/test_htslib_pin.py tests/test_reference_pin.py   (and the same with -m gpu on an MI355X)"
