#!/bin/bash
# AddressSanitizer / ThreadSanitizer runs of the CPU suite (host code of libspx.so: reader, inflate core, staging, host plan,
# gather; GPU sanitizers are not available on the pool).  Builds an instrumented copy of the library under /tmp and loads it
# through SPX_LIB with the sanitizer runtime preloaded.   Usage: tools/sanitize_cpu.sh asan|tsan|ubsan
# (under tsan the two torch.distributed tests report races inside ProcessGroupGloo -- not ours.  The ORACLE can be checked the
#  same way: gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -ffp-contract=off -I include -o oracle/liborc.so oracle/*.c
#  -lm -lpthread, run the suite with LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)",
#  then make -C oracle again.)
set -e
KIND=${1:-asan}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/spx_$KIND
case $KIND in
  tsan) FLAG=thread; RTN=tsan ;;
  ubsan) FLAG="undefined -fno-sanitize=vptr"; RTN=ubsan_standalone ;;
  *) FLAG=address; RTN=asan ;;
esac
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.$RTN-x86_64.so)
rm -rf $W && mkdir -p $W/secphase_amd $W/include
cp -r $ROOT/secphase_amd/csrc $W/secphase_amd/ && cp $ROOT/include/*.h $W/include/ && rm -rf $W/secphase_amd/csrc/obj
make -s -j8 -C $W/secphase_amd/csrc ../libspx.so CXXFLAGS="-O1 -g -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fsanitize=$FLAG -fno-omit-frame-pointer -Wno-inline-asm -Wno-unused-function -Wno-unused-result"
cd $ROOT
rm -f /tmp/spx_${KIND}_log*
SPX_LIB=$W/secphase_amd/libspx.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 TSAN_OPTIONS="halt_on_error=0:log_path=/tmp/spx_${KIND}_log" \
    UBSAN_OPTIONS="print_stacktrace=1:log_path=/tmp/spx_${KIND}_log" \
    python -m pytest tests -q -m "not gpu" || true
grep -h "SUMMARY\|runtime error" /tmp/spx_${KIND}_log* 2>/dev/null | sort | uniq -c
