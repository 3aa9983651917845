#!/bin/bash
# Host-side C++ of the library (beam search object, FLAC / WAV decode, text metrics, ABI checks) under AddressSanitizer on the
# CPU (GPU ASan is not available on this pool).  The instrumented library is built in /tmp and swapped in for the duration
# of the CPU tests that reach that code; the regular library is restored afterwards.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/caiman_asan
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
mkdir -p $OUT
python -c "import sys; sys.path.insert(0, '$ROOT'); from caiman_asr_amd import _lib; _lib.build()"
for f in beam_search audio_decode text_metrics api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-comment -fsanitize=address -fno-gpu-sanitize -shared-libsan \
      -c $ROOT/caiman_asr_amd/csrc/$f.hip -o $OUT/$f.o
done
objs=""
for o in $ROOT/caiman_asr_amd/lib/obj/*.o; do
  b=$(basename $o .hip.o)
  if [ -f $OUT/$b.o ]; then objs="$objs $OUT/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o $OUT/libcaiman_rnnt.so $objs
cp $ROOT/caiman_asr_amd/lib/libcaiman_rnnt.so $OUT/real.so
trap 'cp $OUT/real.so $ROOT/caiman_asr_amd/lib/libcaiman_rnnt.so; touch $ROOT/caiman_asr_amd/lib/libcaiman_rnnt.so' EXIT
cp $OUT/libcaiman_rnnt.so $ROOT/caiman_asr_amd/lib/libcaiman_rnnt.so
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT python -m pytest tests/test_abi.py tests/test_beam_host.py tests/test_evaluate.py \
    tests/test_data_feed.py -x -q -m "not gpu" -p no:cacheprovider
