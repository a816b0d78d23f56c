#!/bin/bash
# The host layer (fmd_host.c, fmd_wav.c: ingest ring, pump bookkeeping, WAV writer, drop-in registry) under AddressSanitizer +
# UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on this pool): a scratch build of the library with the
# host objects instrumented, and the CPU tests that exercise them run against it.     tools/asan_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pkg $T/include $T/tests/c
cp -r $ROOT/rtl_fm_player_amd/csrc $T/pkg/csrc; cp $ROOT/include/*.h $T/include/; cp $ROOT/tests/c/*.c $T/tests/c/
rm -f $T/pkg/csrc/fmd_host.o $T/pkg/csrc/fmd_wav.o          # the kernel objects are reused as they are
make -s -C $T/pkg/csrc ../libfmdemod_mi355x.so EXTRA_CFLAGS="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
A=$(gcc -print-file-name=libasan.so); U=$(gcc -print-file-name=libubsan.so)
cd $ROOT
FMD_LIB_PATH=$T/pkg/libfmdemod_mi355x.so LD_PRELOAD=$A:$U ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 \
  python -m pytest tests/test_ring_ref.py tests/test_wav_cpu.py tests/test_capi_cpu.py -x -q
rm -rf $T
