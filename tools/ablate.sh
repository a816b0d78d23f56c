#!/bin/bash
# Usage: tools/ablate.sh <mask> [waves per SIMD]  -> .ablate/lib_ab<mask>.so with the stages in <mask> compiled out
# (FMD_ABLATE in csrc/fmd_kernels.inc: 1 A, 2 B, 4 C, 8 D, 16 F, +32: the stages behind a compiled-out B or C keep live operands).  Timing builds only: results are garbage.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
M=$1; W=${2:-3}
T=$(mktemp -d)
cp -r $ROOT/rtl_fm_player_amd/csrc $T/csrc; mkdir -p $T/include; cp $ROOT/include/*.h $T/include/
mkdir -p $T/x; mv $T/csrc $T/x/csrc; mkdir -p $T/x/../include
make -s -C $T/x/csrc clean >/dev/null 2>&1 || true
rm -f $T/x/csrc/*.o; make -s -C $T/x/csrc EXTRA_HIPFLAGS="-DFMD_ABLATE=$M -DFMD_FAST_WAVES=$W $FMD_EXTRA" INC="-I$ROOT/include -I. -I/opt/rocm/include" ISA_LINT=true $FMD_MAKEVARS ../libfmdemod_mi355x.so 2>&1 | grep -E "error" || true
mkdir -p $ROOT/.ablate; cp $T/x/libfmdemod_mi355x.so $ROOT/.ablate/lib_ab$M.so; rm -rf $T
echo "built .ablate/lib_ab$M.so"
