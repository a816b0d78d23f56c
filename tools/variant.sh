#!/bin/bash
# Usage: tools/variant.sh <name> '<sed script applied to csrc/*.inc>' [extra hipcc flags]
#   -> .ablate/lib_<name>.so : an experimental build of the library for tools/ab.sh (never the shipped one)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; SED=$2; shift 2
T=$(mktemp -d)
mkdir -p $T/pkg $T/include; cp -r $ROOT/rtl_fm_player_amd/csrc $T/pkg/csrc; cp $ROOT/include/*.h $T/include/
mkdir -p $T/tests/c $T/tools; cp $ROOT/tests/c/*.c $T/tests/c/; cp $ROOT/tools/isa_lint.py $T/tools/
rm -f $T/pkg/csrc/*.o
[ -n "$SED" ] && sed -i "$SED" $T/pkg/csrc/*.inc
make -s -C $T/pkg/csrc EXTRA_HIPFLAGS="$*" ISA_LINT=true ../libfmdemod_mi355x.so 2>&1 | grep -E "error|Error" || true
mkdir -p $ROOT/.ablate; cp $T/pkg/libfmdemod_mi355x.so $ROOT/.ablate/lib_$N.so; rm -rf $T
echo "built .ablate/lib_$N.so"
