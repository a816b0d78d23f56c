#!/bin/bash
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r04d; mkdir -p $O
D=tools/diag/repro/build
L=""; for f in 0 13 14 15 16 17 18; do L="$L $D/form$f.hsaco"; done
( timeout 900 $D/host $D/neighbour.hsaco 0 6 256 128 0 $L 2>&1 ) > $O/forms2.txt
grep -h "neighbour kind\|^==" $O/forms2.txt | cut -c1-200
