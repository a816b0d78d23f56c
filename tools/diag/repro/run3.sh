#!/bin/bash
# GPU box, round 4 call 3: which instruction forms of the join go wrong (same context, only the form changes), with and without one
# wait state in front of them; neighbour kinds 0 (bf16 16x16x32) and 1 (i8 16x16x64).
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r04c; mkdir -p $O
D=tools/diag/repro/build
L=""; for f in 0 1 2 3 4 5 6 7 8 9 10 11 12; do L="$L $D/form$f.hsaco $D/form${f}_pad.hsaco"; done
( timeout 900 $D/host $D/neighbour.hsaco 0,1 4 256 128 0 $L 2>&1 ) > $O/forms.txt
grep -h "neighbour kind\|^==" $O/forms.txt | cut -c1-200
