#!/bin/bash
# GPU box, round 4 call 2: what exactly goes wrong in y2[2] of the base build - data patterns that tell a stale operand from a dropped
# modifier, and single-site timing edits around the failing instruction.
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r04b; mkdir -p $O
D=tools/diag/repro/build
for pat in 0 1 2 3; do for taps in 0 1; do
  ( PATTERN=$pat TAPS=$taps timeout 120 $D/host $D/neighbour.hsaco 0 3 256 128 0 $D/base.hsaco 2>&1 ) > $O/base_pat${pat}_taps${taps}.txt
done; done
L=""; for v in b_nop3_before_this b_nop1_before_this b_nop0_before_this b_vnop_before_this b_nop3_after_this b_nop0_after_this b_nop3_before_others b_wait_first base; do L="$L $D/$v.hsaco"; done
( timeout 600 $D/host $D/neighbour.hsaco 0 4 256 128 0 $L 2>&1 ) > $O/single_site.txt
grep -h "neighbour kind\|^==" $O/single_site.txt | cut -c1-220
