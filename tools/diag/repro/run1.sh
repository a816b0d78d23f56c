#!/bin/bash
# GPU box, round 4 call 1: the product-level co-residency test, the r03m soak as a control, the cut-out victim and its variants.
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r04a; mkdir -p $O
D=tools/diag/repro/build
( timeout 900 python -m pytest tests/test_gpu_coresidency.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -25 ) > $O/coresidency_test.txt
( timeout 300 python tools/diag/coburst.py 2 20 2 0 2>&1 | tail -3 ) > $O/coburst_control.txt
V="base nodpp plainrot noasm scalarfma vtaps noload loadtop prioflip w1 vgpr152 plain_all nop7_all nop1_all nop0_all nop3_pk nop3_sdwa nop3_before_pkadd"
L=""; for v in $V; do L="$L $D/$v.hsaco"; done
( timeout 900 $D/host $D/neighbour.hsaco 0,3 3 256 128 0 $L 2>&1 ) > $O/repro_grid256.txt
( timeout 300 $D/host $D/neighbour.hsaco 0,1,2 3 768 64 0 $D/base.hsaco $D/vgpr152.hsaco 2>&1 ) > $O/repro_grid768.txt
( timeout 300 $D/host $D/neighbour.hsaco 0 3 256 128 2 $D/base.hsaco $D/nop7_all.hsaco 2>&1 ) > $O/repro_prio2.txt
tail -5 $O/coresidency_test.txt; cat $O/coburst_control.txt; grep -c . $O/repro_grid256.txt
