#!/bin/bash
# GPU box: the steps of round 4's hunt for the packed-fp32 hazard (profiles/archive/r04_pk_opsel_hazard.md), one script:
#   tools/diag/repro/run.sh <step>     (build first: python tools/diag/repro/mkvariants.py)
#   1  product-level co-residency test, the r03m soak as a control, the cut-out victim and its NOP / form variants
#   2  data patterns that tell a stale operand from a dropped modifier; single-site timing edits around the failing instruction
#   3  which instruction forms of the join go wrong, with / without one wait state in front (neighbours bf16 16x16x32, i8 16x16x64)
#   4  the remaining forms (13-18) beside the bf16 neighbour
#   5  the product after the forbidden forms were taken out: soak beside both neighbours, then the -m gpu suite
cd "$(dirname "$0")/../../.." || exit 1
STEP=${1:?step 1..5}
O=gpurun_out/r04_repro$STEP; mkdir -p $O
D=tools/diag/repro/build
hs() { local L=""; for v in "$@"; do L="$L $D/$v.hsaco"; done; echo $L; }
case $STEP in
1)
  ( timeout 900 python -m pytest tests/test_gpu_coresidency.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -25 ) > $O/coresidency_test.txt
  ( timeout 300 python tools/diag/coburst.py 2 20 2 0 2>&1 | tail -3 ) > $O/coburst_control.txt
  ( timeout 900 $D/host $D/neighbour.hsaco 0,3 3 256 128 0 $(hs base nodpp plainrot noasm scalarfma vtaps noload loadtop prioflip w1 vgpr152 plain_all nop7_all nop1_all nop0_all nop3_pk nop3_sdwa nop3_before_pkadd) 2>&1 ) > $O/repro_grid256.txt
  ( timeout 300 $D/host $D/neighbour.hsaco 0,1,2 3 768 64 0 $(hs base vgpr152) 2>&1 ) > $O/repro_grid768.txt
  ( timeout 300 $D/host $D/neighbour.hsaco 0 3 256 128 2 $(hs base nop7_all) 2>&1 ) > $O/repro_prio2.txt
  tail -5 $O/coresidency_test.txt; cat $O/coburst_control.txt; grep -c . $O/repro_grid256.txt ;;
2)
  for pat in 0 1 2 3; do for taps in 0 1; do
    ( PATTERN=$pat TAPS=$taps timeout 120 $D/host $D/neighbour.hsaco 0 3 256 128 0 $D/base.hsaco 2>&1 ) > $O/base_pat${pat}_taps${taps}.txt
  done; done
  ( timeout 600 $D/host $D/neighbour.hsaco 0 4 256 128 0 $(hs b_nop3_before_this b_nop1_before_this b_nop0_before_this b_vnop_before_this b_nop3_after_this b_nop0_after_this b_nop3_before_others b_wait_first base) 2>&1 ) > $O/single_site.txt
  grep -h "neighbour kind\|^==" $O/single_site.txt | cut -c1-220 ;;
3)
  L=""; for f in 0 1 2 3 4 5 6 7 8 9 10 11 12; do L="$L $D/form$f.hsaco $D/form${f}_pad.hsaco"; done
  ( timeout 900 $D/host $D/neighbour.hsaco 0,1 4 256 128 0 $L 2>&1 ) > $O/forms.txt
  grep -h "neighbour kind\|^==" $O/forms.txt | cut -c1-200 ;;
4)
  ( timeout 900 $D/host $D/neighbour.hsaco 0 6 256 128 0 $(hs form0 form13 form14 form15 form16 form17 form18) 2>&1 ) > $O/forms2.txt
  grep -h "neighbour kind\|^==" $O/forms2.txt | cut -c1-200 ;;
5)
  for fam in 2 3; do for kind in 0 1; do
    ( timeout 600 python tools/diag/coburst.py $fam 100 2 $kind 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
  done; done
  ( timeout 300 python tools/diag/coburst.py 3 60 1 0 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
  ( timeout 300 python tools/diag/coburst.py 2 60 1 0 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
  cut -c1-330 $O/coburst_after_fix.txt
  ( timeout 1500 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest_gpu.txt
  cat $O/pytest_gpu.txt ;;
*) echo "step 1..5"; exit 2 ;;
esac
