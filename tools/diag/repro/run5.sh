#!/bin/bash
# GPU box: the product after the forbidden packed forms were taken out - soak beside the bf16 / i8 neighbours, parity tests.
cd "$(dirname "$0")/../../.." || exit 1
O=gpurun_out/r04e; mkdir -p $O
for fam in 2 3; do for kind in 0 1; do
  ( timeout 600 python tools/diag/coburst.py $fam 100 2 $kind 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
done; done
( timeout 300 python tools/diag/coburst.py 3 60 1 0 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
( timeout 300 python tools/diag/coburst.py 2 60 1 0 2>&1 | tail -1 ) >> $O/coburst_after_fix.txt
cat $O/coburst_after_fix.txt | cut -c1-330
( timeout 1500 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
