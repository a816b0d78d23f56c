#!/bin/bash
# Run ON THE GPU BOX: kernel ms of several builds of the library (tools/variant.sh) for 1 / 4 / 16 blocks per launch, stereo and mono, interleaved, two rounds.
#   tools/ab_blocks.sh <tag> <lib.so> [<lib.so> ...]      (paths relative to the repository root; NOCHECK=1 for timing builds whose results are wrong)
TAG=$1; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
for r in 1 2; do for m in ${MODES:-stereo mono}; do for b in ${BLOCKS:-1 4 16}; do for L in "$@"; do
FMD_LIB_PATH=$GRAFT_REPO_ROOT/$L python3 bench.py --blocks $b --mode $m --steps 300 --no-cpu --no-e2e --no-extra ${NOCHECK:+--no-check} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('round $r mode $m blocks $b lib $L kernel_ms', d['roofline']['kernel_ms'], 'parity', (d.get('parity') or {}).get('max_abs_lsb'))"
done; done; done; done | tee gpurun_out/$TAG/ab_blocks.txt
