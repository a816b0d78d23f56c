#!/bin/bash
# Run ON THE GPU BOX: kernel ms of the +-1 LSB kernel families of ONE build on the same device, interleaved, two rounds per mode.
# Families are bench.py's --math names (fmd_config.math; the library reads no environment variable for this):
# FAMILIES="fast-valu fast-mfma fast-mfma-f".   tools/ab_math.sh <tag> [modes...] [-- bench flags]
TAG=$1; shift
MODES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do MODES+=("$1"); shift; done; [ "$1" = "--" ] && shift
[ ${#MODES[@]} -eq 0 ] && MODES=(stereo mono nfm)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in "${MODES[@]}"; do
  for round in 1 2; do
    for f in ${FAMILIES:-fast-valu fast-mfma}; do
      timeout 300 python3 bench.py --steps 100 --no-cpu --no-e2e --no-extra --mode $m --math $f "$@" > $OUT/ab_${m}_mfma${f}_r$round.json 2>> $OUT/ab.err
      python3 - $OUT/ab_${m}_mfma${f}_r$round.json $m $f $round <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print("mode", sys.argv[2], "mfma", sys.argv[3], "round", sys.argv[4], "kernel_ms", d["roofline"]["kernel_ms"], "frac", round(d["roofline"]["frac"],4), "parity", d.get("parity"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    done
  done
done
tail -n 5 $OUT/ab.err
