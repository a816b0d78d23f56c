#!/bin/bash
# Run ON THE GPU BOX: the round's standard check after a kernel change.
#   tools/gpu_check.sh <tag> [quick]     -> gpurun_out/<tag>/{pytest.log,fuzz*.log,bench_*.json}
TAG=$1; MODE=${2:-full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -n 4 $OUT/pytest.log
if [ $MODE = full ]; then
  for s in 1 2 3 4; do timeout 200 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v refused > $OUT/fuzz$s.log; done
  grep -h "MISMATCH\|^seed" $OUT/fuzz*.log
fi
for m in stereo mono nfm; do
  timeout 300 python bench.py --steps 100 --no-cpu --no-e2e --no-extra --mode $m > $OUT/bench_$m.json 2>> $OUT/bench.err
  python - $OUT/bench_$m.json $m <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], "ms", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
tail -n 3 $OUT/bench.err
