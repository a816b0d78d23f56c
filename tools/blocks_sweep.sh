#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1 2 4 8 16; do for m in stereo mono; do
  python3 bench.py --blocks $b --mode $m --steps 200 --no-cpu --no-e2e --no-extra > /tmp/bs.json 2>/tmp/bs.err
  python3 - $b $m <<'PY'
import sys,json
try:
    d=json.load(open('/tmp/bs.json')); print("blocks", sys.argv[1], "mode", sys.argv[2], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "Msps", d["value"])
except Exception as e:
    print("blocks", sys.argv[1], sys.argv[2], "FAILED", e, open('/tmp/bs.err').read()[-300:])
PY
done; done
