#!/bin/bash
# Run ON THE GPU BOX: kernel ms of several builds of the library on the same device, interleaved
# (two rounds), for one bench mode.   tools/ab.sh <mode> <lib.so> [<lib.so> ...]
MODE=$1; shift
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for L in "$@"; do
    ms=$(FMD_LIB_PATH=$GRAFT_REPO_ROOT/$L python3 bench.py --steps 100 --no-cpu --no-e2e --no-extra --no-check --mode $MODE 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    echo "round $round mode $MODE $L kernel_ms $ms"
  done
done
