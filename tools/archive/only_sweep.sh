#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02z
for mode in mono stereo; do
for lib in 31 30 29 23 15; do
  L=$GRAFT_REPO_ROOT/.ablate/lib_ab$lib.so
  ms=$(FMD_LIB_PATH=$L timeout 120 python3 bench.py --steps 60 --no-cpu --no-e2e --no-extra --no-check --mode $mode 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "mode=$mode only_stage_mask=$((31-lib)) kernel_ms=$ms" | tee -a gpurun_out/r02z/only.log
done; done
