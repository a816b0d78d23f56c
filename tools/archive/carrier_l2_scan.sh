#!/bin/bash
# GPU box, tuning build (.ablate/lib_tuning.so, EXTRA_CFLAGS=-DFMD_TUNING): which fraction L of the redo threshold K needs the full
# recomputation from the IQ words?  For each L: fuzz_parity.py 400 x seeds, the noise / hand-over tests, the noise bench.
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r04w}; mkdir -p $O
export FMD_LIB_PATH=$GRAFT_REPO_ROOT/.ablate/lib_tuning.so
for L in ${LS:-0.001 0.0625 0.125 0.25 0.5 1.0}; do
  export FMD_CARRIER_L2=$L
  bad=0
  for s in ${SEEDS:-1 2 3 4 5 6 7 8}; do
    out=$(timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu); n=$(echo "$out" | tail -1 | awk '{print $NF}'); bad=$((bad + n))
    echo "$out" | grep MISMATCH | head -3 | sed "s/^/   L=$L /"
  done
  t=$(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q --no-header -p no:cacheprovider -k "noise or hand_over or fuzz or named" 2>&1 | tail -1)
  ms=$(python bench.py --no-cpu --no-e2e --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['noise_input']['ms_per_step'], d['noise_input']['slowdown_vs_timed_input'])")
  echo "L2 $L | fuzz mismatches $bad | pytest: $t | bench fm / noise ms, ratio: $ms"
done | tee $O/carrier_l2.txt
