#!/bin/bash
# Run ON THE GPU BOX: tools/ubench/power_price for each instruction class with rocm-smi sampled beside it -> energy per wave-instruction at the cap
cd $GRAFT_REPO_ROOT
# SPECS="lds128 4,nop 4" picks classes (comma-separated "<kind> <waves per SIMD>")
LIST=("mfma 2" "mfma 4" "fma 4" "pkfma 4" "mix 2" "mix 4" "nop 4" "lds128 2" "lds128 4" "lds64 4" "lds32 4" "ldsw128 4" "perm 4" "cvt 4" "dpp 4")
[ -n "$SPECS" ] && IFS=',' read -ra LIST <<< "$SPECS"
for spec in "${LIST[@]}"; do
  set -- $spec
  tools/ubench/power_price $1 $2 7 > /tmp/pp.txt &
  P=$!
  sleep 3
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; sleep 1; done
  wait $P
  echo; cat /tmp/pp.txt
done
