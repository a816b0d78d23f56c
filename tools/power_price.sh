#!/bin/bash
# Run ON THE GPU BOX: tools/ubench/power_price for each instruction class with rocm-smi sampled beside it -> energy per wave-instruction at the cap
cd $GRAFT_REPO_ROOT
for spec in "mfma 2" "mfma 4" "fma 4" "pkfma 4" "mix 2" "mix 4" "nop 4"; do
  set -- $spec
  tools/ubench/power_price $1 $2 7 > /tmp/pp.txt &
  P=$!
  sleep 3
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; sleep 1; done
  wait $P
  echo; cat /tmp/pp.txt
done
