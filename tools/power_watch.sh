#!/bin/bash
# Run ON THE GPU BOX: board power and clocks (rocm-smi) while the bench kernel loops.  tools/power_watch.sh <mode> [steps]
MODE=${1:-stereo}; STEPS=${2:-30000}
cd $GRAFT_REPO_ROOT
python3 bench.py --steps $STEPS --no-cpu --no-e2e --no-extra --no-check --mode $MODE > /tmp/pw_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';'; echo
  sleep 1
done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/pw_bench.json')); print('mode', '$MODE', 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
