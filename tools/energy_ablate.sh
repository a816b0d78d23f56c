#!/bin/bash
# Run ON THE GPU BOX: what a stage costs in ENERGY.  Builds with stages compiled out (tools/ablate.sh <mask>: 1 A, 2 B, 4 C, 8 D, 16 F -> .ablate/lib_ab<mask>.so)
# and the shipped library, each through bench.py's `sustained` leg (3 s of back-to-back launches) with the package power and shader clock
# sampled from hwmon beside it (bench.py: PowerWatch): energy per launch (joules) = mean power x kernel time.   tools/energy_ablate.sh <tag> [mode] [masks...]
TAG=$1; MODE=${2:-stereo}; shift 2
MASKS=${*:-full 1 2 4 8 16 12 30 31}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
for round in 1 2; do
  for m in $MASKS; do
    L=$GRAFT_REPO_ROOT/.ablate/lib_ab$m.so; [ $m = full ] && L=$GRAFT_REPO_ROOT/rtl_fm_player_amd/libfmdemod_mi355x.so
    FMD_LIB_PATH=$L timeout 200 python3 bench.py --steps 100 --no-cpu --no-e2e --no-check --only-sustained --sustained-seconds 3 --mode $MODE 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['sustained']; p = s.get('power') or {}
w = p.get('package_w_mean'); t = s['kernel_ms']
print('ablate=$m round=$round mode=$MODE kernel_ms', t, 'package_w', w, 'max', p.get('package_w_max'), 'sclk_mhz', p.get('sclk_mhz_mean'), 'J_per_launch', round(w * t * 1e-3, 4) if w else None)"
  done
done | tee $O/energy_ablate_$MODE.txt
