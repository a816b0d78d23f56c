cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
for round in 1 2; do
for lib in full 1 2 4 8 16 12 30 31; do
  if [ $lib = full ]; then L=rtl_fm_player_amd/libfmdemod_mi355x.so; else L=.ablate/lib_ab$lib.so; fi
  FMD_LIB_PATH=$GRAFT_REPO_ROOT/$L python tools/stage_profile.py --math fast-mfma-d --reps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ablate=$lib round=$round kernel_ms', d['kernel_ms_min'], d['kernel_ms_median'], 'clock', d['clock_GHz_est'], 'cycles', int(d['cycles_total_mean']))"
done; done | tee gpurun_out/r5o/ablate_energy.txt
