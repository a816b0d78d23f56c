cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r13w
for i in 1 2; do bash tools/ab.sh stereo .ablate/lib_pilot8.so rtl_fm_player_amd/libfmdemod_mi355x.so 2>&1; done | tee gpurun_out/r13w/ab_pilot_pairs.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
timeout 900 python tools/diag/lsb_hist.py 64 8 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r13w/lsb_hist_default.txt
FMD_LIB_PATH=$GRAFT_REPO_ROOT/.ablate/lib_pilot8.so timeout 900 python tools/diag/lsb_hist.py 64 8 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r13w/lsb_hist_pilot8.txt
