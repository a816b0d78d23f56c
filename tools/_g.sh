cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r13t
timeout 900 python tools/diag/coburst.py 5 400 1 0 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/r13t/mono_d.txt
timeout 900 python tools/diag/coburst.py 5 400 1 1 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/r13t/mono_d.txt
timeout 900 python tools/diag/determinism.py 5 2000 1 2>&1 | tail -1 | tee -a gpurun_out/r13t/mono_d.txt
