cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
( timeout 600 python tools/quiet_time.py 256 4 2>&1 | grep -v amdgpu | tail -12 ) > gpurun_out/r5j/quiet_time.txt; cut -c1-330 gpurun_out/r5j/quiet_time.txt
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r5j/pytest_gpu.txt; tail -4 gpurun_out/r5j/pytest_gpu.txt
