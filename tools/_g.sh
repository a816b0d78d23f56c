cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r5g/pytest_gpu.txt
tail -6 gpurun_out/r5g/pytest_gpu.txt
for s in 1 2 3 4; do timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu | tail -1; done | tee gpurun_out/r5g/fuzz.txt
