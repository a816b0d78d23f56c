cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r5p/pytest.txt; tail -6 gpurun_out/r5p/pytest.txt
for s in 1 2 3 4 5 6; do timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu | tail -1; done | tee gpurun_out/r5p/fuzz.txt
FAMILIES="fast-mfma fast-mfma-d" bash tools/ab_math.sh r5p nfm 2>&1 | grep -v amdgpu | tee gpurun_out/r5p/ab_nfm.txt
