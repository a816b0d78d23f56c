cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
( timeout 1500 python -m pytest tests -m gpu -x -q -k "mfma_d or default_family or quiet or adversarial" 2>&1 | tail -12 ) > gpurun_out/r5n/pytest.txt; tail -6 gpurun_out/r5n/pytest.txt
FAMILIES="fast-mfma fast-mfma-d" bash tools/ab_math.sh r5n mono 2>&1 | grep -v amdgpu | tee gpurun_out/r5n/ab_mono.txt
