cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
( timeout 1500 python -m pytest tests -m gpu -x -q -k "mfma_d" 2>&1 | tail -25 ) > gpurun_out/r5b/pytest_mfma_d.txt
tail -25 gpurun_out/r5b/pytest_mfma_d.txt
FAMILIES="fast-mfma-c fast-mfma-d" bash tools/ab_math.sh r5b stereo 2>&1 | tee gpurun_out/r5b/ab.txt
