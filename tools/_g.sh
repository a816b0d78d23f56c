cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
( timeout 1200 python -m pytest tests/test_gpu_adversarial.py -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r5l/adv_tests.txt; tail -25 gpurun_out/r5l/adv_tests.txt
( timeout 600 python tools/adversarial_time.py 256 4 2>&1 | grep -v amdgpu | tail -6 ) > gpurun_out/r5l/adv_time.txt; cut -c1-600 gpurun_out/r5l/adv_time.txt
