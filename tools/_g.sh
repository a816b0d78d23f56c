cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r14s
for s in $(seq 65 104); do timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/r14s/fuzz_soak.txt
for s in $(seq 1 10); do FUZZ_VOLUMES=1 timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu | tail -1; done > gpurun_out/r14s/fuzz_volumes.txt
awk '{m+=$NF; n+=$4} END {print "fuzz: cases", n, "mismatches", m}' gpurun_out/r14s/fuzz_soak.txt
awk '{m+=$NF; n+=$4} END {print "fuzz with the volume draw: cases", n, "mismatches", m}' gpurun_out/r14s/fuzz_volumes.txt
