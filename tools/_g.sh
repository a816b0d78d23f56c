cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 )
SCAN_AMPS=1,4,20,80,128 timeout 1500 python tools/low_amp_volume_scan.py 2>&1 | grep -v amdgpu > gpurun_out/r5t/volume_scan.txt; tail -1 gpurun_out/r5t/volume_scan.txt
grep -v '"max_diff_and_count": {"valu": \[[01], [0-9]*\], "mfma": \[[01], [0-9]*\], "mfma_c": \[[01], [0-9]*\], "mfma_d": \[[01], [0-9]*\]}' gpurun_out/r5t/volume_scan.txt | head
python bench.py --steps 20 --warmup 5 --no-cpu --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['noise_input']['slowdown_vs_timed_input'], d['quiet_input']['slowdown_vs_timed_input'], {m:(v['kernel_ms'],v['frac']) for m,v in d['modes'].items()}, d['roofline']['busy'])"
