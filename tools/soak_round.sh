cd $GRAFT_REPO_ROOT; O=gpurun_out/${SOAK_TAG:-r06s}; mkdir -p $O
for s in $(seq ${SOAK_SEED0:-21} ${SOAK_SEED1:-60}); do timeout 600 python tools/fuzz_parity.py 400 $s 2>&1 | grep -v amdgpu | tail -1; done > $O/fuzz_soak.txt
FUZZ_VOLUMES=1 timeout 600 python tools/fuzz_parity.py 400 1 2>&1 | grep -v amdgpu | tail -1 >> $O/fuzz_soak.txt
for s in $(seq 1 ${SOAK_F_SEEDS:-10}); do FUZZ_F=1 timeout 900 python tools/fuzz_parity.py 200 $s 2>&1 | grep -v amdgpu | tail -1; done > $O/fuzz_f_soak.txt
for fam in ${SOAK_FAMS:-2 3 7}; do for kind in 0 1; do timeout 900 python tools/diag/coburst.py $fam 400 2 $kind 2>&1 | tail -1; done; done > $O/coburst_soak.txt
for fam in ${SOAK_FAMS_MONO:-2 3 7}; do timeout 900 python tools/diag/coburst.py $fam 400 1 0 2>&1 | tail -1; done >> $O/coburst_soak.txt
timeout 900 python tools/diag/determinism.py 7 2000 2 2>&1 | tail -2 > $O/determinism.txt
timeout 900 python tools/diag/determinism.py 3 1000 2 2>&1 | tail -2 >> $O/determinism.txt
timeout 900 python tools/diag/determinism.py 3 1000 1 2>&1 | tail -2 >> $O/determinism.txt
timeout 900 python tools/diag/determinism.py 7 2000 1 2>&1 | tail -2 >> $O/determinism.txt
awk '{m+=$NF; n+=$4} END {print "fuzz: cases", n, "mismatches", m}' $O/fuzz_soak.txt; awk '{m+=$NF; n+=$4} END {print "fuzz (the configurations of the default family): cases", n, "mismatches", m}' $O/fuzz_f_soak.txt; cut -c1-250 $O/coburst_soak.txt; cat $O/determinism.txt
