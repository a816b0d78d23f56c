cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r14q_stereo 2>&1 | tail -1
bash tools/profile_round.sh r14q_mono --mode mono 2>&1 | tail -1
bash tools/profile_round.sh r14q_nfm --mode nfm 2>&1 | tail -1
mkdir -p gpurun_out/r14
for m in stereo mono nfm; do python tools/stage_profile.py --mode $m 2>/dev/null > gpurun_out/r14/stage_$m.json; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r14/bench_default_line.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --math exact --no-cpu --no-e2e --no-extra 2>/dev/null > gpurun_out/r14/bench_exact_line.json
