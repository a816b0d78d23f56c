cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r5h/bench_default.json 2> gpurun_out/r5h/bench_default.err ) 2>&1 | tail -3
tail -3 gpurun_out/r5h/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5h/bench_default.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "dtype", d["dtype"])
for k in ("sustained","noise_input","quiet_input"):
    print(k, {x:d[k][x] for x in d[k] if x in ("kernel_ms","slowdown_vs_timed_input","parity","frac")})
print("modes", d.get("modes"))
print("busy", d["roofline"].get("busy"))
print("single_stream", d.get("single_stream"))
PY
