#!/bin/bash
# Usage (on the GPU box): tools/pmc_run.sh <outdir-name> "<counters>" -- extra bench args
# Collects PMC counters for the bench workload (counters in their own pass, no tracing flags).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-include-regex fmd_fused --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --preheat 0 --no-cpu --no-e2e --no-extra --no-check "$@" > /dev/null 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(k, r["Counter_Name"])] += 1
for k, d in rows.items():
    if "fmd_" not in k: continue
    print(k, {c: round(v / n[(k, c)], 1) for c, v in d.items()})
PY
