#!/bin/bash
# Usage (GPU box): tools/pmc_variant.sh <lib.so> "<counters>"  -> per-launch counter means for the fused kernel
LIB=$1; CNT="$2"
OUT=/tmp/pmcv_$$
cd /tmp && export TMPDIR=/tmp
FMD_LIB_PATH=$LIB rocprofv3 --kernel-include-regex fmd_fused --pmc $CNT --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --preheat 0 --no-cpu --no-e2e --no-extra --no-check "${@:3}" > /dev/null 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fmd_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / 1e6, 2) for k, v in acc.items()}, "(millions per launch)")
PY
rm -rf $OUT
