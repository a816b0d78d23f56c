#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and HBM traffic counters for the
# bench workload.  Counters are collected in their own passes (no tracing flags with --pmc).
#   tools/profile_round.sh r01 [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG            # raw rocprofv3 output stays on the box (tens of MiB); the summary travels
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-e2e --no-extra $*"
PMCBENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --preheat 0 --no-cpu --no-e2e --no-extra --no-check $*"   # counters serialise launches: few steps
echo "== bench (unprofiled)"; timeout 300 $BENCH | tee $OUT/bench_unprofiled.json
echo "== kernel trace + stats"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_traced.json 2>$OUT/trace.log
echo "== pmc FETCH_SIZE"
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCBENCH > /dev/null 2>$OUT/pmc_fetch.log
echo "== pmc WRITE_SIZE"
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMCBENCH > /dev/null 2>$OUT/pmc_write.log
echo "== pmc SQ"
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $PMCBENCH > /dev/null 2>$OUT/pmc_sq.log
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $PMCBENCH > /dev/null 2>$OUT/pmc_sq2.log
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_sq3 -- $PMCBENCH > /dev/null 2>$OUT/pmc_sq3.log
timeout 200 rocprofv3 --kernel-include-regex fmd_fused --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq4 -- $PMCBENCH > /dev/null 2>$OUT/pmc_sq4.log
python3 $ROOT/tools/summarize_profile.py $TAG $OUT $ROOT/gpurun_out/$TAG
cp $OUT/bench_unprofiled.json $OUT/bench_traced.json $ROOT/gpurun_out/$TAG/ 2>/dev/null
du -sh $OUT $ROOT/gpurun_out/$TAG
