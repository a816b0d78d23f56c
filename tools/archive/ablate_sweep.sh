#!/bin/bash
# Run ON THE GPU BOX: per-stage cost of the fused kernel by leaving one stage out at a time
# (.ablate/lib_ab<mask>.so from tools/ablate.sh): kernel ms (unprofiled bench) and VALU / SALU / LDS
# instruction counts per launch (rocprofv3 PMC).   tools/ablate_sweep.sh <outdir> [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
for lib in full 1 2 3 4 8 16; do
  if [ $lib = full ]; then L=$GRAFT_REPO_ROOT/rtl_fm_player_amd/libfmdemod_mi355x.so; else L=$GRAFT_REPO_ROOT/.ablate/lib_ab$lib.so; fi
  [ -f $L ] || continue
  ms=$(FMD_LIB_PATH=$L timeout 120 python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --no-cpu --no-e2e --no-extra --no-check "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  c1=$(timeout 120 $GRAFT_REPO_ROOT/tools/pmc_variant.sh $L "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "$@")
  echo "ablate=$lib kernel_ms=$ms $c1" | tee -a $OUT/ablate.log
done
