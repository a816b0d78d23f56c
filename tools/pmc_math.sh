#!/bin/bash
# Usage (GPU box): tools/pmc_math.sh <tag> <family: fast-valu|fast-mfma|fast-mfma-f> <mode> -> per-launch PMC means of the fused kernel for one kernel family
TAG=$1; F=$2; MODE=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
for CNT in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tools/pmc_variant.sh $GRAFT_REPO_ROOT/rtl_fm_player_amd/libfmdemod_mi355x.so "$CNT" --mode $MODE --math $F "${@:4}" | tee -a $OUT/pmc_${MODE}_mfma$F.txt
done
