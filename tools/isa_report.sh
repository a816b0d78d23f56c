#!/bin/bash
# Usage: tools/isa_report.sh <kernel-name-substring e.g. ILb0ELi2ELi45>  -> instruction-class counts + resources
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/rtl_fm_player_amd/csrc
F=fast; EXTRA="$EXTRA -fno-slp-vectorize"; case "${1:-ILb0}" in ILb1*) F=exact;; esac; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I$ROOT/include -I$SRC $EXTRA -S --cuda-device-only -o /tmp/k.s $SRC/fmd_kernels_$F.hip 2>/dev/null
K=${1:-ILb0ELi2ELi45}
START=$(grep -n "^_ZN.*fmd_fused_kernel${K}[A-Za-z0-9_]*: " /tmp/k.s | head -1 | cut -d: -f1)
tail -n +$START /tmp/k.s | awk '{print} /s_endpgm/ {exit}' > /tmp/k_sel.s
echo "lines: $(wc -l < /tmp/k_sel.s)"
for p in s_load "v_fma_f32\|v_fmac_f32" "v_mul_f32" "v_add_f32\|v_sub_f32" s_waitcnt scratch_ "v_readlane\|v_writelane" ds_read ds_write v_cvt_f32_ubyte s_barrier global_load global_store v_mov_b32 v_accvgpr; do echo "  $p: $(grep -c "$p" /tmp/k_sel.s)"; done
grep -A40 "\.name:.*fmd_fused_kernel$K" /tmp/k.s | grep -E "vgpr_count|sgpr_count|group_segment_fixed|private_segment_fixed|agpr" 
