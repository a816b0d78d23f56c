#!/bin/bash
# Usage: tools/isa_func.sh <function-label-substring>  (after tools/isa_report.sh produced /tmp/k.s)
K=$1
START=$(grep -n "^_ZN[A-Za-z0-9_]*${K}[A-Za-z0-9_]*: " /tmp/k.s | head -1 | cut -d: -f1)
tail -n +$START /tmp/k.s | awk '{print} /s_setpc_b64|s_endpgm/ {exit}' > /tmp/f_sel.s
echo "lines: $(wc -l < /tmp/f_sel.s)"
for p in "v_fma_f32\|v_fmac_f32" "v_mul_f32" "v_add_f32\|v_sub_f32" s_waitcnt scratch_ "v_readlane\|v_writelane" ds_read ds_write v_cvt_f32_ubyte global_load v_mov_b32 s_cbranch; do echo "  $p: $(grep -c "$p" /tmp/f_sel.s)"; done
