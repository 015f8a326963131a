#!/bin/bash
# usage: tools/isa_stats.sh <mangled-kernel-substring>   -- instruction mix of one kernel in build/*.s
S=$(ls build/*gfx950*.s | head -1)
K=$1
awk -v k="$K" '$0 ~ "^_Z[A-Za-z0-9_]*"k"[A-Za-z0-9_]*:" {f=1} f{print} /\.Lfunc_end/{f=0}' "$S" > /tmp/kernel.s
echo "lines: $(wc -l < /tmp/kernel.s)"
for pat in v_fma_f64 v_mul_f64 v_add_f64 v_accvgpr_write v_accvgpr_read scratch_load scratch_store ds_read ds_write v_cndmask _dpp "v_mov_b32 " v_readlane v_writelane s_waitcnt s_load s_barrier v_rcp_f64 v_sqrt v_rndne; do
  printf "%-18s %s\n" "$pat" "$(grep -c -- "$pat" /tmp/kernel.s)"
done
grep -E "^\s+\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size)" "$S" | head -0
