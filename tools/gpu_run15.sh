#!/bin/bash
# round 3, GPU call 15: HBM-side traffic of the launch the product issues for the fused cross-attention (16 rows, SD2.1 8 rows, 2 rows)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
for c in sd15_cross_n4096_d40_b16_cols32 sd21_cross_n9216_d64_b8_cols32 sd15_cross_n4096_d40_cols32; do
  bash tools/pmc_traffic.sh $c $O/pmc4_$c --product-only > $O/pmc4_$c.log 2>&1; grep -A9 "cross_fused" $O/pmc4_$c.log | head -12
done
