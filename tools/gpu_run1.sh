#!/bin/bash
# round 3, GPU call 1: native harness (all cases) + A/B knobs + phase time lines of the small launches + the GPU test suite
mkdir -p gpurun_out; O=gpurun_out
H=tests/native/attn_check
( timeout 900 $H > $O/r3b_native_all.log 2>&1; echo "exit $?" >> $O/r3b_native_all.log ) 
grep -c PASS $O/r3b_native_all.log; grep FAIL $O/r3b_native_all.log | head -20; tail -1 $O/r3b_native_all.log
# A/B: f16 folded vs exact-scale kernel; bias tile vs per-lane loads; resident workgroups per CU
for c in sd15_self_n4096_d40_f16_b2 d40_n4096_hot_f16_b2 d40_logit12_f16; do
  for f in 1 2; do echo "== PWW_ATTN_FOLD=$f $c"; PWW_ATTN_FOLD=$f timeout 120 $H --only $c | grep -E "TIME|FAIL|PASS.*attn max"; done
done > $O/r3b_ab_fold.log 2>&1
for c in sd15_cross_n4096_d40_cols32 sd15_cross_n4096_d40_b16_cols32 sd15_cross_n4096_f16_b16_cols48 sd15_cross_n1024_d80_b16_cols48 sd15_cross_n256_d160_b16_cols32 sd21_cross_n9216_d64_b8_cols32 sd15_cross_n1024_d80_cols32 sd15_cross_n256_d160_cols32 sd15_cross_n64_d160_cols16; do
  for v in "PWW_CROSS_BIAS_LDS=1 PWW_CROSS_WG_PER_CU=4" "PWW_CROSS_BIAS_LDS=1 PWW_CROSS_WG_PER_CU=2" "PWW_CROSS_BIAS_LDS=0 PWW_CROSS_WG_PER_CU=2" "PWW_CROSS_BIAS_LDS=0 PWW_CROSS_WG_PER_CU=4"; do
    echo "== $v $c"; env $v timeout 120 $H --only $c | grep -E "TIME|FAIL"; done
done > $O/r3b_ab_cross.log 2>&1
for c in sd15_cross_n256_d160_cols32 sd15_cross_n64_d160_cols16 sd15_cross_n1024_d80_cols32 sd15_cross_n4096_d40_cols32 sd15_cross_n4096_d40_b16_cols32 sd15_self_n256_d160 sd15_mid_self_n64_d160 sd15_self_n1024_d80; do
  timeout 120 $H --timeline --only $c | grep -E "TIMELINE|TIME "
done > $O/r3b_timeline.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > $O/r3b_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3b_pytest.log
timeout 300 python tools/ab_to_out.py > $O/r3_to_out.md 2>&1
tail -30 $O/r3b_pytest.log
