#!/bin/bash
# round 3, GPU call 7: K fragments prefetched across the stage barrier (three stage buffers, PWW_ATTN_FOLD3=1) vs the shipped folded kernel
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
for v in 0 1; do
  echo "== PWW_ATTN_FOLD3=$v correctness (folded-kernel cases)"
  PWW_ATTN_FOLD3=$v timeout 600 $H --match d40_ | grep -E "^FAIL|^PASS.*attn max|NATIVE" | cut -c1-190
  PWW_ATTN_FOLD3=$v timeout 300 $H --match tiny_self_d40 | grep -E "^FAIL|^PASS.*attn max" | cut -c1-190
  PWW_ATTN_FOLD3=$v timeout 300 $H --match d24_self | grep -E "^FAIL|^PASS.*attn max" | cut -c1-190
done > $O/r3g_fold3_check.log 2>&1
grep -c "^PASS" $O/r3g_fold3_check.log; grep "^FAIL" $O/r3g_fold3_check.log | head
for rep in 1 2; do
for c in sd15_self_n4096_d40_bf16_b2 sd15_self_n4096_d40_f16_b2 sd15_self_n4096_d40 sd15_self_n4096_d40_bf16_b16 sd15_self_n4096_d40_f16_b16 d40_n2048_b4_bf16; do
  for v in 0 1; do echo "== PWW_ATTN_FOLD3=$v $c"; PWW_ATTN_FOLD3=$v timeout 300 $H --only $c | grep -E "^TIME|FAIL" | cut -c1-170; done
done
done > $O/r3g_fold3_time.log 2>&1
cat $O/r3g_fold3_time.log
