#!/bin/bash
# round 3, GPU call 12: the reworked multi-block fused cross-attention kernel (LDS-direct double-buffered bias tile, unconditional Q loads, Q ring in pass 1)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 600 $H > $O/r3l_native_all.log 2>&1; echo "exit $?" >> $O/r3l_native_all.log )
grep -c "^PASS" $O/r3l_native_all.log; grep "^FAIL" $O/r3l_native_all.log | head -20; tail -2 $O/r3l_native_all.log
grep -E "^TIME" $O/r3l_native_all.log | grep -E "fused" | grep -E "b16|b8_|b20|ragged|n4096_d40_cols32" | cut -c1-170
for c in sd15_cross_n4096_d40_b16_cols32 sd21_cross_n9216_d64_b8_cols32 sd15_cross_n4096_d40_cols32; do
  timeout 120 $H --timeline --only $c 2>&1 | grep -E "TIMELINE|TIME " | cut -c1-200 >> $O/r3l_timeline.log
done
grep -E "fused cross|stamp|shader" $O/r3l_timeline.log | cut -c1-150
for v in "PWW_CROSS_GATE_WEIGHT=3"; do
  echo "== $v" >> $O/r3l_ab.log
  for c in sd15_cross_n4096_d40_b16_cols32 sd15_cross_n4096_f16_b16_cols48 sd21_cross_n9216_d64_b8_cols32 sd15_cross_n1024_d80_b16_cols48 sd15_cross_n4096_d40_b16_cols16; do
    env $v timeout 120 $H --only $c 2>&1 | grep -E "^FAIL|TIME.*fused" | cut -c1-170 >> $O/r3l_ab.log
  done
done
cat $O/r3l_ab.log
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -m gpu -q --timeout 600 -k "bias_hints or compact or config3_forward or fused or handoff or one_graph" > $O/r3l_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3l_pytest.log
tail -4 $O/r3l_pytest.log
