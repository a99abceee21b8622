#!/bin/bash
# round 3, GPU call 17: running extremes in pass 1 -- harness, time line, full GPU suite, smoke
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 400 $H > $O/r3p_native_all.log 2>&1; echo "exit $?" >> $O/r3p_native_all.log )
grep -c "^PASS" $O/r3p_native_all.log; grep "^FAIL" $O/r3p_native_all.log | head -10; tail -2 $O/r3p_native_all.log
grep -E "^TIME" $O/r3p_native_all.log | grep -E "fused_ex" | grep -v compact | grep -E "b16|b8_|b20|n4096_d40_cols32|n256_d160_cols32" | cut -c1-150
timeout 100 $H --timeline --only sd15_cross_n4096_d40_b16_cols32 2>&1 | grep -E "TIMELINE" | head -9 | cut -c1-140 | tee $O/r3p_timeline.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x > $O/r3p_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3p_pytest.log
tail -3 $O/r3p_pytest.log; grep "^FAILED" $O/r3p_pytest.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3p_smoke.log 2>&1; tail -1 $O/r3p_smoke.log
