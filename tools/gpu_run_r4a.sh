#!/bin/bash
# round 4, GPU call 1: the hand-off-free cross-attention route (pww_qproj_stat + pww_cross_attn_fwd_parts): native harness, per-shape route
# timing vs the round-3 route, the new GPU tests (config 4 end to end, pipeline classes vs the reference's classes, hygiene items)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 400 $H --match qproj > $O/r4a_native_qproj.log 2>&1; echo "exit $?" >> $O/r4a_native_qproj.log )
echo "qproj harness: PASS $(grep -c '^PASS' $O/r4a_native_qproj.log) FAIL $(grep -c '^FAIL' $O/r4a_native_qproj.log)"; grep "^FAIL" $O/r4a_native_qproj.log | head -12 | cut -c1-330; tail -1 $O/r4a_native_qproj.log
grep "^TIME" $O/r4a_native_qproj.log | cut -c1-330
( timeout 300 $H --quick > $O/r4a_native_quick.log 2>&1; echo "exit $?" >> $O/r4a_native_quick.log )
echo "quick harness: PASS $(grep -c '^PASS' $O/r4a_native_quick.log) FAIL $(grep -c '^FAIL' $O/r4a_native_quick.log)"; grep "^FAIL" $O/r4a_native_quick.log | head -8 | cut -c1-250; tail -2 $O/r4a_native_quick.log
( timeout 300 $H --match sd15_cross > $O/r4a_native_cross.log 2>&1; echo "exit $?" >> $O/r4a_native_cross.log )
echo "cross harness: PASS $(grep -c '^PASS' $O/r4a_native_cross.log) FAIL $(grep -c '^FAIL' $O/r4a_native_cross.log)"; grep "^FAIL" $O/r4a_native_cross.log | head -8 | cut -c1-250
timeout 400 python tools/time_qproj.py $O/r04_qproj.md > $O/r4a_time_qproj.log 2>&1; echo "time_qproj exit $?"; cat $O/r4a_time_qproj.log | cut -c1-220 | tail -30
timeout 1500 python -m pytest tests/test_qproj_gpu.py tests/test_round4_gpu.py -m gpu -q --timeout 900 -s ${PYTEST_EXTRA} > $O/r4a_pytest.log 2>&1; echo "pytest exit $?" >> $O/r4a_pytest.log
tail -5 $O/r4a_pytest.log; grep -E "^FAILED|^ERROR" $O/r4a_pytest.log | head -20 | cut -c1-300
grep -E "rel-L2|qproj .*parts|default path vs|2 ranks vs|zero fn" $O/r4a_pytest.log | cut -c1-220 | head -60
