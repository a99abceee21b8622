#!/bin/bash
# round 4, GPU call 6: cross kernel pass 2 with first / lazy softmax steps and row sums from the ones channel -- harness, timing, tests
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
timeout 600 $H --match cross > $O/r4f_cross.log 2>&1; echo "cross exit $?"; grep -c PASS $O/r4f_cross.log; grep FAIL $O/r4f_cross.log | head -10 | cut -c1-250
timeout 600 $H --match qproj > $O/r4f_qproj.log 2>&1; echo "qproj exit $?"; grep -c PASS $O/r4f_qproj.log; grep FAIL $O/r4f_qproj.log | head -10 | cut -c1-250
grep -i "us\b" $O/r4f_qproj.log | head -30 | cut -c1-220
echo skip timing
timeout 1500 python -m pytest tests/test_round2_gpu.py tests/test_qproj_gpu.py tests/test_attention_gpu.py tests/test_native_gpu.py -m gpu -q --timeout 900 2>&1 | tail -6
