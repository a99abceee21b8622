#!/bin/bash
# round 4, GPU call 3: qproj_stat with two operand sets in flight, 160-channel tiles: harness, time lines, route timing
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 400 $H --match qproj > $O/r4c_native_qproj.log 2>&1; echo "exit $?" >> $O/r4c_native_qproj.log )
echo "qproj harness: PASS $(grep -c '^PASS' $O/r4c_native_qproj.log) FAIL $(grep -c '^FAIL' $O/r4c_native_qproj.log)"; grep "^FAIL" $O/r4c_native_qproj.log | head -12 | cut -c1-330; tail -1 $O/r4c_native_qproj.log
grep "^TIME" $O/r4c_native_qproj.log | cut -c1-175
for c in qproj_sd15_n4096_b2 qproj_sd15_n1024_b2 qproj_sd15_n256_b2 qproj_sd15_n4096_b16 qproj_sd21_n9216_b8; do
  timeout 120 $H --timeline --only $c 2>&1 | grep -E "TIMELINE" | cut -c1-200 >> $O/r4c_timeline.log
done
cat $O/r4c_timeline.log | cut -c1-170
timeout 400 python tools/time_qproj.py $O/r04_qproj.md > $O/r4c_time_qproj.log 2>&1; echo "time_qproj exit $?"; cut -c1-220 $O/r4c_time_qproj.log | tail -22
