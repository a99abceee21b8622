#!/bin/bash
# round 3, GPU call 3: correctness after the kernel changes, time lines with the shader clock, A/B of the wide stores, GPU tests with
# their printed numbers, the default bench line and its rocprofv3 kernel trace
mkdir -p gpurun_out; O=gpurun_out
H=tests/native/attn_check
( timeout 900 $H > $O/r3c_native_all.log 2>&1; echo "exit $?" >> $O/r3c_native_all.log )
grep -c "^PASS" $O/r3c_native_all.log; grep "^FAIL" $O/r3c_native_all.log | head -10; tail -2 $O/r3c_native_all.log
for c in sd15_cross_n4096_d40_cols32 sd15_cross_n4096_d40_b16_cols32 sd15_cross_n256_d160_cols32 sd15_cross_n1024_d80_cols32 sd15_self_n256_d160 sd15_mid_self_n64_d160 sd15_self_n1024_d80 sd15_self_n4096_d40_bf16_b2; do
  timeout 120 $H --timeline --only $c | grep -E "TIMELINE|TIME "
done > $O/r3c_timeline.log 2>&1
for c in sd15_self_n256_d160 sd15_mid_self_n64_d160 sd15_self_n1024_d80 sd15_self_n4096_d40_bf16_b2 sd15_cross_n256_d160_cols32 sd15_cross_n4096_d40_cols32 sd15_cross_n4096_d40_b16_cols32 sd15_self_n4096_d40_f16_b2; do
  for w in 1 0; do echo "== PWW_ATTN_WIDE_STORE=$w $c"; PWW_ATTN_WIDE_STORE=$w timeout 120 $H --only $c | grep -E "^TIME|FAIL"; done
done > $O/r3c_ab_store.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -rP --timeout 900 --durations=10 > $O/r3c_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3c_pytest.log
tail -4 $O/r3c_pytest.log
timeout 900 python bench.py > $O/r3c_bench_c2.json 2> $O/r3c_bench_c2.log; tail -1 $O/r3c_bench_c2.json | cut -c1-400
bash tools/gpu_profile.sh trace
