#!/bin/bash
# round 3, GPU call 6: f16 range-free mode + diagonal-first stage order (correctness + A/B), then the round's measurement set
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 900 $H > $O/r3f_native_all.log 2>&1; echo "exit $?" >> $O/r3f_native_all.log )
grep -c "^PASS" $O/r3f_native_all.log; grep "^FAIL" $O/r3f_native_all.log | head -10; tail -2 $O/r3f_native_all.log
for c in sd15_self_n4096_d40_f16_b2 sd15_self_n4096_d40_bf16_b2 d40_n4096_hot_f16_b2 d64_self_n2304_f16_b8 d64_self_n2304_bf16_b8 sd15_self_n1024_d80 sd15_self_n4096_d40_f16_b16 d40_late_outlier_n4096_f16; do
  for v in 1 0; do echo "== PWW_ATTN_RF=$v $c"; PWW_ATTN_RF=$v timeout 300 $H --only $c | grep -E "^TIME|FAIL|^PASS.*attn max" | cut -c1-200; done
done > $O/r3f_ab_rf.log 2>&1
grep -E "==|TIME" $O/r3f_ab_rf.log | cut -c1-150
timeout 2400 python -m pytest tests -m gpu -q -rP --timeout 900 --durations=5 > $O/r3f_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3f_pytest.log
tail -3 $O/r3f_pytest.log
timeout 900 python bench.py > $O/r3f_bench_c2.json 2> $O/r3f_bench_c2.log; tail -1 $O/r3f_bench_c2.json | cut -c1-200
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --live-traffic > $O/r3f_bench_c2_live.json 2> $O/r3f_bench_c2_live.log; python3 -c "
import json;d=json.loads(open('$O/r3f_bench_c2_live.json').read().strip().splitlines()[-1]);print('live traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))"
bash tools/gpu_profile.sh trace configs
for c in sd15_self_n4096_d40_f16_b16 sd15_self_n4096_d40_bf16_b16 sd21_self_n9216_d64_b8; do
  bash tools/pmc_traffic.sh $c $O/pmc2_$c > $O/pmc2_$c.log 2>&1; tail -9 $O/pmc2_$c.log | head -9
done
