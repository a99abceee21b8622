#!/bin/bash
# round 3, GPU call 8: validation of the build (natural key order, f16 range-free with the self-logit floor) + the round's measurement set
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 900 $H > $O/r3h_native_all.log 2>&1; echo "exit $?" >> $O/r3h_native_all.log )
grep -c "^PASS" $O/r3h_native_all.log; grep "^FAIL" $O/r3h_native_all.log | head -10; tail -2 $O/r3h_native_all.log
grep -E "^TIME" $O/r3h_native_all.log | grep -E "fused|self_n4096|hot|n2304|n9216" | cut -c1-170
timeout 2400 python -m pytest tests -m gpu -q -rP --timeout 900 --durations=5 > $O/r3h_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3h_pytest.log
tail -3 $O/r3h_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3h_smoke.log 2>&1; tail -2 $O/r3h_smoke.log
timeout 900 python bench.py > $O/r3h_bench_c2.json 2> $O/r3h_bench_c2.log; tail -1 $O/r3h_bench_c2.json | cut -c1-200
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --live-traffic > $O/r3h_bench_c2_live.json 2> $O/r3h_bench_c2_live.log; python3 -c "
import json;d=json.loads(open('$O/r3h_bench_c2_live.json').read().strip().splitlines()[-1]);print('live traffic', d['roofline'].get('traffic'), d['value'])"
bash tools/gpu_profile.sh trace configs
for c in sd15_self_n4096_d40_bf16_b2 sd15_self_n4096_d40_f16_b16 sd15_self_n4096_d40_bf16_b16 sd21_self_n9216_d64_b8; do
  bash tools/pmc_traffic.sh $c $O/pmc3_$c > $O/pmc3_$c.log 2>&1; grep -A8 "attn_fwd" $O/pmc3_$c.log | grep -E "attn_fwd|FETCH|WRITE_SIZE" | head -3
done
