#!/bin/bash
# round 4, GPU call 4: full GPU suite with the new default route, bench lines of configs 2 - 5, smoke
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 300 $H --match qproj > $O/r4d_native_qproj.log 2>&1; echo "exit $?" >> $O/r4d_native_qproj.log )
echo "qproj harness: PASS $(grep -c '^PASS' $O/r4d_native_qproj.log) FAIL $(grep -c '^FAIL' $O/r4d_native_qproj.log)"; grep "^FAIL" $O/r4d_native_qproj.log | head -5 | cut -c1-250
timeout 300 python tools/time_qproj.py $O/r04_qproj.md > $O/r4d_time_qproj.log 2>&1; grep -E "SD1.5" $O/r4d_time_qproj.log | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/r4d_pytest.log 2>&1; echo "pytest exit $?" >> $O/r4d_pytest.log
tail -4 $O/r4d_pytest.log; grep -E "^FAILED|^ERROR" $O/r4d_pytest.log | head -20 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r4d_smoke.log 2>&1; tail -2 $O/r4d_smoke.log
timeout 900 python bench.py > $O/r04_bench_c2.json 2> $O/r04_bench_c2.log; tail -c 1500 $O/r04_bench_c2.json | cut -c1-1500
for c in 3 4 5; do
  timeout 900 python bench.py --config $c --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r04_bench_c$c.json 2> $O/r04_bench_c$c.log; python - <<PY
import json
try:
    r=json.loads(open("$O/r04_bench_c$c.json").read().strip().splitlines()[-1])
    print("config $c:", r["value"], r["unit"], "ms/step", r["ms_per_step"], "roofline", {k:r.get("roofline",{}).get(k) for k in ("avg_us","frac")})
    for k in r.get("kernels",[])[:12]: print("   ", k.get("kernel")[:60], k.get("B"),k.get("N"),k.get("D"), k.get("launches"), k.get("avg_us"), k.get("frac"))
except Exception as e: print("config $c failed", e)
PY
done
