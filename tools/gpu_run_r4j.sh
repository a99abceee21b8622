#!/bin/bash
# round 4, GPU call 10: full GPU suite, smoke, and the four BASELINE configs with the block plug in place
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > $O/r4j_pytest.log; cat $O/r4j_pytest.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r04b_bench_c$c.json 2> $O/r04b_bench_c$c.log; echo "bench c$c exit $?"
  python -c "
import json;d=json.loads(open('$O/r04b_bench_c$c.json').read().strip().splitlines()[-1]);print('config $c:', d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('avg_us'))"
done
