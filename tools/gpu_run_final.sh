#!/bin/bash
# final validation of the round: what the driver runs (GPU suite, smoke, default bench line)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; R=$PWD
if [ "$1" == "full" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 > $O/final_pytest.log
else
  timeout 1200 python -m pytest tests/test_norm_gpu.py tests/test_loop_gpu.py tests/test_round4_gpu.py -m gpu -q --timeout 900 2>&1 | tail -6 > $O/final_pytest_subset.log; cat $O/final_pytest_subset.log | cut -c1-200
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
S=$(date +%s); timeout 900 python bench.py > $O/final_bench.json 2> $O/final_bench.log; echo "default bench exit $? in $(( $(date +%s) - S )) s"
python -c "
import json;d=json.loads(open('$O/final_bench.json').read().strip().splitlines()[-1]);print('default bench:', d['value'],d['ms_per_step'],d['steps'],d['warmup'],d['roofline']['frac'],d['roofline'].get('traffic'),d.get('cpu_baseline'))" | cut -c1-500
