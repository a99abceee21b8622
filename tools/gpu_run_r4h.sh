#!/bin/bash
# round 4, GPU call 8: GroupNorm kernels -- parity tests, A/B of the headline with and without them, kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_norm_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4h_c2_fused.json 2> $O/r4h_c2_fused.log; echo "bench fused exit $?"; python -c "
import json;d=json.loads(open('$O/r4h_c2_fused.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['block_norms'][:30])"
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-fused-norm > $O/r4h_c2_stock.json 2> $O/r4h_c2_stock.log; echo "bench stock exit $?"; python -c "
import json;d=json.loads(open('$O/r4h_c2_stock.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['block_norms'][:30])"
tail -3 $O/r4h_c2_fused.log | cut -c1-300
