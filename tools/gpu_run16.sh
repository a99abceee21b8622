#!/bin/bash
# round 3, GPU call 16: the python-level gated-images variants + config 3 with two timed steps
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q --timeout 600 -k "bias_hints or compact" > $O/r3o_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3o_pytest.log; tail -3 $O/r3o_pytest.log
timeout 600 python bench.py --config 3 --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-roofline-pass > $O/r3o_c3.json 2> $O/r3o_c3.log; tail -1 $O/r3o_c3.json | cut -c1-200
