#!/bin/bash
# round 3, GPU call 13: final validation of the build (reworked multi-block fused cross-attention) + the round's bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -rP --timeout 900 --durations=5 > $O/r3m_pytest.log 2>&1; echo "pytest exit $?" >> $O/r3m_pytest.log
tail -3 $O/r3m_pytest.log; grep "^FAILED" $O/r3m_pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r3m_smoke.log 2>&1; tail -2 $O/r3m_smoke.log
timeout 900 python bench.py > $O/r3m_bench_c2.json 2> $O/r3m_bench_c2.log; tail -1 $O/r3m_bench_c2.json | cut -c1-200
bash tools/gpu_profile.sh trace configs
