#!/bin/bash
# round 4, GPU call 9: kernel trace of the default bench command restricted to the timed steps (what a steady-state image is made of)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; R=$PWD
timeout 600 python -m pytest tests/test_norm_gpu.py -m gpu -q --timeout 600 2>&1 | tail -3
OUT=/tmp/pww_prof_r04i; rm -rf $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $R/$O/r4i_bench_prof.json 2> $R/$O/r4i_bench_prof.log) || true
DB=$(find $OUT -name "*.db" | head -1)
W=$(grep "timed region CLOCK_MONOTONIC" $O/r4i_bench_prof.log | sed 's/.*ns //')
echo "window $W"
{ echo "# Kernels of the TIMED steps of \`bench.py --steps 2 --warmup 1\` (config 2, fused block norms), rocprofv3 --kernel-trace"; python tools/rocpd_stats.py "$DB" --top 60 --window $W; } > $O/r04_steady_kernels.md 2>&1
head -50 $O/r04_steady_kernels.md | cut -c1-220
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4i_c2.json 2> $O/r4i_c2.log; echo "bench exit $?"; python -c "
import json;d=json.loads(open('$O/r4i_c2.json').read().strip().splitlines()[-1]);print('config 2 without profiler:', d['value'],d['ms_per_step'])"
