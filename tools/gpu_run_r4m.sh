#!/bin/bash
# round 4, GPU call 15: rocprofv3 --kernel-trace --stats of the bench command of the FINAL build: pww kernels of the whole process, the dominant kernel split
# into in-workload / back-to-back launches, and the timed steps alone
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; R=$PWD
OUT=/tmp/pww_prof_r04m; rm -rf $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $R/$O/r04b_bench_c2_prof.json 2> $R/$O/r04b_bench_c2_prof.log) || true
DB=$(find $OUT -name "*.db" | head -1)
W=$(grep "timed region CLOCK_MONOTONIC" $O/r04b_bench_c2_prof.log | sed 's/.*ns //')
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops (final build of round 4)"; echo; echo "## pww kernels, whole process (workload + roofline pass)"; python tools/rocpd_stats.py "$DB" --top 40 --grid --match pww --split-b2b attn_fwd_fold_kernel; echo; echo "## every kernel of the TIMED steps"; python tools/rocpd_stats.py "$DB" --top 70 --window $W; } > $O/r04b_bench_c2_kernel_stats.md 2>&1
grep -n "attn_fwd_fold_kernel" $O/r04b_bench_c2_kernel_stats.md | head -5 | cut -c1-250
python -c "
import json;d=json.loads(open('$O/r04b_bench_c2_prof.json').read().strip().splitlines()[-1]);print('under profiler:', d['value'],d['roofline']['avg_us'],d['roofline']['frac'])"
