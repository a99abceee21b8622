#!/bin/bash
# round 3 profiles: (1) rocprofv3 kernel trace of the default bench command -> pww kernel table, (2) bench lines of configs 3/4/5 with the
# shipped defaults, (3) HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the dominant launches and the batched cross-attention launch.
#   tools/gpu_profile.sh [trace] [configs] [pmc]
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; R=$PWD
what="${@:-trace configs pmc}"
if [[ $what == *trace* ]]; then
  OUT=/tmp/pww_prof_r03; rm -rf $OUT
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $R/$O/r03_bench_c2_prof.json 2> $R/$O/r03_bench_c2_prof.log) || true
  DB=$(find $OUT -name "*.db" | head -1)
  { python tools/rocpd_stats.py "$DB" --top 40 --grid --match pww --split-b2b attn_fwd_fold_kernel; echo; echo "## all kernels, top 25"; python tools/rocpd_stats.py "$DB" --top 25 --grid; } > $O/r03_bench_c2_kernel_stats.md 2>&1
  head -12 $O/r03_bench_c2_kernel_stats.md
fi
if [[ $what == *configs* ]]; then
  for c in 3 4 5; do
    timeout 900 python bench.py --config $c --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r03_bench_c$c.json 2> $O/r03_bench_c$c.log; tail -1 $O/r03_bench_c$c.json | cut -c1-300
  done
fi
if [[ $what == *pmc* ]]; then
  for c in sd15_self_n4096_d40_bf16_b2 sd15_self_n4096_d40_f16_b16 sd21_self_n9216_d64_b8 sd15_cross_n4096_d40_b16_cols32 sd15_cross_n4096_d40_cols32; do
    bash tools/pmc_traffic.sh $c $O/pmc_$c > $O/pmc_$c.log 2>&1; tail -12 $O/pmc_$c.log
  done
fi
