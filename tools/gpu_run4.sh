#!/bin/bash
# round 3, GPU call 4: scalar words through pww_store_f32 (graph-mode tests), HIP_FORCE_DEV_KERNARG A/B (bench + kernel time lines),
# bench lines of configs 3/4/5, HBM traffic passes
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_loop_gpu.py -m gpu -q -k "one_graph or tiny_loop or repeat_calls or plms or config3_lms50 or error_word" --timeout 600 > $O/r3d_pytest_graph.log 2>&1; tail -3 $O/r3d_pytest_graph.log
for k in 0 1; do
  HIP_FORCE_DEV_KERNARG=$k timeout 600 python bench.py --steps 6 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r3d_bench_kernarg$k.json 2> $O/r3d_bench_kernarg$k.log
  echo "HIP_FORCE_DEV_KERNARG=$k: $(tail -1 $O/r3d_bench_kernarg$k.json | cut -c1-130)"
done
for k in 0 1; do
  for c in sd15_cross_n4096_d40_cols32 sd15_cross_n256_d160_cols32 sd15_self_n256_d160 sd15_mid_self_n64_d160 sd15_self_n1024_d80; do
    echo "== HIP_FORCE_DEV_KERNARG=$k $c"; HIP_FORCE_DEV_KERNARG=$k timeout 120 $H --timeline --only $c | grep -E "TIMELINE|TIME "
  done
done > $O/r3d_timeline_kernarg.log 2>&1
bash tools/gpu_profile.sh configs pmc
