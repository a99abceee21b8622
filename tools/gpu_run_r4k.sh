#!/bin/bash
# round 4, GPU call 11: tests touched since call 10, steady-state kernel table of config 4 (8 images per step, 16 folded rows)
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; R=$PWD
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_qproj_gpu.py -m gpu -q --timeout 600 -k "bias_hints or product_path or fused_handoff or error_word or torchrun or rank_count" 2>&1 | tail -4
OUT=/tmp/pww_prof_r04k; rm -rf $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --config 4 --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops > $R/$O/r4k_c4_prof.json 2> $R/$O/r4k_c4_prof.log) || true
DB=$(find $OUT -name "*.db" | head -1)
W=$(grep "timed region CLOCK_MONOTONIC" $O/r4k_c4_prof.log | sed 's/.*ns //')
{ echo "# Kernels of the TIMED step of \`bench.py --config 4 --steps 1 --warmup 1\` (SD1.5-inpainting, 8 images = 16 folded rows, block plug on), rocprofv3 --kernel-trace"; python tools/rocpd_stats.py "$DB" --top 50 --window $W; } > $O/r04_steady_kernels_c4.md 2>&1
head -45 $O/r04_steady_kernels_c4.md | cut -c1-200
