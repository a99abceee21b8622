#!/bin/bash
# round 3, GPU call 14: config 4 end to end, repeated, with the round's cross-attention changes switched off one at a time
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
for v in "X=0" "PWW_CROSS_GATE_WEIGHT=1" "PWW_CROSS_BIAS_LDS=1"; do
  env $v timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-roofline-pass > $O/r3n_c4_$v.json 2> $O/r3n_c4_$v.log
  echo "$v: $(tail -1 $O/r3n_c4_$v.json | cut -c1-160)"; grep "timed region" $O/r3n_c4_$v.log
done
