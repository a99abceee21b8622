#!/bin/bash
# round 4, GPU call 7: does the kernel-argument placement (HIP_FORCE_DEV_KERNARG) change the small launches' start-up?
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 300 $H --only qproj_sd15_n4096_b2 --timeline > $O/r4g_kernarg$v.log 2>&1
  echo "HIP_FORCE_DEV_KERNARG=$v"; grep -E "^TIME|stamp [016]:" $O/r4g_kernarg$v.log | cut -c1-200
done
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python tools/time_ingraph.py $O/r4g_ingraph_kernarg$v.md > /dev/null 2>&1; echo "HIP_FORCE_DEV_KERNARG=$v"; cut -c1-200 $O/r4g_ingraph_kernarg$v.md | tail -6
done
