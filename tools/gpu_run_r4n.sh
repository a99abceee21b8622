#!/bin/bash
# round 4, GPU call 16: MIOpen's split-K igemm convolutions bring two tensor-op launches each (zero + cast): is the solver family worth them?
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
run() { timeout 600 env "$@" python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4n.json 2> $O/r4n.log; python -c "
import json;d=json.loads(open('$O/r4n.json').read().strip().splitlines()[-1]);print('$*', d['value'],d['ms_per_step'],d['config'].get('warmup_s'))"; }
run X=0
run MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_FWD_GTC_XDLOPS_NHWC=0
run MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_FWD_GTC_XDLOPS_NHWC=0 MIOPEN_USER_DB_PATH=/tmp/miopen_alt
