export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_norm_gpu.py tests/test_round4_gpu.py -m gpu -q -x 2>&1 | tail -4
for v in 1 0 1 0; do PWW_FUSE_FF_RESIDUAL=$v timeout 300 python bench.py --steps 4 --warmup 2 --cpu-steps 0 --no-reference-ops --no-live-counters --no-roofline-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_FF_RESIDUAL=$v', d['value'], d['ms_per_step'])"; done
