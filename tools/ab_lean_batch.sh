# A/B of the pass-2-only cross-attention kernel choice on the batched configs (PWW_DEBUG=cross_lean: 0 general kernel everywhere, 1 default, 2 small kernel everywhere)
export TMPDIR=/tmp
for c in 4 3; do for v in 1 0 2 1 0; do PWW_DEBUG=cross_lean=$v timeout 400 python bench.py --config $c --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters --no-roofline-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c cross_lean=$v', d['value'], d['ms_per_step'])"; done; done
(cd tests/native && for c in qproj_sd15_n1024_b16 qproj_sd15_n4096_b16; do for v in 1 0 2; do echo "cross_lean=$v"; PWW_DEBUG=cross_lean=$v timeout 120 ./attn_check --only $c 2>&1 | grep "^TIME" | cut -c1-330; done; done)
