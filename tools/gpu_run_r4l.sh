#!/bin/bash
# round 4, GPU call 14: does PyTorch's TunableOp (GEMM solution search across hipBLASLt / rocBLAS, a stock-op setting like MIOpen's find mode) move the headline?
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
S=$(date +%s)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop_c2.csv \
  timeout 1200 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4l_c2_tuned.json 2> $O/r4l_c2_tuned.log; echo "tuned bench exit $? in $(( $(date +%s) - S )) s"
python -c "
import json;d=json.loads(open('$O/r4l_c2_tuned.json').read().strip().splitlines()[-1]);print('tuning run:', d['value'],d['ms_per_step'],d['config'].get('warmup_s'))"
ls -la $O/tunableop_c2*.csv; wc -l $O/tunableop_c2*.csv | tail -1
F=$(ls $O/tunableop_c2*.csv | head -1)
S=$(date +%s)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=$PWD/$F \
  timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4l_c2_replay.json 2> $O/r4l_c2_replay.log; echo "replay bench exit $? in $(( $(date +%s) - S )) s"
python -c "
import json;d=json.loads(open('$O/r4l_c2_replay.json').read().strip().splitlines()[-1]);print('tuned file, no tuning:', d['value'],d['ms_per_step'],d['config'].get('warmup_s'))"
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r4l_c2_plain.json 2> $O/r4l_c2_plain.log
python -c "
import json;d=json.loads(open('$O/r4l_c2_plain.json').read().strip().splitlines()[-1]);print('plain:', d['value'],d['ms_per_step'],d['config'].get('warmup_s'))"
