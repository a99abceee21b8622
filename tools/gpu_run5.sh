#!/bin/bash
# round 3, GPU call 5: bench after the asynchronous error check, pair-major workgroup order A/B (time + fetched bytes), error-path tests
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check; R=$PWD
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q -rP -k "error_word or timeout_raises or inpaint_pipeline or runner_inpaint" --timeout 600 > $O/r3e_pytest_err.log 2>&1; tail -3 $O/r3e_pytest_err.log
timeout 600 python bench.py --steps 6 --warmup 1 --cpu-steps 0 --no-reference-ops > $O/r3e_bench_c2.json 2> $O/r3e_bench_c2.log; tail -1 $O/r3e_bench_c2.json | cut -c1-140
timeout 600 python bench.py --steps 6 --warmup 1 --cpu-steps 0 --no-reference-ops --no-roofline-pass > $O/r3e_bench_c2_b.json 2> $O/r3e_bench_c2_b.log; tail -1 $O/r3e_bench_c2_b.json | cut -c1-140
for c in sd15_self_n4096_d40_f16_b16 sd15_self_n4096_d40_bf16_b16 sd21_self_n9216_d64_b8 sd15_self_n4096_d40_bf16_b2 sd21_self_n9216_d64_b4 d64_self_n2304_bf16_b8; do
  for v in 0 16; do echo "== PWW_ATTN_PAIR_MAJOR=$v $c"; PWW_ATTN_PAIR_MAJOR=$v timeout 300 $H --only $c | grep -E "^TIME|FAIL|^PASS.*attn max"; done
done > $O/r3e_ab_pair_major.log 2>&1
for v in 0 16; do
  OUT=/tmp/pmc_pm$v; rm -rf $OUT
  (cd /tmp && PWW_ATTN_PAIR_MAJOR=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o pmc -- $R/$H --only sd15_self_n4096_d40_f16_b16 > /dev/null 2>&1)
  python3 - $OUT $v <<'PY'
import csv, glob, sys
vals=[]
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'attn_fwd_fold' in row['Kernel_Name'] and row['Counter_Name'] == 'FETCH_SIZE': vals.append(float(row['Counter_Value']))
print("PWW_ATTN_PAIR_MAJOR=%s sd15_self_n4096_d40_f16_b16 FETCH_SIZE mean %.1f MB over %d dispatches" % (sys.argv[2], sum(vals) / max(len(vals), 1) * 1024 / 1e6, len(vals)))
PY
done >> $O/r3e_ab_pair_major.log 2>&1
cat $O/r3e_ab_pair_major.log | grep -E "==|TIME|FETCH" | cut -c1-160
