#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as the guide prescribes) for one native-harness case.
R=$PWD; CASE=${1:-sd15_self_n4096_d40_f16_b2}; OUT=$R/${2:-gpurun_out/pmc_traffic}; FLAGS=${3:-}   # e.g. --product-only
export TMPDIR=/tmp; cd /tmp; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -o pmc -- $R/tests/native/attn_check $FLAGS --only $CASE > $OUT/$n.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, d in agg.items():
    if 'attn_fwd' not in k and 'qk_reduce_kernel' not in k and 'cross_fused' not in k and 'cross_lean' not in k: continue
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    print(k[:70]); [print('   %-26s %14.1f (n=%d)' % (c, sum(v) / len(v), len(v))) for c, v in sorted(d.items())]
json.dump(res, open(out + '/traffic.json', 'w'), indent=1)
PY
