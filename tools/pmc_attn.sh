#!/bin/bash
# PMC passes for one attention case of the native harness (run on the GPU box via gpurun).
# usage: tools/pmc_attn.sh <case-name> <outdir>
R=$PWD; CASE=${1:-sd15_self_n4096_d40_f16_b2}; OUT=$R/${2:-gpurun_out/pmc}
export TMPDIR=/tmp; cd /tmp
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- $R/tests/native/attn_check --only $CASE > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[(k, row['Counter_Name'])] += 1
for k, d in agg.items():
    if 'attn_fwd' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        n = cnt[(k, c)]
        print('   %-28s %16.0f per-dispatch (n=%d)' % (c, v / n, n))
PY
