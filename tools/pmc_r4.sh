#!/bin/bash
# round 4: issue / matrix-pipe / LDS counters and HBM-side traffic of the kernels that ship, one native-harness
# case at a time, separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md: SQ 8 slots, FETCH_SIZE and WRITE_SIZE apart).
#   tools/pmc_r4.sh <outdir> <case> [flags] ...      (flags: "--product-only" or "-")
R=$PWD; OUT=$R/$1; shift
export TMPDIR=/tmp; cd /tmp; mkdir -p $OUT
while [ $# -gt 0 ]; do
  CASE=$1; FLAGS=$2; shift 2; [ "$FLAGS" == "-" ] && FLAGS=""
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$CASE/p$i -o pmc -- $R/tests/native/attn_check $FLAGS --only $CASE > $OUT/$CASE.p$i.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res = {}
for case in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, case)): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, case) + '/p*/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, d in agg.items():
        if not any(t in k for t in ('attn_fwd', 'cross_fused', 'qproj_stat', 'qk_reduce_kernel')): continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        m['dispatches'] = max(len(v) for v in d.values())
        g = m.get('GRBM_GUI_ACTIVE')
        if g and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            m['kernel_cycles'] = g / 8.0                                                  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            m['mfma_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (g / 8.0 * 1024.0)    # 256 CUs x 4 SIMDs
        if 'SQ_WAVE_CYCLES' in m:
            for c in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'):
                if c in m: m[c + '_frac_of_wave_cycles'] = m[c] / m['SQ_WAVE_CYCLES']
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            m['hbm_side_bytes'] = (2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0      # a read request moves a 128-byte line, FETCH_SIZE tallies 64 (profiles/r06_linefill.txt; rounds 4 - 5 summed them 1 : 1)
        res.setdefault(case, {})[k[:110]] = m
        print(case, '|', k[:90])
        for c, v in sorted(m.items()): print('     %-34s %16.4f' % (c, v))
json.dump(res, open(out + '/pmc.json', 'w'), indent=1)
PY
