#!/bin/bash
# Round-6 GPU runner: `bash tools/gpu_r6.sh <stage> [...]`; everything lands under gpurun_out/r6_<stage>*.
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
SHORT="--steps 3 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters"
summ() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ap = d.get("attention_path") or {}
print(sys.argv[1], "img/s", d["value"], "ms", d["ms_per_step"], "parity", (d.get("parity") or {}).get("rel_l2"), "dom", ap.get("dominant_us_per_forward"), "others", ap.get("others_us_per_forward"), "b2b", ap.get("others_us_per_forward_back_to_back"))
for r in d.get("kernels", []):
    if "N" in r:
        print("   %-44s N=%-5s D=%-4s n=%-4s us=%-7s b2b=%s" % (r["kernel"][:44], r.get("N"), r.get("D"), r.get("launches"), r.get("avg_us"), r.get("avg_us_back_to_back")))
PY
}
for stage in "$@"; do
case $stage in
kernarg)
  # cold start of the small launches: where do the kernel arguments live? (HIP_FORCE_DEV_KERNARG: device memory instead of host-coherent memory)
  for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v timeout 400 python bench.py $SHORT > $O/r6_bench_devkernarg$v.json 2> $O/r6_bench_devkernarg$v.log; summ $O/r6_bench_devkernarg$v.json; done
  ;;
qprojroute)
  PWW_QPROJ_STAT=0 timeout 400 python bench.py $SHORT > $O/r6_bench_qproj0.json 2> $O/r6_bench_qproj0.log; summ $O/r6_bench_qproj0.json
  ;;
bench)
  timeout 400 python bench.py $SHORT > $O/r6_bench.json 2> $O/r6_bench.log; tail -2 $O/r6_bench.log; summ $O/r6_bench.json
  ;;
benchfull)
  timeout 900 python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.log; tail -2 $O/r6_bench_default.log; summ $O/r6_bench_default.json
  ;;
native)
  (cd tests/native && timeout 900 ./attn_check --quick > ../../$O/r6_native.log 2>&1; grep -c "^PASS" ../../$O/r6_native.log; grep -v "^PASS\|^TIME\|^TIMELINE" ../../$O/r6_native.log | tail -15)
  ;;
pytest)
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/r6_pytest.log
  ;;
subset)
  timeout 1200 python -m pytest tests/test_attention_gpu.py tests/test_qproj_gpu.py tests/test_round3_gpu.py tests/test_round5_gpu.py tests/test_round6_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/r6_subset.log
  ;;
configs)
  for c in 3 4 5; do
    timeout 900 python bench.py --config $c --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters > $O/r6_bench_c$c.json 2> $O/r6_bench_c$c.log; summ $O/r6_bench_c$c.json
  done
  ;;
k1route)
  # pww_qproj_stat (K1) at 2 rows against the stock to_q GEMM + pww_qk_parts, end to end, alternating runs on ONE box
  rm -f $O/r6_k1route.txt
  for i in 1 2 3; do for v in 1 0; do
    PWW_QPROJ_STAT=$v timeout 300 python bench.py --steps 6 --warmup 1 --no-roofline-pass --no-reference-ops --cpu-steps 0 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PWW_QPROJ_STAT=$v', d['value'], d['ms_per_step'])" | tee -a $O/r6_k1route.txt
  done; done
  ;;
halftl)
  # time line of the N = 1024 d = 80 self-attention, 2 x 4 half-tile key groups against 2 x 2
  (cd tests/native && for v in "" "attn_ksplit_half=0"; do echo "PWW_DEBUG=$v"; PWW_DEBUG="$v" timeout 120 ./attn_check --timeline --only sd15_self_n1024_d80 2>&1 | grep "^TIMELINE\|^PASS\|^FAIL"; done) | tee $O/r6_half_timeline.log | cut -c1-220
  ;;
cold)
  # what a small launch costs over operands that are not L2-resident (rotating buffer sets)
  timeout 600 python tools/time_cold_start.py 2>&1 | grep "^|" | tee $O/r6_cold_start.md
  ;;
preload)
  # kernarg preload for every small pww kernel (K5 block kernels, mask / CFG kernels, pww_qk_parts): the same sources built without the flag, alternating runs
  rm -f $O/r6_preload_ab.txt
  for i in 1 2 3; do for v in preload nopreload; do
    L=""; [ $v = nopreload ] && L="$PWD/paint-with-words-sd_amd/build/ab/libpww_hip_nopreload.so"
    PWW_HIP_LIB=${L:-$PWD/paint-with-words-sd_amd/pww_hip/libpww_hip.so} timeout 300 python bench.py --steps 6 --warmup 1 --no-roofline-pass --no-reference-ops --cpu-steps 0 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['parity']['rel_l2'])" | tee -a $O/r6_preload_ab.txt
  done; done
  ;;
prof)
  OUT=/tmp/pww_prof_r06; rm -rf $OUT; R=$PWD
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters > $R/$O/r6_bench_c2_prof.json 2> $R/$O/r6_bench_c2_prof.log) || true
  DB=$(find $OUT -name "*.db" | head -1)
  W=$(grep "timed region CLOCK_MONOTONIC" $O/r6_bench_c2_prof.log | sed 's/.*ns //')
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters (round 6)"; echo; echo "## pww kernels, whole process (workload + roofline pass)";
    python tools/rocpd_stats.py "$DB" --top 60 --grid --match pww --split-b2b attn_fwd_fold_kernel; echo; echo "## pww kernels of the TIMED steps (hipGraph replay: what the product pays per launch)"; python tools/rocpd_stats.py "$DB" --top 60 --grid --match pww --window $W; echo; echo "## every kernel of the TIMED steps"; python tools/rocpd_stats.py "$DB" --top 45 --grid --window $W; } > $O/r6_bench_c2_kernel_stats.md 2>&1
  tail -1 $O/r6_bench_c2_prof.json | cut -c1-200; grep -c "" $O/r6_bench_c2_kernel_stats.md
  ;;
profc3)
  # config 3 (fp16, 16 folded rows: the throughput mode) under the kernel trace: where the time of the batched workload goes
  OUT=/tmp/pww_prof_r06_c3; rm -rf $OUT; R=$PWD
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --config 3 --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters --no-roofline-pass > $R/$O/r6_bench_c3_prof.json 2> $R/$O/r6_bench_c3_prof.log) || true
  DB=$(find $OUT -name "*.db" | head -1)
  W=$(grep "timed region CLOCK_MONOTONIC" $O/r6_bench_c3_prof.log | sed 's/.*ns //')
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config 3 --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters --no-roofline-pass (round 6)"; echo; echo "## pww kernels of the TIMED step (8 images, 16 folded rows)"; python tools/rocpd_stats.py "$DB" --top 60 --grid --match pww --window $W; echo; echo "## every kernel of the TIMED step"; python tools/rocpd_stats.py "$DB" --top 45 --grid --window $W; } > $O/r6_bench_c3_kernel_stats.md
  tail -1 $O/r6_bench_c3_prof.json | cut -c1-200; grep -c "" $O/r6_bench_c3_kernel_stats.md
  ;;
igemm)
  # stock-op setting A/B: MIOpen's bf16 NHWC asm implicit-GEMM forward solver brings two tensor-op launches per convolution (fp32 workspace
  # zero + cast: 1736 + 1736 launches per 2 images in the trace); with the solver disabled find mode picks among the others
  rm -f $O/r6_igemm_ab.txt
  for i in 1 2; do for v in 1 0; do
    MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=$v timeout 400 python bench.py --steps 6 --warmup 1 --no-roofline-pass --no-reference-ops --cpu-steps 0 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ASM_FWD_GTC_XDLOPS_NHWC=$v', d['value'], d['ms_per_step'], d['parity']['rel_l2'], d['config']['warmup_s'])" | tee -a $O/r6_igemm_ab.txt
  done; done
  ;;
hot)
  # the fp16 dominant launch on hot logits -- shipped / round 5's behaviour / always lazy / a raised magnitude-guard limit
  rm -f $O/r6_hot.md
  for v in "" "attn_hot_sum=0" "attn_hot_sum=-1" "attn_fold_limit_f16=44" "attn_fold_limit_f16=52"; do PWW_DEBUG="$v" timeout 600 python tools/diag_hot_f16.py 2>&1 | grep "^|" | tee -a $O/r6_hot.md; echo >> $O/r6_hot.md; done
  ;;
sectors)
  # d = 40: heads 2m / 2m + 1 share a 32-byte sector of every Q / K / V / O row. A/B of the workgroup orders that keep them on one XCD:
  # HBM-side traffic (FETCH_SIZE / WRITE_SIZE passes) and time of the dominant self-attention launch (2 and 16 rows) and the 16-row cross launch
  rm -f $O/r6_sectors.log
  for v in "" "attn_head_pairs=0,cross_head_major=0"; do
    echo "=== PWW_DEBUG=$v" | tee -a $O/r6_sectors.log
    (cd tests/native && for c in sd15_self_n4096_d40_bf16_b2 sd15_self_n4096_d40_f16_b16 qproj_sd15_n4096_b16; do PWW_DEBUG="$v" timeout 200 ./attn_check --only $c 2>&1 | grep "^TIME\|^FAIL"; done) | cut -c1-330 | tee -a $O/r6_sectors.log
    t=default; [ -n "$v" ] && t=round5
    for c in sd15_self_n4096_d40_bf16_b2 sd15_self_n4096_d40_f16_b16 qproj_sd15_n4096_b16; do
      f=""; [ $c == qproj_sd15_n4096_b16 ] && f="--product-only"
      PWW_DEBUG="$v" bash tools/pmc_traffic.sh $c gpurun_out/pmc_sectors_${t}_$c $f 2>&1 | sed "s/^/$c: /" | tee -a $O/r6_sectors.log
    done
  done
  ;;
sectors4)
  # groups of FOUR heads per XCD at 16 rows (6 line fills per row instead of 8; 2.6 MB of K / V per group and L2) against groups of two; and the
  # small self-attention launches (d = 80 / 160: 160- / 320-byte slices share lines too) under the new order against round 5's
  rm -f $O/r6_sectors4.log
  for v in "attn_head_pairs=4" "" "attn_head_pairs=0"; do
    echo "=== PWW_DEBUG=$v" | tee -a $O/r6_sectors4.log
    (cd tests/native && for c in sd15_self_n4096_d40_f16_b16 sd15_self_n4096_d40_bf16_b16 sd15_self_n4096_d40_bf16_b2 sd15_self_n1024_d80 sd15_self_n256_d160; do PWW_DEBUG="$v" timeout 200 ./attn_check --only $c 2>&1 | grep "^TIME\|^FAIL"; done) | cut -c1-200 | tee -a $O/r6_sectors4.log
  done
  for v in "attn_head_pairs=4"; do
    PWW_DEBUG="$v" bash tools/pmc_traffic.sh sd15_self_n4096_d40_f16_b16 gpurun_out/pmc_sectors_quads_f16_b16 2>&1 | grep -A12 fold_kernel | tee -a $O/r6_sectors4.log
  done
  for v in "" "attn_head_pairs=0"; do PWW_DEBUG="$v" timeout 300 python tools/time_small_attn.py self --out $O/r6_sectors4_small.md 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/r6_sectors4.log; done
  ;;
leanmulti)
  # the batched cross launch (16 rows x 4096 tokens): the small kernel walking several blocks per workgroup (default) against the general kernel
  rm -f $O/r6_leanmulti.log
  for v in "" "cross_lean_multi=0"; do
    echo "=== PWW_DEBUG=$v" | tee -a $O/r6_leanmulti.log
    (cd tests/native && for c in qproj_sd15_n4096_b16 qproj_sd15_n4096_b16_f16 qproj_sd21_n9216_b8; do PWW_DEBUG="$v" timeout 200 ./attn_check --only $c 2>&1 | grep "^TIME\|^FAIL\|^PASS"; done) | cut -c1-330 | tee -a $O/r6_leanmulti.log
    PWW_DEBUG="$v" timeout 300 python tools/diag_wg_order.py 2>&1 | grep "CASE cross" | tee -a $O/r6_leanmulti.log
  done
  PWW_DEBUG="" bash tools/pmc_traffic.sh qproj_sd15_n4096_b16 gpurun_out/pmc_leanmulti --product-only 2>&1 | grep -A12 "cross_lean\|cross_fused" | tee -a $O/r6_leanmulti.log
  ;;
linefill)
  # what one TCC_EA0_RDREQ moves (tools/ubench_linefill.cpp): time per touched line for full / half / sector / head-slice reads of a 2 GiB buffer,
  # and the request counters of the same kernels
  timeout 300 tools/ubench_linefill 2048 | tee $O/r6_linefill.txt
  export TMPDIR=/tmp; R=$PWD; cd /tmp
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_linefill/$n -o pmc -- $R/tools/ubench_linefill 2048 > $R/gpurun_out/pmc_linefill_$n.log 2>&1
  done
  cd $R
  python3 - <<'PY' | tee -a $O/r6_linefill.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_linefill/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'reader' in row['Kernel_Name']: agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in sorted(agg.items()):
    # the LAST dispatch of each kernel is a full-size one (the first is the 1/16 warm-up)
    print(k[:60], {c: max(v) for c, v in sorted(d.items())})
PY
  ;;
smallself)
  # self-attention N = 1024 d = 80 at 2 rows: 2 x 4 half-tile key groups (default) against round 5's 2 x 2
  rm -f $O/r6_smallself.md
  for v in "" "attn_ksplit_half=0"; do PWW_DEBUG="$v" timeout 300 python tools/time_small_attn.py self --out $O/r6_smallself.md 2>&1 | grep -v amdgpu.ids | tail -6; done
  timeout 600 python -m pytest tests/test_round5_gpu.py tests/test_attention_gpu.py -m gpu -q -x -k "small_self or self or attention" 2>&1 | grep -v amdgpu.ids | tail -5
  ;;
*) if [ -f "tools/gpu_r6_$stage.sh" ]; then bash tools/gpu_r6_$stage.sh; else echo "unknown stage $stage"; fi;;
esac
done
