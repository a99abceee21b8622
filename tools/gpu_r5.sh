#!/bin/bash
# Round-5 GPU runner: `bash tools/gpu_r5.sh <stage> [...]`, stages run in the order given; everything lands under gpurun_out/r5_<stage>*.
#   native     tests/native/attn_check --quick (PASS count, FAIL lines) + TIME lines of the product shapes
#   small      tools/time_small_attn.py under the library variants (one process per PWW_DEBUG setting)
#   timeline   phase time stamps of the small launches
#   outproj    row f-1: tests, timing table and in-kernel time line of pww_cross_attn_fwd_parts_out
#   subset     the GPU tests that touch the attention path
#   pytest     the whole GPU suite
#   bench      python bench.py (short) -> r5_bench.json
#   benchfull  python bench.py with its defaults
#   prof       rocprofv3 --kernel-trace --stats of a short bench run -> r5_bench_c2_kernel_stats.md (pww kernels of the whole process + every kernel of the TIMED steps)
#   configs    bench lines of BASELINE configs 3 / 4 / 5
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out
for stage in "$@"; do
case $stage in
native)
  (cd tests/native && timeout 600 ./attn_check --quick > ../../$O/r5_native.log 2>&1; grep -c "^PASS" ../../$O/r5_native.log; grep -v "^PASS\|^TIME\|^TIMELINE" ../../$O/r5_native.log | tail -15)
  (cd tests/native && for c in qproj_sd15_n4096_b2 qproj_sd15_n1024_b2 qproj_sd15_n256_b2 qproj_sd15_n64_b2 qproj_sd15_n4096_b16 qproj_sd15_n256_b16 qproj_sd21_n576_b8; do timeout 120 ./attn_check --only $c 2>&1 | grep "^TIME\|^FAIL"; done) | tee $O/r5_native_time.log | cut -c1-400
  ;;
small)
  rm -f $O/r5_small.md
  for v in "" "cross_lean=0"; do PWW_DEBUG="$v" timeout 300 python tools/time_small_attn.py cross --out $O/r5_small.md 2>&1 | grep -v amdgpu.ids | tail -6; done
  for v in "" "attn_ksplit1=1"; do PWW_DEBUG="$v" timeout 300 python tools/time_small_attn.py self --out $O/r5_small.md 2>&1 | grep -v amdgpu.ids | tail -6; done
  timeout 300 python tools/time_small_attn.py toout --out $O/r5_small.md 2>&1 | grep -v amdgpu.ids | tail -6
  ;;
outproj)
  # row f-1: attention + to_out in one launch against the two-launch route (profiles/r05_to_out_epilogue.md)
  timeout 300 python -m pytest tests/test_round5_gpu.py -m gpu -q -k "to_out" 2>&1 | tail -3
  timeout 300 python tools/time_small_attn.py outproj --out $O/r5_outproj.md 2>&1 | grep -v amdgpu.ids | tail -8
  timeout 120 python tools/timeline_out.py 2 8 16 2>&1 | grep TIMELINE | tee $O/r5_outproj_timeline.log
  ;;
timeline)
  (cd tests/native && for c in qproj_sd15_n256_b2 qproj_sd15_n4096_b2; do timeout 120 ./attn_check --timeline --only $c 2>&1 | grep "^TIMELINE"; done; for c in sd15_self_n1024_d80 sd15_self_n256_d160; do timeout 120 ./attn_check --timeline --only $c 2>&1 | grep "^TIMELINE"; done) > $O/r5_timeline.log 2>&1; tail -60 $O/r5_timeline.log | cut -c1-200
  ;;
subset)
  timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_qproj_gpu.py tests/test_round3_gpu.py tests/test_round5_gpu.py -m gpu -q -x 2>&1 | tail -40 | tee $O/r5_subset.log
  ;;
pytest)
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $O/r5_pytest.log
  ;;
bench)
  timeout 600 python bench.py --steps 3 --warmup 2 --cpu-steps 0 --no-reference-ops > $O/r5_bench.json 2> $O/r5_bench.log; tail -3 $O/r5_bench.log; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("attention_path"), d["roofline"]["frac"])
for r in d.get("kernels", []):
    print(r.get("kernel")[:60], r.get("N"), r.get("D"), r.get("launches"), r.get("avg_us"), r.get("avg_us_back_to_back"))
PY
  ;;
benchfull)
  timeout 900 python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.log; tail -2 $O/r5_bench_default.log; tail -1 $O/r5_bench_default.json | cut -c1-300
  ;;
prof)
  OUT=/tmp/pww_prof_r05; rm -rf $OUT; R=$PWD
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters > $R/$O/r5_bench_c2_prof.json 2> $R/$O/r5_bench_c2_prof.log) || true
  DB=$(find $OUT -name "*.db" | head -1)
  W=$(grep "timed region CLOCK_MONOTONIC" $O/r5_bench_c2_prof.log | sed 's/.*ns //')
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters (round 5)"; echo; echo "## pww kernels, whole process (workload + roofline pass)";
    python tools/rocpd_stats.py "$DB" --top 60 --grid --match pww --split-b2b attn_fwd_fold_kernel; echo; echo "## pww kernels of the TIMED steps (hipGraph replay: what the product pays per launch)"; python tools/rocpd_stats.py "$DB" --top 60 --grid --match pww --window $W; echo; echo "## every kernel of the TIMED steps"; python tools/rocpd_stats.py "$DB" --top 45 --grid --window $W; } > $O/r5_bench_c2_kernel_stats.md 2>&1
  tail -1 $O/r5_bench_c2_prof.json | cut -c1-200; grep -c "" $O/r5_bench_c2_kernel_stats.md
  ;;
configs)
  for c in 3 4 5; do
    timeout 900 python bench.py --config $c --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops --no-live-counters > $O/r5_bench_c$c.json 2> $O/r5_bench_c$c.log; tail -1 $O/r5_bench_c$c.json | cut -c1-200
  done
  ;;
*) echo "unknown stage $stage";;
esac
done
