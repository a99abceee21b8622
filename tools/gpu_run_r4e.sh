#!/bin/bash
# round 4, GPU call 5: in-graph penalty experiment, kernel trace of the default bench command, PMC passes of the final cross-attention route, thread-trace attempt
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check; R=$PWD
timeout 300 python tools/time_ingraph.py $O/r04_ingraph.md > $O/r4e_ingraph.log 2>&1; echo "ingraph exit $?"; cut -c1-400 $O/r4e_ingraph.log | tail -6
timeout 600 python -m pytest tests/test_round3_gpu.py -m gpu -q --timeout 600 -k "set_error_word" 2>&1 | tail -2
# thread trace (rocprofv3 --att): needs the trace decoder library, which this image does not ship -- recorded either way
( cd /tmp && timeout 120 rocprofv3 --att --att-target-cu 1 --kernel-trace -d /tmp/att_out -o att -- $R/$H --only sd15_self_n4096_d40_bf16_b2 > $R/$O/r4e_att.log 2>&1; echo "att exit $?" >> $R/$O/r4e_att.log )
tail -5 $O/r4e_att.log | cut -c1-300; ls /tmp/att_out 2>/dev/null | head
# kernel trace of the default bench command
OUT=/tmp/pww_prof_r04; rm -rf $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops > $R/$O/r04_bench_c2_prof.json 2> $R/$O/r04_bench_c2_prof.log) || true
DB=$(find $OUT -name "*.db" | head -1)
{ python tools/rocpd_stats.py "$DB" --top 40 --grid --match pww --split-b2b attn_fwd_fold_kernel; echo; echo "## all kernels, top 25"; python tools/rocpd_stats.py "$DB" --top 25 --grid; } > $O/r04_bench_c2_kernel_stats.md 2>&1
head -24 $O/r04_bench_c2_kernel_stats.md | cut -c1-250
# PMC of the final cross-attention route (product-only: 20 x (qproj_stat, cross_attn_fwd_parts))
bash tools/pmc_r4.sh $O/pmc_r4b qproj_sd15_n4096_b16 --product-only qproj_sd15_n4096_b2 --product-only qproj_sd15_n256_b2 --product-only qproj_sd21_n9216_b8 --product-only > $O/r4e_pmc.log 2>&1
grep -E "\||mfma_busy_frac|hbm_side_bytes|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE" $O/r4e_pmc.log | cut -c1-150 | head -60
