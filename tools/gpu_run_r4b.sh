#!/bin/bash
# round 4, GPU call 2: LDS-staged qproj_stat kernel: harness + route timing; the two fixed tests + config 5 end to end; PMC passes of the shipping kernels
export TMPDIR=/tmp
mkdir -p gpurun_out; O=gpurun_out; H=tests/native/attn_check
( timeout 400 $H --match qproj > $O/r4b_native_qproj.log 2>&1; echo "exit $?" >> $O/r4b_native_qproj.log )
echo "qproj harness: PASS $(grep -c '^PASS' $O/r4b_native_qproj.log) FAIL $(grep -c '^FAIL' $O/r4b_native_qproj.log)"; grep "^FAIL" $O/r4b_native_qproj.log | head -12 | cut -c1-330; tail -1 $O/r4b_native_qproj.log
grep "^TIME" $O/r4b_native_qproj.log | cut -c1-200
timeout 400 python tools/time_qproj.py $O/r04_qproj.md > $O/r4b_time_qproj.log 2>&1; echo "time_qproj exit $?"; cut -c1-220 $O/r4b_time_qproj.log | tail -22
timeout 1500 python -m pytest tests/test_qproj_gpu.py tests/test_round4_gpu.py -m gpu -q --timeout 900 -s -k "stale or orig_map or config5 or unsupported or hw384" > $O/r4b_pytest.log 2>&1; echo "pytest exit $?" >> $O/r4b_pytest.log
tail -4 $O/r4b_pytest.log; grep -E "^FAILED|^ERROR" $O/r4b_pytest.log | head -20 | cut -c1-300
grep -E "rel-L2|zero fn|thresholded" $O/r4b_pytest.log | cut -c1-220 | head -20
bash tools/pmc_r4.sh $O/pmc_r4 sd15_self_n4096_d40_bf16_b2 - sd15_self_n4096_d40_bf16_b16 - sd15_self_n4096_d40_f16_b16 - sd21_self_n9216_d64_b8 - qproj_sd15_n4096_b16 --product-only qproj_sd15_n4096_b2 --product-only > $O/r4b_pmc.log 2>&1
grep -E "\||mfma_busy_frac|hbm_side_bytes|GRBM_GUI_ACTIVE|SQ_INSTS_MFMA|SQ_INSTS_VALU  |frac_of_wave" $O/r4b_pmc.log | cut -c1-160 | head -90
