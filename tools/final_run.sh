# round validation on the GPU box: native harness, GPU test-suite, smoke, default bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
(cd tests/native && ./attn_check --quick 2>&1 | grep -c "^PASS"; ./attn_check --quick 2>&1 | grep -v "^PASS" | tail -4)
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 400 python bench.py --cpu-steps 0 --no-reference-ops > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log; tail -1 gpurun_out/bench_final.json | cut -c1-200
