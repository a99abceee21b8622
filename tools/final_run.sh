# final round validation on the GPU box: GPU test-suite, smoke, default bench, batch-8 bench
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log; tail -1 gpurun_out/bench_final.json | cut -c1-400
timeout 300 python bench.py --batch 8 --steps 2 --warmup 1 --cpu-steps 0 --no-reference-ops --no-roofline-pass 2>/dev/null | tail -1 | cut -c1-330 | tee gpurun_out/bench_b8.json
timeout 300 python bench.py --dtype fp16 --steps 3 --warmup 1 --cpu-steps 0 --no-reference-ops --no-roofline-pass 2>/dev/null | tail -1 | cut -c1-330 | tee gpurun_out/bench_fp16.json
