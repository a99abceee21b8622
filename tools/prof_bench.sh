# rocprofv3 kernel-trace of one bench step -> per-kernel stats table (run on the GPU box; writes gpurun_out/)
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.log || true
tail -1 gpurun_out/bench_default.json | head -c 6000
OUT=/tmp/pww_prof; rm -rf $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o run -- python bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-roofline-pass --no-reference-ops > gpurun_out/prof_bench.log 2>&1 || true
DB=$(find $OUT -name "*.db" | head -1)
python tools/rocpd_stats.py "$DB" --top 60 --grid > gpurun_out/bench_v6_kernel_stats.md
head -5 gpurun_out/bench_v6_kernel_stats.md
