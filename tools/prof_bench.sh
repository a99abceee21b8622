# rocprofv3 kernel-trace of one bench step -> per-kernel stats table (run on the GPU box; writes gpurun_out/)
#   tools/prof_bench.sh [tag] [bench args...]
set -e
export TMPDIR=/tmp
TAG=${1:-r02_bench}; shift || true
mkdir -p gpurun_out
OUT=/tmp/pww_prof_$TAG; rm -rf $OUT
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o run -- python $R/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-reference-ops "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1) || true
DB=$(find $OUT -name "*.db" | head -1)
python tools/rocpd_stats.py "$DB" --top 70 --grid --split-b2b attn_fwd_fold_kernel > gpurun_out/${TAG}_kernel_stats.md
tail -2 gpurun_out/${TAG}_kernel_stats.md
head -3 gpurun_out/${TAG}_kernel_stats.md
