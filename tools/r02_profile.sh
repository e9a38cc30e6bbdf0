set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2prof
# kernel-trace + stats of the headline command (no secondary / fp32 reference: per-kernel averages are the headline's)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2prof/stats -- python3 bench.py --steps 20 --warmup 3 --no-secondary --no-fp32-reference > gpurun_out/r2prof/bench_headline.json 2> gpurun_out/r2prof/stats.err
echo stats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r2prof/pmc_default_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-secondary > /dev/null 2> gpurun_out/r2prof/pmc_default_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r2prof/pmc_fused_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-secondary --latent-path fused > /dev/null 2> gpurun_out/r2prof/pmc_fused_$c.err
  echo $c done
done
find gpurun_out/r2prof -name "*.csv" | head -30
du -sh gpurun_out/r2prof
