# Round-3 profile collection (GPU box): bash tools/r03_profile.sh   -> gpurun_out/r3prof/, folded by tools/r03_collect_profiles.py
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3prof
# 1. kernel-trace + stats of the headline command (no secondary / fp32 reference: per-kernel averages are the headline's)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/stats -- python3 bench.py --steps 20 --warmup 3 --no-secondary --no-fp32-reference --sustained-seconds 0 > gpurun_out/r3prof/bench_headline.json 2> gpurun_out/r3prof/stats.err
echo stats done
# 2. the same with the secondary configurations (split decoders, table decode, bits path)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/stats_all -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3prof/bench_all.json 2> gpurun_out/r3prof/stats_all.err
echo stats_all done
# 3. HBM-side traffic, one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3prof/pmc_default_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-secondary --sustained-seconds 0 > /dev/null 2> gpurun_out/r3prof/pmc_default_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3prof/pmc_split_$c -- python3 tools/bench_split_dec.py > /dev/null 2> gpurun_out/r3prof/pmc_split_$c.err
  echo $c done
done
find gpurun_out/r3prof -name "*.csv" | head -40
du -sh gpurun_out/r3prof
