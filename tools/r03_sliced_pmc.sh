# SQ / TCC counters of the three refinement launches (select, slice-major chains, rank) at the headline shape, one rocprofv3 pass
# per counter group:  bash tools/r03_sliced_pmc.sh  (GPU box) -> gpurun_out/r3sl/summary.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r3sl
mkdir -p $out
run() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $out/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-secondary --pipeline 1 --sustained-seconds 0 > /dev/null 2> $out/$1.err
  echo "$1 done"
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run tcp1 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
run clk "GRBM_GUI_ACTIVE"
python3 - > $out/summary.txt <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r3sl/*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:44]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', d)
    for k, cs in agg.items():
        if not any(s in k for s in ('refine', 'sweep_xstat', 'fill_zero')):
            continue
        print(' ', k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
cat $out/summary.txt
