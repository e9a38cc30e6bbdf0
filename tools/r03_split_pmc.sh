# clock and matrix-pipe occupancy of the split decoders: rocprofv3 --pmc passes over tools/bench_split_dec.py (GPU box)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3split
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3split/$c -- python3 tools/bench_split_dec.py > /dev/null 2> gpurun_out/r3split/$c.err || echo "$c failed"
  echo $c done
done
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(dict)
for d in glob.glob("gpurun_out/r3split/*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "split_dec_bf16_kernel" in r["Kernel_Name"]:
                k = "ternary" if "<0" in r["Kernel_Name"] else "matryoshka"
                out[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "split_dec_bf16_kernel" in r["Kernel_Name"]:
                k = "ternary" if "<0" in r["Kernel_Name"] else "matryoshka"
                out[k].setdefault("duration_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in out.items():
    print(k, {n: sum(x) / len(x) for n, x in v.items()})
PY
