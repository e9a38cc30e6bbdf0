# clock and pipe occupancy of the headline kernels: rocprofv3 --pmc passes over bench.py (GPU box; kernels are serialised under
# counter collection, so the sweep is seen WITHOUT the zero-fill kernel beside it)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3head
for c in GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r3head/$c -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-secondary --sustained-seconds 0 > /dev/null 2> gpurun_out/r3head/$c.err || echo "$c failed"
  echo $c done
done
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(dict)
tags = {"sweep_xstat_f16": "sweep", "refine_topk": "refine", "fill_zero_co": "fill"}
for d in glob.glob("gpurun_out/r3head/*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for t, k in tags.items():
                if t in r["Kernel_Name"]:
                    out[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for t, k in tags.items():
                if t in r["Kernel_Name"]:
                    out[k].setdefault("duration_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in out.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    if "GRBM_GUI_ACTIVE" in m:
        m["clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8 / m["duration_ns"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["clock_GHz"] * m["duration_ns"])
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        m["wait_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    print(k, m)
PY
