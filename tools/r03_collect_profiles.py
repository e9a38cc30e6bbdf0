#!/usr/bin/env python3
"""Fold the output of tools/r03_profile.sh (gpurun_out/r3prof) into profiles/: headline bench line + kernel stats CSV, the
full bench line (secondary configurations) + its kernel stats, and profiles/r03_traffic.json (per-launch FETCH_SIZE /
WRITE_SIZE means of the sweep / fill / refinement / split-decoder kernels; FETCH_SIZE doubled per the gfx950 correction of
MI355X_MICROARCH.md)."""
import csv
import glob
import json
import shutil
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = ROOT / "gpurun_out" / "r3prof"
newest = lambda pat: max(glob.glob(str(src / pat), recursive=True), key=lambda f: Path(f).stat().st_mtime)
shutil.copy(newest("stats/**/*kernel_stats.csv"), ROOT / "profiles" / "r03_bench_headline_kernel_stats.csv")
shutil.copy(newest("stats_all/**/*kernel_stats.csv"), ROOT / "profiles" / "r03_bench_all_kernel_stats.csv")
for name in ("bench_headline", "bench_all"):
    line = (src / f"{name}.json").read_text().strip().splitlines()[-1]
    json.loads(line)
    (ROOT / "profiles" / f"r03_{name}.json").write_text(line + "\n")


def means(run, counter):
    f = newest(f"{run}/**/*counter_collection.csv")
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


TAGS = ("sweep_xstat_f16", "fill_zero_co", "refine_topk", "refine_select", "refine_slice_chain", "refine_rank",
        "split_dec_bf16_kernel<0>", "split_dec_bf16_kernel<1>", "gemm_nt_f32_kernel")
raw = {}
for run in ("pmc_default", "pmc_split"):
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in means(f"{run}_{counter}", counter).items():
            for tag in TAGS:
                if tag in k:
                    raw.setdefault(f"{run}:{tag}", {})[counter + "_KiB"] = round(v, 1)
doc = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/r03_profile.sh), mean per launch; "
               "hbm_side_bytes_per_launch = 2 x FETCH_SIZE (gfx950: wide coalesced reads are tallied at half their bytes) + "
               "WRITE_SIZE; Infinity-Cache hits are counted as traffic",
       "kernels": {}}
for key, v in raw.items():
    if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
        v["hbm_side_bytes_per_launch"] = int((2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024)
    doc["kernels"][key] = v
k = doc["kernels"]
if "pmc_default:sweep_xstat_f16" in k and "pmc_default:fill_zero_co" in k:
    sw, fl = k["pmc_default:sweep_xstat_f16"], k["pmc_default:fill_zero_co"]
    k["sweep_xstat_f16+fill_zero_co"] = {
        "sweep_FETCH_SIZE_KiB": sw["FETCH_SIZE_KiB"], "sweep_WRITE_SIZE_KiB": sw["WRITE_SIZE_KiB"],
        "fill_WRITE_SIZE_KiB": fl["WRITE_SIZE_KiB"],
        "hbm_side_bytes_per_launch": int((2 * sw["FETCH_SIZE_KiB"] + sw["WRITE_SIZE_KiB"] + fl["WRITE_SIZE_KiB"]) * 1024)}
(ROOT / "profiles" / "r03_traffic.json").write_text(json.dumps(doc, indent=1) + "\n")
print(json.dumps(doc, indent=1))
