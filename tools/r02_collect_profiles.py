#!/usr/bin/env python3
"""Fold the output of tools/r02_profile.sh (gpurun_out/r2prof) into profiles/: headline bench line, kernel stats CSV, and
the per-launch FETCH_SIZE / WRITE_SIZE means of the sweep / fill / refinement kernels (printed; the judged numbers live in
profiles/r02_traffic.json, whose refinement and sweep entries are updated in place)."""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = ROOT / "gpurun_out" / "r2prof"
newest = lambda pat: max(glob.glob(str(src / pat), recursive=True), key=lambda f: Path(f).stat().st_mtime)
shutil.copy(newest("stats/**/*kernel_stats.csv"), ROOT / "profiles" / "r02_bench_headline_kernel_stats.csv")
line = (src / "bench_headline.json").read_text().strip().splitlines()[-1]
json.loads(line)
(ROOT / "profiles" / "r02_bench_headline.json").write_text(line + "\n")


def means(run, counter):
    f = newest(f"{run}/**/*counter_collection.csv")
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = {}
for run, counter in (("pmc_default_FETCH_SIZE", "FETCH_SIZE"), ("pmc_default_WRITE_SIZE", "WRITE_SIZE"),
                     ("pmc_fused_FETCH_SIZE", "FETCH_SIZE"), ("pmc_fused_WRITE_SIZE", "WRITE_SIZE")):
    for k, v in means(run, counter).items():
        for tag in ("sweep_xstat_f16", "fill_zero_co", "refine_topk", "gemm_nt_f32_dma_kernel<qsae::EpiFilter"):
            if tag in k:
                out[f"{run.split('_')[1]}:{tag}:{counter}_KiB"] = round(v, 1)
print(json.dumps(out, indent=1))
t = json.loads((ROOT / "profiles" / "r02_traffic.json").read_text())
k = t["kernels"]
sw = k["sweep_xstat_f16+fill_zero_co"]
sw["sweep_FETCH_SIZE_KiB"] = out["default:sweep_xstat_f16:FETCH_SIZE_KiB"]
sw["sweep_WRITE_SIZE_KiB"] = out["default:sweep_xstat_f16:WRITE_SIZE_KiB"]
sw["fill_WRITE_SIZE_KiB"] = out["default:fill_zero_co:WRITE_SIZE_KiB"]
sw["hbm_side_bytes_per_launch"] = int((2 * sw["sweep_FETCH_SIZE_KiB"] + sw["sweep_WRITE_SIZE_KiB"] + sw["fill_WRITE_SIZE_KiB"]) * 1024)
rf = k["refine_topk_kernel"]
rf["FETCH_SIZE_KiB"] = out["default:refine_topk:FETCH_SIZE_KiB"]
rf["WRITE_SIZE_KiB"] = out["default:refine_topk:WRITE_SIZE_KiB"]
rf["hbm_side_bytes_per_launch"] = int((2 * rf["FETCH_SIZE_KiB"] + rf["WRITE_SIZE_KiB"]) * 1024)
(ROOT / "profiles" / "r02_traffic.json").write_text(json.dumps(t, indent=1) + "\n")
print("sweep+fill", sw["hbm_side_bytes_per_launch"], "refine", rf["hbm_side_bytes_per_launch"])
