#!/usr/bin/env python3
"""Twelve headline forwards (BinarySAE 512 -> 32768, 65536 rows, two alternating batches) for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hl -- python3 tools/prof_headline_kernels.py
   python3 tools/prof_headline_kernels.py --report gpurun_out/prof_hl     # newest *_kernel_stats.csv, the library's kernels"""
import csv
import glob
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    files = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])):
        if "qsae" in r["Name"] and int(r["Calls"]) >= 12:
            print(f"{r['Name'][:64]:64s} {int(r['Calls']):4d} {float(r['AverageNs']) / 1e6:8.4f} ms")
    sys.exit(0)

import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
xs = [torch.randn(65536, bench.D, device=dev) for _ in range(2)]
model.decoder.packed()
for i in range(12):
    out = model(xs[i % 2])
torch.cuda.synchronize()
