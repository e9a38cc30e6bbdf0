"""The N > 1 path of bench.py on real kernels without an N-GPU node: two ranks launched exactly as the driver launches
them (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), sharing the one card of the test box
(QSAE_BENCH_BACKEND=gloo: the three scalar reductions go over gloo; the kernels, the row sharding and the max-over-ranks
timing are the production code).  The reduced recon-MSE must equal the MSE this process computes over the two rank
shards.  No 1 -> 8 GPU curve is measured here."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_card_reduce_the_right_mse():
    sys.path.insert(0, str(ROOT))
    import bench
    from quantizedsae_amd import ops
    rows, steps = 4096, 2
    env = dict(os.environ, QSAE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--rows", str(rows), "--steps", str(steps),
           "--warmup", "1", "--no-cpu-baseline", "--no-fp32-reference", "--no-secondary"]
    # children of this process (which has initialised the GPU): started, never exec'ed into
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE JSON line
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == steps and doc["scaling"] == "weak"
    assert doc["config"]["rows_per_gpu"] == rows and doc["config"]["latent_path"] == "prefilter"
    assert doc["value"] == pytest.approx(2 * rows * steps / (doc["ms_per_step"] * steps * 1e-3), rel=1e-9)
    assert doc["cpu_baseline"] is None and doc["roofline"]["avg_kernel_ms"] > 0
    assert doc["input_batches"] == 4 and doc["forward_blocking"]["recon_mse"] == pytest.approx(doc["recon_mse"], rel=1e-12)
    # the same shards, here: rank r draws its batches from torch.Generator(seed 1000 + r), step i takes batch i mod 4 (bench.py)
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    sq = torch.zeros((), dtype=torch.float64, device=dev)
    for rank in range(2):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + rank)
        for _ in range(steps):
            x = torch.randn((rows, bench.D), device=dev, generator=g)
            _lat, rec, _ = model(x)
            ops.sq_err_sum(rec, x, sq)
    want = float(sq) / (2 * steps * rows * bench.D)
    assert doc["recon_mse"] == pytest.approx(want, rel=1e-9)
