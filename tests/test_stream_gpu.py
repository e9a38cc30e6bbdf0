"""Config 5 end to end on one card: a [ctx, tok, 512] fp16 chunk in the reference's format (data/dataset.py:16-33) ->
HiddenStatesTorchDataset -> row shards -> 65536-row device batches with the host -> device copies in the loop -> two batches
in flight -> device-side MSE (the reference's loop: scripts/analysis/dynamic_analyze.py:9-24, dynamic_analysis.py:345-362).
Run (a) in this process as one rank and (b) as two fresh child ranks sharing the card (gloo for the two scalar reductions,
as in test_multigpu_gpu.py): the reduced MSE of (b) equals (a)'s over the same rows.  Two stand-ins for the pythia residuals,
which do not exist offline: bell-shaped rows, and a heavy-tailed stream (outlier dimensions, 10x row-scale spread, encoder
biases, heavy encoder rows)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle
from quantizedsae_amd import data, sharding
from quantizedsae_amd.stream import stream_reconstruction_error

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
DEV = "cuda:0"


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["gauss", "heavy"])
def test_chunk_stream_one_rank_equals_two_ranks(tmp_path, mode):
    sys.path.insert(0, str(ROOT / "tools"))
    import run_chunk_stream as tool
    contexts = 1050                                           # 262500 rows: per rank 65536 + 65536 + 178 (ragged tail)
    chunk = tmp_path / "chunk0.pt"
    tool.make_chunk(chunk, mode, contexts)
    ds = data.HiddenStatesTorchDataset(chunk)
    assert len(ds) == contexts * 250 and ds.data.dtype == torch.float16
    model = tool.build_model(mode, DEV)
    one = stream_reconstruction_error(model, [ds], DEV)
    blocking = stream_reconstruction_error(model, [ds], DEV, in_flight=1)
    assert one["rows"] == len(ds) and one["batches"] == 5 and blocking["recon_mse"] == pytest.approx(one["recon_mse"], rel=1e-12)
    assert max(one["flagged_rows_per_batch"]) <= 64           # the candidate lists serve (nearly) every row of either stream
    # the metric itself: 96 rows of the stream against the CPU oracle (bit-exact reconstruction -> equal squared error)
    sel = np.arange(0, len(ds), len(ds) // 96)[:96]
    xs = ds.rows()[sel].numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    want = oracle.binary_forward(xs, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"], sd["decoder.bias"],
                                 n_bits=4, gamma=4.0, k=model.top_k)
    _idx, _val, rec = model.forward_compact(torch.from_numpy(xs).to(DEV))
    assert np.array_equal(rec.cpu().numpy(), want["reconstruction"])
    # (b) two ranks, fresh processes, same card
    env = dict(os.environ, QSAE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tools" / "run_chunk_stream.py"), "--chunk", str(chunk), "--mode", mode,
           "--passes", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc["world_size"] == 2 and doc["rows_total"] == len(ds)
    s0, e0 = sharding.shard_rows(len(ds), 2, 0)
    assert doc["rank0_rows"] == e0 - s0 and doc["rank0_batches"] == 3
    assert doc["recon_mse"] == pytest.approx(one["recon_mse"], rel=1e-9)


def test_chunk_stream_serves_every_model_class(tmp_path):
    """stream_reconstruction_error over a small chunk with each module class == the plain loop over the same batches
    (forward, reconstruction as the reference's adapters pick it, squared error summed in fp64)."""
    from quantizedsae_amd import (BaselineSparseAutoencoder, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,
                                  TernarySparseAutoencoder, ops)
    sys.path.insert(0, str(ROOT / "tools"))
    import run_chunk_stream as tool
    chunk = tmp_path / "small.pt"
    tool.make_chunk(chunk, "gauss", 20)                    # 5000 rows
    ds = data.HiddenStatesTorchDataset(chunk)
    tern = TernarySparseAutoencoder(512, 4096)
    with torch.no_grad():
        tern.decoder.weight.normal_(0, 0.5)
    mat = QuantizedMatryoshkaSAE(512, 8192, top_k=32, abs_range=4, n_bits=4)
    with torch.no_grad():
        mat.encoder[0].bias.fill_(-0.8)
    models = [BaselineSparseAutoencoder(512, 8192), tern, mat, ResidualQuantizedSAE(512, 4096, top_k=32, abs_range=1.5, n_bits=4)]
    for model in models:
        model = model.to(DEV).eval()
        got = stream_reconstruction_error(model, [ds], DEV, batch_rows=2048)
        sq = torch.zeros((), dtype=torch.float64, device=DEV)
        for xb in ds.iter_batches(2048, DEV):
            outs = model(xb)
            rec = outs[-1][-1] if isinstance(outs[-1], (list, tuple)) else outs[-1]
            ops.sq_err_sum(rec, xb, sq)
        want = float(sq) / (len(ds) * 512)
        assert got["rows"] == len(ds) and got["batches"] == 3
        assert got["recon_mse"] == pytest.approx(want, rel=1e-6), type(model).__name__
