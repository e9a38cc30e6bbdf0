"""torch.ops.qsae.*: the dispatcher surface of the kernels (quantizedsae_amd/torch_ops.py).  CPU part: every op has a
schema and a fake implementation, and the module classes' forwards trace -- fullgraph, no graph break -- into one qsae node
each (the reference's forwards are plain ATen sequences, sae/binary.py:91-103 etc., and trace as such).  Nothing is computed
here: there is no CPU kernel, the compiled call ends in the package's "no CPU fallback" error after the graph was captured."""
import copy
import pickle

import pytest
import torch

import quantizedsae_amd  # noqa: F401  (registers the ops)
from quantizedsae_amd import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,
                              TernarySparseAutoencoder, torch_ops)

KERNEL_OPS = ["kperm_rows", "encode_dense", "encode_bits", "encode_bits_prefilter", "topk_rows", "encode_topk",
              "encode_topk_latent", "prefilter_pack_w", "encode_topk_prefilter", "binary_forward_prefilter",
              "table_forward_prefilter", "densify", "pack_binary", "unpack_binary", "binary_soft_table",
              "decode_binary_sparse", "decode_table_sparse", "pack_ternary", "decode_ternary_dense", "pack_matryoshka",
              "pack_matryoshka_rows", "decode_matryoshka", "pack_bits_gt", "residual_update", "threshold_ge",
              "scale_bias_rows", "sq_err_sum", "activation_counts", "activation_counts_bits", "coactivation_sparse",
              "quantize_bits"]
MODEL_OPS = ["binary_sae_forward", "baseline_sae_forward", "ternary_sae_forward", "levels_sae_forward"]


def test_every_op_is_registered_with_a_schema():
    for name in KERNEL_OPS + MODEL_OPS:
        op = getattr(torch.ops.qsae, name).default
        assert str(op._schema).startswith(f"qsae::{name}(")
    s = str(torch.ops.qsae.topk_rows.default._schema)
    assert "Tensor(a0!) latent" in s                     # the in-place top-k declares what it mutates
    assert "Tensor(a2!) acc" in str(torch.ops.qsae.sq_err_sum.default._schema)


def test_fake_implementations_give_the_shapes_of_the_kernels():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        B, D, H, k = 96, 512, 4096, 8
        x, W, b = torch.empty(B, D), torch.empty(H, D), torch.empty(H)
        Wq, meta = torch.ops.qsae.prefilter_pack_w(W, b)
        assert Wq.shape == (H, D) and Wq.dtype == torch.float16 and meta.shape == (4,)
        packed, pol, gap = torch.ops.qsae.pack_binary(torch.empty(H, D * 4), D, 4)
        assert packed.shape == (H, 256) and packed.dtype == torch.uint8 and pol.dtype == torch.float64 and gap.shape == ()
        idx, val, dense, recon, flagged = torch.ops.qsae.binary_forward_prefilter(x, W, b, Wq, meta, k, packed, 4, 0.5, None, True, 0)
        assert idx.shape == (B, k) and idx.dtype == torch.int32 and val.shape == (B, k) and dense.shape == (B, H)
        assert recon.shape == (B, D) and flagged.shape == () and flagged.device.type == "cpu"
        assert torch.ops.qsae.table_forward_prefilter(x, W, b, Wq, meta, k, torch.empty(H, D), 1.0, None, False, 0)[2].shape == (0, H)
        assert torch.ops.qsae.encode_bits(x, W, b).shape == (B, H // 32)
        levels, counts = torch.ops.qsae.decode_matryoshka(torch.empty(B, H // 32, dtype=torch.int32), H, D, 4,
                                                          torch.empty(D, H // 16, dtype=torch.int32), torch.empty(H), None,
                                                          True, None, False)
        assert levels.shape == (4, B, D) and counts.shape == (4,) and counts.dtype == torch.int64
        assert torch.ops.qsae.decode_ternary_dense(torch.empty(B, H), torch.empty(D, H // 16, dtype=torch.int32), D).shape == (B, D)
        assert torch.ops.qsae.quantize_bits(x, 4, 2.0, True).shape == (B, D * 4)


MODELS = [
    ("binary_sae_forward", lambda: BinarySAE(64, 1024, gamma=4.0, n_bits=4)),
    ("baseline_sae_forward", lambda: BaselineSparseAutoencoder(64, 1024)),
    ("ternary_sae_forward", lambda: TernarySparseAutoencoder(64, 1024)),
    ("levels_sae_forward", lambda: QuantizedMatryoshkaSAE(64, 1024, top_k=8, abs_range=4, n_bits=4)),
    ("levels_sae_forward", lambda: ResidualQuantizedSAE(64, 1024, top_k=8, abs_range=4, n_bits=4)),
]


@pytest.mark.parametrize("op_name,make", MODELS, ids=[m[1]().__class__.__name__ for m in MODELS])
def test_forward_traces_into_one_node_without_graph_break(op_name, make):
    model = make().eval()
    graphs = []

    def backend(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    with pytest.raises(RuntimeError, match="no CPU fallback"):      # traced fully, then run: CPU tensors have no kernel
        torch.compile(model, fullgraph=True, backend=backend)(torch.randn(8, 64))
    assert len(graphs) == 1
    targets = [str(n.target) for n in graphs[0].graph.nodes if n.op == "call_function" and "qsae" in str(n.target)]
    assert targets == [f"qsae.{op_name}"]


def test_a_copy_of_a_module_gets_a_graph_handle_of_its_own():
    m = BinarySAE(64, 1024, gamma=4.0, n_bits=4)
    c = copy.deepcopy(m)
    u = pickle.loads(pickle.dumps(m))
    handles = {m._qsae_handle, c._qsae_handle, u._qsae_handle}
    assert len(handles) == 3
    for mod in (m, c, u):
        assert torch_ops._module(mod._qsae_handle) is mod
    assert torch.equal(c.encoder[0].weight, m.encoder[0].weight) and torch.equal(u.decoder.weight, m.decoder.weight)
