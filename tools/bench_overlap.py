"""Experiment: zero-fill of the dense latent on a side stream, concurrent with the prefilter pipeline
(instead of fused into the sweep epilogue)."""
import json, sys, torch
sys.path.insert(0, ".")
from quantizedsae_amd import ops, synthetic

B, D, H, k = 65536, 512, 32768, 65
dev = torch.device("cuda:0")
p = synthetic.binary_sae_params(1, D, H, 4)
W = torch.from_numpy(p["encoder.0.weight"]).to(dev)
b = torch.from_numpy(p["encoder.0.bias"]).to(dev)
x = torch.from_numpy(synthetic.activations(2, B, D)).to(dev)
Wq, meta = ops.prefilter_pack_w(W, b)
dense = torch.empty((B, H), device=dev)
side = torch.cuda.Stream()

def fused():
    ops.encode_topk_prefilter(x, W, b, Wq, meta, k)     # allocates dense from the caching allocator (free after warm-up)

def overlapped():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dense.zero_()
    ops.encode_topk_prefilter(x, W, b, Wq, meta, k, want_dense=False)
    main.wait_stream(side)

def nodense():
    ops.encode_topk_prefilter(x, W, b, Wq, meta, k, want_dense=False)

def fill_only():
    dense.zero_()

for name, fn in (("fused", fused), ("overlapped", overlapped), ("nodense", nodense), ("fill_only", fill_only)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({name: round(e0.elapsed_time(e1) / 5, 3)}), flush=True)
