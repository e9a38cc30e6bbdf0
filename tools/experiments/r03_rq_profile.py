import sys, torch
sys.path.insert(0, '.')
from quantizedsae_amd import ResidualQuantizedSAE
dev='cuda:0'
m = ResidualQuantizedSAE(512, 32768, top_k=32, abs_range=1.5, n_bits=4).to(dev).eval()
x = torch.randn(32768, 512, device=dev)
for _ in range(6): out = m(x)
torch.cuda.synchronize()
