import sys, numpy as np, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
from quantizedsae_amd import ops, synthetic as S
DEV='cuda:0'
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for B,Dm,H in [(300,512,4096),(1024,512,32768),(257,128,1000)]:
    W = S.xavier_uniform(400, H, Dm, stream=1); b = S.normal(400, (H,), stream=3, std=0.1); x = S.activations(401, B, Dm)
    Wc, meta2 = ops.emu_pack_w(dev(W))
    got = ops.encode_dense_emu(dev(x), Wc, meta2, dev(b), 0).cpu().numpy()
    f32 = ops.encode_dense(dev(x), dev(W), dev(b), 0).cpu().numpy()
    pre = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    scale = np.abs(pre).max(axis=1, keepdims=True)
    de, df = (got - pre), (f32 - pre)
    print(B,Dm,H, 'emu: max rel', (np.abs(de)/scale).max(), 'mean signed', (de/scale).mean(), 'rms', np.sqrt(((de/scale)**2).mean()),
          '| f32: max rel', (np.abs(df)/scale).max(), 'rms', np.sqrt(((df/scale)**2).mean()))
