"""Plain bf16, T 200 / D 50 / B 3: is the block backward's dW2 reproducible, and where do two compilations differ?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import castrec_amd.engine as E
from test_model_gpu import make_batch

T, D, B, H = 200, 50, 3, 1
prec = "bf16"
rs = np.random.RandomState(5)
hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, seed=13)
engs = {}
for name, env in (("const", None), ("runtime", "CASTREC_B1_NO_NKT"), ("tile", "CASTREC_NO_STACK_BWD")):
    if env:
        os.environ[env] = "1"
    engs[name] = E.Engine("cast_1", 9, 45, hp, B, training=True, n_slabs=7, attn_precision=prec)
    if env:
        del os.environ[env]
a = engs["const"]
a.P.add_(0.05 * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
batch = make_batch(rs, B, T, 45, 9)
G = {}
for n, e in engs.items():
    e.P.copy_(a.P)
    e.set_batch(*batch)
    runs = []
    for rep in range(3):
        e.launch_step(apply=False)
        torch.cuda.synchronize()
        runs.append({k: v.clone() for k, v in e.grads().items()})
    G[n] = runs[0]
    print(n, "reproducible over 3 runs:", all(torch.equal(runs[0][k], r[k]) for r in runs[1:] for k in r))
for k in ("ctx_time.1.w2", "trunk.0.w2", "ctx_time.1.w1"):
    c, r, t = G["const"][k], G["runtime"][k], G["tile"][k]
    print(k, "shape", tuple(c.shape), "|tile| max %.4f" % float(t.abs().max()))
    for nm, x, y in (("const-runtime", c, r), ("const-tile", c, t), ("runtime-tile", r, t)):
        dd = (x - y).abs()
        rows = dd.max(1).values
        top = torch.topk(rows, 6)
        print("   %-14s max %.4e  rows with the largest difference: %s  (row maxima %s)" % (nm, float(dd.max()), top.indices.tolist(), ["%.1e" % v for v in top.values.tolist()]))
        cols = dd.max(0).values
        topc = torch.topk(cols, 6)
        print("   %-14s                cols: %s (%s)" % ("", topc.indices.tolist(), ["%.1e" % v for v in topc.values.tolist()]))
# the hidden activations the backward reads: how large are they?
for n, e in engs.items():
    pass
