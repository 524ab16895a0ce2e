"""Does the block backward with the headline's tile count / length as constants (DS = D | nkt << 8 | T << 16) produce the same bits as the
run-time form (CASTREC_B1_NO_NKT=1)?  And how far is each from the tile kernels (CASTREC_NO_STACK_BWD) in plain bf16?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import castrec_amd.engine as E
from test_model_gpu import make_batch

T, D, B, H = 200, 50, 3, 1
for prec in ("bf16", "bf16x3"):
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
        e.launch_step(apply=False)
        torch.cuda.synchronize()
        G[n] = e.grads()
    gmax = max(float(v.abs().max()) for v in G["tile"].values())
    worst = {}
    for k in G["tile"]:
        worst[k] = (float((G["const"][k] - G["runtime"][k]).abs().max()), float((G["const"][k] - G["tile"][k]).abs().max()) / gmax,
                    float((G["runtime"][k] - G["tile"][k]).abs().max()) / gmax)
    print(prec, "gmax", gmax, "const==runtime bitwise:", all(torch.equal(G["const"][k], G["runtime"][k]) for k in G["tile"]))
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1][1])[:6]:
        print("   %-22s const-runtime %.3e   const-tile %.4f   runtime-tile %.4f of gmax" % (k, *v))
