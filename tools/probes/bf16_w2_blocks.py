"""Plain bf16, T 200 / D 50, live window [64, 96): the dW2 error of the block backward by 16 x 16 output block (it = in-column tile = wave >> 1,
jt = out-column tile: waves with wave & 1 = 0 own jt 0, 1; the others 2, 3)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import castrec_amd.engine as E
from test_model_gpu import make_batch

T, D, B = 200, 50, int(os.environ.get("PB", 3))
hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=9, seed=13)
a = E.Engine("sasrec", 9, 45, hp, B, training=True, n_slabs=7, attn_precision="bf16")
os.environ["CASTREC_NO_STACK_BWD"] = "1"
b = E.Engine("sasrec", 9, 45, hp, B, training=True, n_slabs=7, attn_precision="bf16")
del os.environ["CASTREC_NO_STACK_BWD"]
a.P.add_(0.05 * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
b.P.copy_(a.P)
for lo, hi in ((64, 96), (64, 80), (80, 96)):
    rs = np.random.RandomState(5)
    batch = [x.copy() for x in make_batch(rs, B, T, 45, 9)]
    for x in batch[:3]:
        x[:, :lo] = 0; x[:, hi:] = 0
        x[:, lo:hi] = np.maximum(x[:, lo:hi], 1)
    g = []
    for e in (a, b):
        for rep in range(2):
            e.set_batch(*batch)
            e.set_step(1)
            e.Gflat.zero_()
            e.launch_step(apply=False)
            torch.cuda.synchronize()
            g.append({k: v.clone() for k, v in e.grads().items()})
    for k in ("trunk.0.w2", "trunk.1.w2"):
        x, x2, y = g[0][k], g[1][k], g[2][k]
        print("live [%d, %d) %s: own max %.4f, two runs of the block backward equal: %s" % (lo, hi, k, float(y.abs().max()), torch.equal(x, x2)))
        e = torch.zeros(64, 64); e[:50, :50] = (x - y).abs().cpu() / float(y.abs().max())
        blk = e.reshape(4, 16, 4, 16).amax(dim=(1, 3))
        for it in range(4):
            print("      it %d: " % it + "  ".join("%.3f" % float(v) for v in blk[it]))
        r = e.reshape(4, 4, 4, 64).amax(dim=(0, 1, 3))
        print("      by r = row & 3: " + "  ".join("%.3f" % float(v) for v in r))
