"""Plain bf16: dW2 of the one-launch block backward against the tile kernels over shapes; reproducibility by key."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import castrec_amd.engine as E
from test_model_gpu import make_batch

def run(T, D, B, prec="bf16", model="cast_1", scale=0.05):
    rs = np.random.RandomState(5)
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=9, seed=13)
    a = E.Engine(model, 9, 45, hp, B, training=True, n_slabs=7, attn_precision=prec)
    os.environ["CASTREC_NO_STACK_BWD"] = "1"
    b = E.Engine(model, 9, 45, hp, B, training=True, n_slabs=7, attn_precision=prec)
    del os.environ["CASTREC_NO_STACK_BWD"]
    a.P.add_(scale * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    b.P.copy_(a.P)
    batch = make_batch(rs, B, T, 45, 9)
    out = {}
    for n, e in (("a", a), ("b", b)):
        e.set_batch(*batch)
        reps = []
        for r in range(3):
            e.Gflat.zero_()
            e.launch_step(apply=False)
            torch.cuda.synchronize()
            reps.append({k: v.clone() for k, v in e.grads().items()})
        out[n] = reps
    ga, gb = out["a"][0], out["b"][0]
    bad = [k for k in ga if not all(torch.equal(ga[k], r[k]) for r in out["a"][1:])]
    line = []
    for k in ga:
        if k.endswith(".w2") or k.endswith(".w1") or k.endswith(".b2"):
            line.append("%s %.3f" % (k.replace("ctx_time", "c").replace("trunk", "t"), float((ga[k] - gb[k]).abs().max() / gb[k].abs().max())))
    print("T %3d D %2d B %d %s scale %.2f | not reproducible: %s | (block - tile) / own max: %s" % (T, D, B, prec, scale, bad, "  ".join(line)), flush=True)

for T, D, B in ((200, 50, 3), (40, 50, 6), (112, 50, 3), (128, 50, 3), (200, 48, 3), (200, 56, 3), (200, 40, 3)):
    run(T, D, B)
run(200, 50, 3, scale=0.0)
run(200, 50, 3, scale=0.02)
run(200, 50, 3, model="sasrec")
