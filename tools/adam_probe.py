#!/usr/bin/env python
"""cr_adam_step alone at the headline shape: HIP-event time per launch as built, with one slab, and the sizes behind it."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E, lib as L, synth
from castrec_amd.sampler import WarpSampler

B, T = 128, 200
corpus = synth.preset("ml-1m")
sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T)
u, seq, pos, neg, ts_, rat, hrs, dys, _ = smp.next_batch()
smp.close()
hp = E.Hyper(maxlen=T, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=200, lr=1e-3)
eng = E.Engine("cast_1", corpus.usernum, corpus.itemnum, hp, B, training=True)
eng.set_batch(seq, pos, neg, ts_, hrs, dys)
for _ in range(3):
    eng.launch_step()
torch.cuda.synchronize()
ad = eng._adam[2][0]._obj
lay = eng.layout
print("n_table", lay.n_table, "n_dense", lay.n_dense, "n_slabs", ad.n_slabs, "slab counts", None if eng.slab_counts is None else eng.slab_counts.cpu().numpy().tolist())
s = torch.cuda.current_stream().cuda_stream
big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")


def timed(label, reps=40, dirty=False):
    ts = []
    for _ in range(reps):
        if dirty:
            big.add_(1.0)                                  # 512 MB through the caches: the slabs come from HBM
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng._run([eng._adam], s)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("%-40s median %.1f us  min %.1f" % (label, np.median(ts), np.min(ts)))


timed("as built (slabs warm in the caches)")
timed("as built, caches flushed before", dirty=True)
ns, cnt = ad.n_slabs, ad.slab_counts
ad.n_slabs = 1
timed("one slab")
timed("one slab, caches flushed", dirty=True)
ad.n_slabs = ns
nd = ad.n_dense
ad.n_dense = 256
timed("256 dense parameters (table section + launch)")
ad.n_dense = nd
