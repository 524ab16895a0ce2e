"""Two engines, the same batches, N optimiser steps each (the headline shape, every arithmetic): the parameters must hold the same bits at the end,
and a third run of the first engine's last step from a snapshot must reproduce them.  A soak for rare schedule-dependent defects
(profiles/r04_flake, DESIGN.md section 4 "a mixed-shape accumulate")."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E, synth
from castrec_amd.sampler import WarpSampler

N = int(os.environ.get("STEPS", 400))
B, T = 128, 200
corpus = synth.preset("ml-1m")
sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T)
batches = []
for _ in range(8):
    u, seq, pos, neg, ts_, rat, hrs, dys, _ = smp.next_batch()
    batches.append((seq, pos, neg, ts_, hrs, dys))
smp.close()
for prec in ("bf16x3", "bf16", "f32"):
    hp = E.Hyper(maxlen=T, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=200, lr=1e-3, seed=3)
    engs = [E.Engine("cast_1", corpus.usernum, corpus.itemnum, hp, B, training=True, attn_precision=prec) for _ in range(2)]
    engs[1].P.copy_(engs[0].P)
    for e in engs:
        for i in range(N):
            e.train_step(*batches[i % len(batches)])
    torch.cuda.synchronize()
    same = torch.equal(engs[0].P, engs[1].P) and torch.equal(engs[0].Mom, engs[1].Mom) and torch.equal(engs[0].Vel, engs[1].Vel)
    print("%-7s %d steps x 2 engines: parameters and moments bitwise equal: %s   (finite: %s, loss %.5f)" %
          (prec, N, same, bool(torch.isfinite(engs[0].P).all()), engs[0].loss_auc()[0]), flush=True)
