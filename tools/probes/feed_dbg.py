import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E
from test_model_gpu import make_batch

def run(mode, pattern):
    rs = np.random.RandomState(17)
    B, T, D, itemnum = 5, 32, 50, 80
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=6)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b.P.copy_(a.P)
    if mode == "capture_first":
        b.capture()
    b.enable_feed(n_slots=4)
    if mode == "capture_after":
        b.capture()
    batches = [make_batch(rs, B, T, itemnum, 20) for _ in range(8)]
    la, lb = [], []
    for bt in batches:
        a.train_step(*bt); la.append(a.loss_auc()[0])
    it = iter(batches)
    if pattern == "ahead":
        b.feed(*next(it))
        for i in range(7):
            b.feed(*next(it)); b.train_fed(); lb.append(b.loss_auc()[0])
        b.train_fed(); lb.append(b.loss_auc()[0])
    else:
        for i in range(8):
            b.feed(*next(it)); b.train_fed(); lb.append(b.loss_auc()[0])
    print(mode, pattern, "max |loss diff|", max(abs(x - y) for x, y in zip(la, lb)), ["%.4f/%.4f" % (x, y) for x, y in zip(la, lb)])

for mode in ("eager", "capture_first", "capture_after"):
    for pattern in ("ahead", "none"):
        run(mode, pattern)
