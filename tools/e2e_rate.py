#!/usr/bin/env python
"""End-to-end training rate of the product loop (main.py's inner loop: sampler -> model.train_step) on the synthetic ml-1m
corpus at the headline shape, beside the rate of its parts: the sampler alone, the device step alone."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import synth
from castrec_amd.models import build_model
from castrec_amd.sampler import WarpSampler

B, T, N = 128, 200, int(os.environ.get("N", 600))
W = int(os.environ.get("WORKERS", 1))
corpus = synth.preset("ml-1m")
args = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False, maxlen=T, hidden_units=50, num_blocks=2, num_heads=1,
                             dropout_rate=0.2, l2_emb=0.0, lr=1e-3, num_context_blocks=2, batch_size=B, input_context=False, max_norm=5.0)
model = build_model("cast_1", corpus.usernum, corpus.itemnum, 5, args)
smp = WarpSampler(args, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T, n_workers=W)
t0 = time.perf_counter()
bs = [smp.next_batch() for _ in range(200)]
dt = time.perf_counter() - t0
print("sampler alone (%d worker): %.0f seq/s (%.3f ms per batch)" % (W, 200 * B / dt, dt / 200 * 1e3))
for i in range(20):
    u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
    model.train_step(u, seq, pos, neg, ts, hrs, dys, fetch=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
    out = model.train_step(u, seq, pos, neg, ts, hrs, dys, fetch=(i == N - 1))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("product loop: %.0f seq/s (%.3f ms per step), loss %.4f" % (N * B / dt, dt / N * 1e3, out[1]))
def nb():
    u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
    return u, seq, pos, neg, ts, hrs, dys
if os.environ.get("FEED", "1") == "1":
    model.feed(*nb())
    for i in range(20):
        model.feed(*nb()); model.train_fed(fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        model.feed(*nb())
        out = model.train_fed(fetch=(i == N - 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("product loop, batches fed one ahead (main.py): %.0f seq/s (%.3f ms per step), loss %.4f" % (N * B / dt, dt / N * 1e3, out[1]))
    model.train_fed(fetch=False)
    torch.cuda.synchronize()
    # host time of the loop's three calls (the device runs beside them)
    tn = tf = tt = 0.0
    model.feed(*nb())
    for i in range(300):
        t0 = time.perf_counter(); b = nb(); t1 = time.perf_counter(); model.feed(*b); t2 = time.perf_counter(); model.train_fed(fetch=False); t3 = time.perf_counter()
        tn += t1 - t0; tf += t2 - t1; tt += t3 - t2
    torch.cuda.synchronize()
    model.train_fed(fetch=False)
    print("host time per step: sampler.next_batch %.3f ms, model.feed %.3f ms, model.train_fed %.3f ms" % (tn / 300 * 1e3, tf / 300 * 1e3, tt / 300 * 1e3))
    sys.exit(0)
u, seq, pos, neg, ts, rat, hrs, dys, _ = bs[0]
t0 = time.perf_counter()
for i in range(N):
    model.train_step(u, seq, pos, neg, ts, hrs, dys, fetch=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("train_step on one host batch (no sampler): %.0f seq/s (%.3f ms per step)" % (N * B / dt, dt / N * 1e3))
eng = model._train
t0 = time.perf_counter()
for i in range(N):
    eng.graph.launch() if eng.graph is not None else eng.launch_step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("device step alone (graph %s): %.0f seq/s (%.3f ms per step)" % (eng.graph is not None, N * B / dt, dt / N * 1e3))
smp.close()
