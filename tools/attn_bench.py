#!/usr/bin/env python3
"""Attention kernels at the headline shape (B=128, T=200, d=50, one head, dropout 0.2, 55 % filled left-padded
sequences): HIP-event time per launch for each precision (0 fp32 MFMA, 1 bf16 hi+lo split, 2 plain bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import castrec_amd  # noqa: F401
from castrec_amd import ops as O

B = int(os.environ.get("B", 128)); T = int(os.environ.get("T", 200)); H = int(os.environ.get("H", 1)); d = int(os.environ.get("DH", 50))
M, C = B * T, H * d
rs = np.random.RandomState(0)
lens = np.clip(rs.lognormal(4.6, 0.9, B), 3, T).astype(int)
ids = np.zeros((B, T), np.int32)
for b in range(B):
    ids[b, T - lens[b]:] = 1
print("fill %.3f" % ids.mean())
f = lambda *s: torch.randn(*s, device="cuda")
Q, K, V, R, dO = f(M, C), f(M, C), f(M, C), f(M, C), f(M, C)
kv = torch.tensor(ids.reshape(-1), dtype=torch.float32, device="cuda"); qv = torch.ones(M, device="cuda")
idd = torch.tensor(ids.reshape(-1), dtype=torch.int32, device="cuda")
state = torch.zeros(16, device="cuda"); state[4:5].view(torch.int32)[0] = 1
drop = O.Drop(0.2, 1, state)


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


flops_f = 2.0 * C * T * (T + 1) * B
for prec in (0, 1, 2):
    out = torch.empty(M, C, device="cuda"); rst = torch.zeros(H * B * T * 4, device="cuda")
    desc = O.attn_desc(Q, K, V, C, kv, qv, R, C, out, C, B, T, H, d, rng=drop.rng(5), dead_ids=idd, row_stats=rst, precision=prec)
    tf = timeit(lambda: O.attn_fwd(desc))
    dQ, dK, dV, part = (torch.empty(M, C, device="cuda") for _ in range(4))
    stats = torch.empty(H * B * T * 4, device="cuda")
    delta = (dO * (out - R)).sum(1).contiguous() if H == 1 else None
    if prec == 0 and H == 1:
        tb = timeit(lambda: O.attn_bwd(desc, dO, C, dQ, dK, dV, C, stats, delta=delta, dQ_part=part))
    else:
        tb = timeit(lambda: O.attn_bwd(desc, dO, C, dQ, dK, dV, C, stats, delta=delta))
    print("prec %d: fwd %6.1f us (%5.1f TF/s algorithmic)   bwd %6.1f us (%5.1f TF/s)" % (prec, tf, flops_f / tf / 1e6, tb, 2 * flops_f / tb / 1e6), flush=True)
