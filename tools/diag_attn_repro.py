"""Bitwise reproducibility of the bf16-MFMA attention kernels over many launches of the same inputs (forward + backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import ops as O
B, T, H, d = int(os.environ.get("B", 128)), int(os.environ.get("T", 200)), int(os.environ.get("H", 4)), int(os.environ.get("DH", 32))
N = int(os.environ.get("NRUNS", 2000))
M, D = B * T, H * d
rs = np.random.RandomState(0)
f = lambda *s: torch.randn(*s, device="cuda")
Q, K, V, R, dO = f(M, D), f(M, D), f(M, D), f(M, D), f(M, D)
out, dQ, dK, dV = f(M, D), f(M, D), f(M, D), f(M, D)
lens = np.clip(rs.lognormal(4.6, 0.9, B), 3, T).astype(int)
ids = np.zeros((B, T), np.int32)
for b in range(B): ids[b, T - lens[b]:] = 1
idd = torch.tensor(ids.reshape(-1), device="cuda")
kv = torch.tensor((ids != 0).astype(np.float32).reshape(-1), device="cuda"); qv = kv.clone()
state = torch.zeros(16, device="cuda")
drop = O.Drop(0.2, 1, state)
row_stats = torch.empty(H * B * T * 4, device="cuda")
desc = O.attn_desc(Q, K, V, D, kv, qv, R, D, out, D, B, T, H, d, rng=drop.rng(3), dead_ids=idd, row_stats=row_stats, precision=1)
stats = torch.empty(H * B * T * 4, device="cuda")
delta = f(H * M) if T <= 256 else None
ref = None; bad = 0
for it in range(N):
    O.attn_fwd(desc)
    O.attn_bwd(desc, dO, D, dQ, dK, dV, D, stats, delta=delta)
    if it % 50 == 49 or it == 0:
        torch.cuda.synchronize()
    cur = [out.clone(), dQ.clone(), dK.clone(), dV.clone()]
    if ref is None:
        ref = cur
    else:
        for nme, a, b in zip(("out", "dQ", "dK", "dV"), cur, ref):
            if not torch.equal(a, b):
                bad += 1
                print("run", it, nme, "differs: max", float((a - b).abs().max()))
print("B %d T %d H %d d %d: %d launches, %d differing tensors" % (B, T, H, d, N, bad))
