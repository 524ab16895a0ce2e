#!/usr/bin/env python
"""Per-wave phase timeline of the bf16-MFMA attention kernels (cr_attn_bf.hip) at the headline shape."""
import ctypes as C, os, sys
os.environ["CASTREC_TIMELINE"] = "1"      # instrumented library: python -m castrec_amd.build --timeline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import ops as O, lib as L

PREC = int(os.environ.get("PREC", "1"))
B, T, H, d = 128, 200, int(os.environ.get("H", 1)), int(os.environ.get("DH", 50))
M, D = B * T, H * d
rs = np.random.RandomState(0)
f = lambda *s: torch.randn(*s, device="cuda")
Q, K, V, R, dO = f(M, D), f(M, D), f(M, D), f(M, D), f(M, D)
out, dQ, dK, dV = f(M, D), f(M, D), f(M, D), f(M, D)
lens = np.clip(rs.lognormal(4.6, 0.9, B), 3, T).astype(int)
ids = np.zeros((B, T), np.int32)
for b in range(B): ids[b, T - lens[b]:] = 1
idd = torch.tensor(ids.reshape(-1), device="cuda")
kv = torch.tensor((ids != 0).astype(np.float32).reshape(-1), device="cuda")
qv = kv.clone()
state = torch.zeros(16, device="cuda")
drop = O.Drop(0.2, 1, state)
row_stats = torch.empty(H * B * T * 4, device="cuda")
desc = O.attn_desc(Q, K, V, D, kv, qv, R, D, out, D, B, T, H, d, rng=drop.rng(3), dead_ids=idd, row_stats=row_stats, precision=PREC)
stats = torch.empty(H * B * T * 4, device="cuda")
delta = f(H * M)
fn = getattr(L._lib, "cr_debug_attn_ts"); fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = None

def run():
    O.attn_fwd(desc)
    O.attn_bwd(desc, dO, D, dQ, dK, dV, D, stats, delta=delta)

def timeline(which, title, names):
    ts = torch.zeros(2 * B * H * 8 * 16, dtype=torch.int64, device="cuda")     # every wave of both workgroups of every (sample, head)
    for _ in range(3): run()
    torch.cuda.synchronize()
    fn(ts.data_ptr(), which)
    run()
    torch.cuda.synchronize()
    fn(None, 0)
    t = ts.cpu().numpy().reshape(-1, 16).astype(np.float64)
    live = t[:, 15] > 0
    idx = np.arange(len(live))[live]
    t = t[live]
    t0 = np.where(t[:, 0] > 0, t[:, 0], np.nan)             # (the key-owner workgroups of the fused kernel stamp slots 8.. only)
    w0 = np.nanmin(t0)
    start, end = (np.nan_to_num(t0, nan=w0) - w0) * 10.0, (t[:, 15] - w0) * 10.0
    print("==", title, " waves", len(t), " span %.1f us" % (end.max() / 1e3))
    print("wave start ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(start, [10, 50, 90, 100])))
    print("wave life  ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(end - start, [10, 50, 90, 100])))
    for sel_name, sel in (("all waves", np.ones(len(t), bool)), ("wave 0 of workgroups y = 0", ((idx % 8) == 0) & ((idx // 8) < B * H)),
                          ("wave 0 of workgroups y = 1", ((idx % 8) == 0) & ((idx // 8) >= B * H))):
        tt = t[sel]
        if len(tt) == 0:
            continue
        print(" --", sel_name, len(tt))
        used = [i for i in range(1, 15) if (tt[:, i] > 0).mean() > 0.3]
        for a, b in zip(used[:-1], used[1:]):
            ok = (tt[:, a] > 0) & (tt[:, b] > 0)
            dlt = (tt[:, b] - tt[:, a])[ok]
            if len(dlt) == 0:
                continue
            print("  stamp %2d -> %2d  %-40s median %6.0f ticks (%.2f us)  p90 %6.0f" % (a, b, names.get((a, b), ""), np.median(dlt), np.median(dlt) / 2.2e3, np.percentile(dlt, 90)))

timeline(4, "bf fwd", {(1, 2): "issue frag + K/V staging", (2, 3): "barrier", (3, 4): "scores (tile 0)", (4, 5): "softmax, mask/dropout", (5, 6): "A V", (6, 7): "store"})
timeline(5, "bf bwd (fused: query-owner pass, then key-owner pass)",
         {(1, 2): "issue frags + K/V staging", (2, 3): "barrier", (3, 4): "frag finish", (4, 5): "key-pair loop (dQ)", (5, 6): "dQ stores",
          (6, 8): "barrier between the passes", (8, 9): "issue frags + Q/dO staging", (9, 10): "barrier + tile flags + barrier", (10, 11): "frag finish",
          (11, 12): "query-pair loop (dK, dV)", (12, 13): "dK / dV stores"})
