#!/usr/bin/env python
"""Per-wave phase timeline of the attention kernels at the headline shape (debug stamps, cr_debug_attn_ts)."""
import ctypes as C, os, sys
os.environ["CASTREC_TIMELINE"] = "1"      # instrumented library: python -m castrec_amd.build --timeline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import importlib.util as _u
_b = _u.spec_from_file_location("cr_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "context-aware-sequential-recommendation_amd", "build.py"))
_m = _u.module_from_spec(_b); _b.loader.exec_module(_m); _m.build(timeline=True)
import castrec_amd
from castrec_amd import ops as O, lib as L

B, T, H, d = 128, 200, 1, 50
M, D = B * T, H * d
rs = np.random.RandomState(0)
f = lambda *s: torch.randn(*s, device="cuda")
Q, K, V, R, dO = f(M, D), f(M, D), f(M, D), f(M, D), f(M, D)
out, dQ, dK, dV = f(M, D), f(M, D), f(M, D), f(M, D)
lens = np.clip(rs.normal(165, 60, B).astype(int), 5, T)          # ml-1m-like: most sequences fill the window
ids = np.zeros((B, T), np.int32)
for b in range(B): ids[b, T - lens[b]:] = 1
idd = torch.tensor(ids.reshape(-1), device="cuda")
kv = torch.tensor((ids != 0).astype(np.float32).reshape(-1), device="cuda")
qv = kv.clone()
state = torch.zeros(16, device="cuda")
drop = O.Drop(0.2, 1, state)
desc = O.attn_desc(Q, K, V, D, kv, qv, R, D, out, D, B, T, H, d, rng=drop.rng(3), dead_ids=idd)
stats = torch.empty(H * B * T * 4, device="cuda")
row_stats = torch.empty(H * B * T * 4, device="cuda")
desc.row_stats = row_stats.data_ptr()
delta, dQp = f(M), f(M, D)
ONE = os.environ.get("ONE_PASS", "1") == "1"
fn = getattr(L._lib, "cr_debug_attn_ts"); fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = None

def run():
    O.attn_fwd(desc)
    if ONE:
        O.attn_bwd(desc, dO, D, dQ, dK, dV, D, stats, delta=delta, dQ_part=dQp)
    else:
        O.attn_bwd(desc, dO, D, dQ, dK, dV, D, stats)

def timeline(which, title, names):
    ts = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3): run()
    torch.cuda.synchronize()
    fn(ts.data_ptr(), which)
    run()
    torch.cuda.synchronize()
    fn(None, 0)
    t = ts.cpu().numpy().reshape(-1, 16).astype(np.float64)
    live = t[:, 15] > 0
    t = t[live]
    w0 = t[:, 0].min()
    start, end = (t[:, 0] - w0) * 10.0, (t[:, 15] - w0) * 10.0
    print("==", title, " waves", len(t), " span %.1f us" % (end.max() / 1e3))
    print("wave start ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(start, [10, 50, 90, 100])))
    print("wave life  ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(end - start, [10, 50, 90, 100])))
    if os.environ.get("HEAVY") == "1":                  # only the waves that own the heaviest tile (wave 0 of workgroup y = 0)
        idx = np.arange(len(live))[live]
        nw_ = 8
        sel = ((idx % nw_) == 0) & ((idx // nw_) < B * H)
        t = t[sel]
        print("   heaviest-tile waves:", len(t))
    used = [i for i in range(1, 15) if (t[:, i] > 0).mean() > 0.3]
    for a, b in zip(used[:-1], used[1:]):
        ok = (t[:, a] > 0) & (t[:, b] > 0)
        dlt = (t[:, b] - t[:, a])[ok]
        print("  stamp %2d -> %2d  %-36s median %6.0f ticks (%.2f us)  p90 %6.0f" % (a, b, names.get((a, b), ""), np.median(dlt), np.median(dlt) * 0.46e-3, np.percentile(dlt, 90)))

FINE = os.environ.get("FINE") == "1"
def fine_fwd():
    ts = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3): run()
    torch.cuda.synchronize(); fn(ts.data_ptr(), 0); run(); torch.cuda.synchronize(); fn(None, 0)
    t = ts.cpu().numpy().reshape(-1, 16).astype(np.float64)
    t = t[t[:, 15] > 0]
    heavy = t[(t[:, 14] > 0)]            # waves whose first tile reaches the 7th key-tile pair
    print("waves with a 13-key-tile first tile:", len(heavy))
    seq = [3, 8, 9, 10, 11, 12, 13, 14, 4, 5, 6, 7]
    for a, b in zip(seq[:-1], seq[1:]):
        dlt = heavy[:, b] - heavy[:, a]
        print("  stamp %2d -> %2d median %6.0f ticks  min %6.0f" % (a, b, np.median(dlt), dlt.min()))
if FINE:
    fine_fwd(); sys.exit(0)
timeline(0, "attn fwd", {(1, 2): "issue frag + K/V staging", (2, 3): "barrier", (3, 4): "scores + softmax (tile 0)", (4, 5): "mask/dropout", (5, 6): "P V", (6, 7): "store"})
if ONE:
    timeline(3, "attn bwd (single pass)", {(1, 2): "issue frags, stats, Q/dO staging, zero dQ", (2, 3): "barrier", (3, 4): "tile flags + barrier",
                                           (4, 5): "step 0 (+ frag finish)", (5, 6): "steps 1-5", (6, 7): "steps 6-12", (7, 8): "dK / dV stores",
                                           (8, 9): "barrier"})
    sys.exit(0)
timeline(1, "attn bwd (query-owner)", {(1, 2): "issue frags + K/V staging", (2, 3): "barrier", (3, 4): "scores + softmax (tile 0)", (4, 5): "dP + softmax bwd", (5, 6): "dS scale + dQ mma", (6, 7): "stores"})
timeline(2, "attn bwd (key-owner)", {(1, 2): "issue frags + Q/dO staging", (2, 3): "barrier", (3, 4): "tile flags + barrier", (4, 5): "q-tile loop (key tile 0)", (5, 6): "stores"})
