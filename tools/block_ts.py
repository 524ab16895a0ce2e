#!/usr/bin/env python
"""Per-wave phase timeline of the fused LN1+QKV forward kernel (debug stamps, cr_debug_block_ts)."""
import ctypes as C, os, sys
os.environ["CASTREC_TIMELINE"] = "1"      # instrumented library: python -m castrec_amd.build --timeline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import importlib.util as _u
_b = _u.spec_from_file_location("cr_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "context-aware-sequential-recommendation_amd", "build.py"))
_m = _u.module_from_spec(_b); _b.loader.exec_module(_m); _m.build(timeline=True)
import castrec_amd
from castrec_amd import ops as O, lib as L

B, T, D = 128, 200, 50
M = B * T
f = lambda *s: torch.randn(*s, device="cuda")
state = torch.zeros(16, device="cuda")
drop = O.Drop(0.2, 1, state)
ids = torch.randint(0, 5, (M,), device="cuda", dtype=torch.int32)
x, q_in, qkv, kv, qv, o, f_in, hid, y = f(M, D), f(M, D), f(M, 3 * D), f(M), f(M), f(M, D), f(M, D), f(M, D), f(M, D)
w = [f(D) for _ in range(4)] + [f(D, 3 * D), f(3 * D), f(D, D), f(D), f(D, D), f(D)]
bd = L.BlockDesc(M, D, w[0].data_ptr(), w[1].data_ptr(), w[4].data_ptr(), w[5].data_ptr(), w[2].data_ptr(), w[3].data_ptr(),
                 w[6].data_ptr(), w[7].data_ptr(), w[8].data_ptr(), w[9].data_ptr(), x.data_ptr(), q_in.data_ptr(), qkv.data_ptr(),
                 kv.data_ptr(), qv.data_ptr(), o.data_ptr(), f_in.data_ptr(), hid.data_ptr(), y.data_ptr(), ids.data_ptr(),
                 drop.rng(1), drop.rng(2))
s = torch.cuda.current_stream().cuda_stream
NS = int(os.environ.get("NS", "256"))
stride = 4 * D * D + 8 * D + 3 * D * D + 3 * D + 64
slabs = torch.zeros(NS * stride, device="cuda")
dy, d_o, dqkv, dx = f(M, D), f(M, D), f(3 * M, D), f(M, D)
def sl(off): return slabs.data_ptr() + 4 * off
bb = L.BlockBwdDesc(bd, dy.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(), dx.data_ptr(), 0,
                    sl(0), sl(D), sl(2 * D), sl(2 * D + 3 * D * D), sl(5 * D + 3 * D * D), sl(6 * D + 3 * D * D),
                    sl(7 * D + 3 * D * D), sl(7 * D + 4 * D * D), sl(8 * D + 4 * D * D), sl(8 * D + 5 * D * D), stride, NS)
dll = L._lib
fn = getattr(dll, "cr_debug_block_ts")
fn.argtypes = [C.c_void_p]; fn.restype = None

def timeline(entry, desc, nwg, names):
    ts = torch.zeros(nwg * 8 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3): L.call(entry, C.byref(desc), s)
    torch.cuda.synchronize()
    fn(ts.data_ptr())
    L.call(entry, C.byref(desc), s)
    torch.cuda.synchronize()
    fn(None)
    t = ts.cpu().numpy().reshape(-1, 1, 16).astype(np.float64)
    live = t[:, :, 15] > 0
    w0 = t[:, :, 0][live].min()
    start = (t[:, :, 0][live] - w0) * 10.0          # ns (100 MHz)
    end = (t[:, :, 15][live] - w0) * 10.0
    print("==", entry, " waves", int(live.sum()), " span (first start -> last end stamp) %.1f us" % (end.max() / 1e3))
    print("wave start  ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(start, [10, 50, 90, 100])))
    print("wave life   ns: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(end - start, [10, 50, 90, 100])))
    used = [i for i in range(1, 15) if (t[:, :, i][live] > 0).mean() > 0.5]   # slots 0 and 15 are wall-clock stamps (span), 1..14 s_memtime
    for a, b in zip(used[:-1], used[1:]):
        dlt = (t[:, :, b] - t[:, :, a])[live]
        print("  stamp %2d -> %2d  %-34s median %6.0f ticks (%.2f us)  p90 %6.0f" % (a, b, names.get((a, b), ""), np.median(dlt), np.median(dlt) * 0.46e-3, np.percentile(dlt, 90)))

timeline("cr_block_ln_qkv_fwd", bd, (M + 63) // 64, {(1, 2): "issue loads", (2, 3): "barrier wait", (3, 4): "LN + q_in store", (4, 5): "Q proj + store", (5, 6): "K,V proj + store"})
timeline("cr_block_ln_qkv_bwd", bb, NS, {(1, 2): "weights + streams issued + weights->LDS", (2, 3): "phase-1 puts", (3, 4): "barrier", (4, 5): "wgrad q + dq_in mma", (5, 6): "barrier + phase 2 (dK, x) puts + wgrad k + mma", (6, 7): "barrier + phase 3 (dV) + wgrad v + mma", (7, 8): "barrier", (8, 9): "phase 4: LN1 bwd + store", (9, 14): "to the start of the group fold + slab stores"})
timeline("cr_block_ln_ffn_bwd", bb, NS, {(1, 2): "issue weight + stream loads", (2, 3): "weights -> LDS", (3, 4): "mask + zero rows", (4, 5): "put g2", (5, 6): "put hid, f_in", (6, 7): "barrier + wgrad2 + barrier", (7, 8): "dhid + gate + barrier", (8, 9): "wgrad1 + df + barrier", (9, 10): "LN2 bwd + store + barrier", (10, 14): "to the start of the group fold + slab stores"})
