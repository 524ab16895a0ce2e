#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the headline shapes (B=128, T=200, D=50): HIP-event timed, 50 reps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import castrec_amd
from castrec_amd import ops as O, lib as L

B, T, D, H = 128, 200, 50, 1
M = B * T
NS = int(os.environ.get("NS", "256"))
dev = "cuda"
f = lambda *s: torch.randn(*s, device=dev)


def timeit(name, fn, reps=50, flops=None, bytes_=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    extra = ""
    if flops:
        extra = "%7.2f TF/s" % (flops / us / 1e6)
    if bytes_:
        extra = "%7.0f GB/s" % (bytes_ / us / 1e3)
    print("%-34s %8.1f us  %s" % (name, us, extra), flush=True)


state = torch.zeros(16, device=dev)
drop = O.Drop(0.2, 1, state)
x, y, w, b, r = f(M, D), f(M, D), f(D, D), f(D), f(M, D)
ids = torch.randint(0, 5, (M,), device=dev, dtype=torch.int32)
g_plain = O.gemm_desc(x, D, w, D, y, D, M, D, D)
timeit("gemm_rows plain 25600x50x50", lambda: O.gemm_rows([g_plain]), flops=2.0 * M * D * D)
g_epi = O.gemm_desc(x, D, w, D, y, D, M, D, D, bias=b, rng=drop.rng(3), residual=r, ldr=D, mask_ids=ids)
timeit("gemm_rows bias+drop+res+mask", lambda: O.gemm_rows([g_epi]), flops=2.0 * M * D * D)
g_t = O.gemm_desc(x, D, w, D, y, D, M, D, D, trans_b=True)
timeit("gemm_rows trans_b", lambda: O.gemm_rows([g_t]), flops=2.0 * M * D * D)
w3, qkv = f(D, 3 * D), f(M, 3 * D)
gq = O.gemm_desc(x, D, w3, 3 * D, qkv, 3 * D, M, D, D)
gkv = O.gemm_desc(r, D, w3[:, D:], 3 * D, qkv[:, D:], 3 * D, M, 2 * D, D)
timeit("gemm_rows QKV batched", lambda: O.gemm_rows([gq, gkv]), flops=2.0 * M * D * 3 * D)
slabs = torch.zeros(NS, 70000, device=dev)
wd = O.wgrad_desc(x, D, y, D, slabs, slabs[0, 2500:], M, D, D)
timeit("wgrad 50x50 NS=%d" % NS, lambda: O.gemm_wgrad([wd], 70000, NS), flops=2.0 * M * D * D)
wd2 = O.wgrad_desc(r, D, y, D, slabs[0, 3000:], slabs[0, 6000:], M, D, D)
timeit("wgrad 2x(50x50) NS=%d" % NS, lambda: O.gemm_wgrad([wd, wd2], 70000, NS), flops=4.0 * M * D * D)
gam, bet = f(D), f(D)
kv, qv = torch.ones(M, device=dev), torch.ones(M, device=dev)
timeit("layernorm_fwd", lambda: O.layernorm_fwd(x, D, gam, bet, y, D, M, D, x_nonzero=kv, y_nonzero=qv), bytes_=2.0 * M * D * 4)
timeit("layernorm_bwd NS=%d" % NS, lambda: O.layernorm_bwd(x, D, gam, y, D, r, D, slabs, slabs[0, D:], 70000, NS, M, D), bytes_=3.0 * M * D * 4)
timeit("eltwise gradprep drop", lambda: O.eltwise(L.ELT_GRADPREP, x, D, y, D, M, D, rng=drop.rng(5), mask_ids=ids), bytes_=2.0 * M * D * 4)
out = f(M, D)
seqids = torch.randint(1, 3000, (M,), device=dev, dtype=torch.int32)
seqids.view(B, T)[:, :40] = 0
kvalid = (seqids != 0).float()
for rate in (0.0, 0.2):
    dr_ = O.Drop(rate, 1, state)
    ad = O.attn_desc(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, kvalid, qv, x, D, out, D, B, T, H, D // H, rng=dr_.rng(7), dead_ids=seqids)
    timeit("attn_fwd rate=%.1f" % rate, lambda: O.attn_fwd(ad), flops=2.0 * D * T * (T + 1) * B)
    dqkv = f(M, 3 * D); stats = torch.zeros(H * B * T * 4, device=dev)
    timeit("attn_bwd rate=%.1f" % rate, lambda: O.attn_bwd(ad, out, D, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, stats), flops=4.0 * D * T * (T + 1) * B)
table = f(3417, D); tg = torch.zeros(3417, D, device=dev)
pos = torch.randint(1, 3417, (M,), device=dev, dtype=torch.int32)
timeit("head_fwd_bwd", lambda: O.head_fwd_bwd(x, D, table, pos, seqids, M, D, state, d_seq_emb=y, ldd=D, table_grad=tg))
fd = O.embed_fwd(seqids, table, T, out, D, scale=7.07, pos_table=f(T, D), rng=drop.rng(1), mask_ids=seqids)
timeit("embed_fwd", lambda: O.embed_fwd(seqids, table, T, out, D, scale=7.07, rng=drop.rng(1), mask_ids=seqids))
timeit("embed_bwd (atomics)", lambda: O.embed_bwd(fd, y, table_grad=tg))
n_t, n_d = 3417 * D, 62000
P_, M_, V_ = f(n_t + n_d), torch.zeros(n_t + n_d, device=dev), torch.zeros(n_t + n_d, device=dev)
state[2] = 100.0; state[4:5].view(torch.int32)[0] = 3
gt = torch.zeros(n_t, device=dev)
timeit("adam NS=%d" % NS, lambda: O.adam_step(P_, M_, V_, gt, slabs, n_t, n_d, NS, 1e-3, state), bytes_=4.0 * (7 * (n_t + n_d) + NS * n_d))
