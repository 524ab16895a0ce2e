#!/usr/bin/env python3
"""The two gather kernels of bench.py's `gather` block alone (C5 table: 10 M x 256 fp32, uniform rows), 24 launches each with a
FRESH row set per launch (as bench.gather_block times them), for rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE
passes (tools/prof.sh <dir> all -- python3 tools/gather_pmc.py):
  k_embed_fwd_vec<64>    65 536 rows gathered, scaled, + positional row, masked, written as fp32 activations
  k_test_logits_v4<4,7>  413 696 rows gathered and reduced against the sequence embedding (read-only form)
  k_head_ln<64,8>        the training step's head kernel (round 5): 2 x 65 536 pos / neg rows gathered, the sequence-embedding and
                         LayerNorm-input rows streamed, one gradient row written per batch row (table gradient: occurrence index)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd  # noqa: F401
from castrec_amd import ops as O
V, D, T, N = 10_000_000, 256, 512, 24
table = torch.empty(V, D, device="cuda", dtype=torch.float32).uniform_(-0.01, 0.01)
rs = np.random.RandomState(0)
M = 128 * T
ids = torch.from_numpy(rs.randint(1, V, (N, M)).astype(np.int32)).cuda()
out = torch.empty(M, D, device="cuda"); pos = torch.randn(T, D, device="cuda")
Bq, nc = 4096, 101
cand = torch.from_numpy(rs.randint(1, V, (N, Bq, nc)).astype(np.int32)).cuda()
semb = torch.randn(Bq, D, device="cuda"); logits = torch.empty(Bq, nc, device="cuda")
for k in range(N):
    O.embed_fwd(ids[k], table, T, out, D, scale=float(D) ** 0.5, pos_table=pos, mask_ids=ids[k])
for k in range(N):
    O.test_logits(semb, D, table, cand[k], Bq, 1, D, logits)
import ctypes as C
from castrec_amd import lib as L
pn = torch.from_numpy(rs.randint(1, V, (N, 2, M)).astype(np.int32)).cuda()
semb_h, x_h, dx_h = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda"), torch.empty(M, D, device="cuda")
gam = torch.ones(D, device="cuda"); n_sl = 256
slabs = torch.zeros(n_sl, 2 * D, device="cuda"); coef = torch.empty(2, M, device="cuda"); hstate = torch.zeros(16, device="cuda")
for k in range(N):
    hd = L.HeadDesc(semb_h.data_ptr(), D, table.data_ptr(), pn[k, 0].data_ptr(), pn[k, 1].data_ptr(), M, D, V, hstate.data_ptr(), None, 0,
                    None, None, None, coef.data_ptr())
    nd = L.LnBwdDesc(x_h.data_ptr(), D, gam.data_ptr(), None, 0, dx_h.data_ptr(), D, 0, slabs.data_ptr(), slabs.data_ptr() + 4 * D, 2 * D, n_sl, M, D, 1e-8)
    L.call("cr_head_fwd_bwd_ln", C.byref(hd), C.byref(nd), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("head_ln algorithmic bytes per launch: table rows %d, all %d" % (2 * M * (D * 4 + 4), 2 * M * (D * 4 + 4) + 3 * M * D * 4 + 8 * M))
print("algorithmic bytes per launch: embed_fwd read %d write %d; test_logits read %d" % (M * (D * 4 + 4), M * D * 4, Bq * nc * (D * 4 + 4)))
