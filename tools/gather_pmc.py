#!/usr/bin/env python3
"""The two gather kernels of bench.py's `gather` block alone (C5 table: 10 M x 256 fp32, uniform rows), 24 launches each with a
FRESH row set per launch (as bench.gather_block times them), for rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE
passes (tools/prof.sh <dir> all -- python3 tools/gather_pmc.py):
  k_embed_fwd_vec<64>    65 536 rows gathered, scaled, + positional row, masked, written as fp32 activations
  k_test_logits_v4<4,7>  413 696 rows gathered and reduced against the sequence embedding (read-only form)"""
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
torch.cuda.synchronize()
print("algorithmic bytes per launch: embed_fwd read %d write %d; test_logits read %d" % (M * (D * 4 + 4), M * D * 4, Bq * nc * (D * 4 + 4)))
