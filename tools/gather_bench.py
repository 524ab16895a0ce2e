#!/usr/bin/env python3
"""HBM roofline of the item-embedding gather (BASELINE config 5: 10 M items x D=256 fp32 = 10.24 GB table,
far beyond the 256 MiB Infinity Cache).  Rows are drawn uniformly (worst case for caches) and Zipf-like.

Algorithmic bytes per gathered row: D*4 table bytes + 4 index bytes + D*4 activation write."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import ops as O

def run(V, D, M, T, zipf=False, reps=20):
    table = torch.empty(V, D, device="cuda", dtype=torch.float32).uniform_(-0.01, 0.01)
    rs = np.random.RandomState(0)
    if zipf:
        w = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** 1.05
        cdf = np.cumsum(w / w.sum())
        ids = (rs.permutation(V)[np.searchsorted(cdf, rs.random_sample(M)).clip(0, V - 1)] % (V - 1) + 1).astype(np.int32)
    else:
        ids = rs.randint(1, V, M).astype(np.int32)
    ids = torch.from_numpy(ids).cuda()
    out = torch.empty(M, D, device="cuda")
    pos = torch.randn(T, D, device="cuda")
    f = lambda: O.embed_fwd(ids, table, T, out, D, scale=float(D) ** 0.5, pos_table=pos, mask_ids=ids)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    read_b, write_b = M * (D * 4 + 4), M * D * 4
    r = dict(V=V, D=D, rows=M, dist="zipf1.05" if zipf else "uniform", us=round(us, 1),
             read_GBps=round(read_b / us / 1e3, 1), total_GBps=round((read_b + write_b) / us / 1e3, 1),
             read_frac_of_8TBps=round(read_b / us / 1e3 / 8000.0, 4), total_frac_of_8TBps=round((read_b + write_b) / us / 1e3 / 8000.0, 4))
    print(json.dumps(r), flush=True)
    del table, out
    torch.cuda.empty_cache()
    return r

if __name__ == "__main__":
    res = []
    res.append(run(10_000_000, 256, 128 * 512, 512))            # C5 per-GPU step: 65 536 rows
    res.append(run(10_000_000, 256, 1 << 20, 512))              # sustained: 1 M rows (1 GiB read + 1 GiB write)
    res.append(run(10_000_000, 256, 1 << 20, 512, zipf=True))
    res.append(run(3_000_000, 128, 200 * 5000, 200))            # Books-like width
    res.append(run(3417, 50, 128 * 200, 200))                   # headline config (cache resident, scalar path)
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
