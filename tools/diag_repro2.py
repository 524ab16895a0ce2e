"""One engine, one batch, NRUNS replays of the same step: every float buffer of the step is compared bit for bit with the first
run's.  For each run that differs, the differing buffers are listed with the positions (row, column -> sequence, tile, lane group,
register of layout R), the reference and the deviating values and their bit patterns -- the evidence round 3's flake note lacked.
    CASTREC_LIB=<another build of libcastrec.so> MODEL=cast_3 NRUNS=6000 python tools/diag_repro2.py > out.json
Writes ONE JSON object to stdout; nothing else goes there."""
import json
import os
import struct
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import numpy as np
import torch

import castrec_amd  # noqa: F401
from castrec_amd import engine as E
import test_model_gpu as tm

model = os.environ.get("MODEL", "cast_3")
T, D, B, H = int(os.environ.get("T", 200)), int(os.environ.get("D", 50)), int(os.environ.get("B", 3)), int(os.environ.get("H", 1))
n_slabs, nruns = int(os.environ.get("NSLABS", 5)), int(os.environ.get("NRUNS", 3000))
rs = np.random.RandomState(250)
hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.1, max_bins=200, num_context_blocks=1, lr=1e-3, seed=11)
eng = E.Engine(model, 9, 300, hp, B, training=True, n_slabs=n_slabs, attn_precision="bf16x3")
eng.P.add_(0.05 * torch.randn(eng.P.numel(), generator=torch.Generator().manual_seed(5)).to(eng.P.device))
batch = tm.make_batch(rs, B, T, 300, 200)
names = {v.data_ptr(): k for k, v in eng._bufs.items()}


def nm(k):
    return "d(" + names.get(int(k[2:]), k) + ")" if k.startswith("d@") else k


def hexf(x):
    return "%08x" % struct.unpack("<I", struct.pack("<f", float(x)))[0]


ref = None
events = []
nan_bufs = set()
for it in range(nruns):
    eng.set_batch(*batch)
    eng.set_step(1)
    eng.Gflat.zero_()
    eng.launch_step(apply=False)
    torch.cuda.synchronize()
    cur = {k: v for k, v in eng._bufs.items() if v.dtype == torch.float32}
    cur["Gs"] = eng.Gs
    if ref is None:
        ref = {k: v.clone() for k, v in cur.items()}
        for k, v in ref.items():
            if torch.isnan(v).any():
                nan_bufs.add(nm(k))
        continue
    bad = [k for k in cur if not torch.equal(cur[k], ref[k]) and nm(k) not in nan_bufs]
    if not bad:
        continue
    ev = {"run": it, "buffers": {}}
    for k in bad:
        a, b = cur[k].detach().cpu().numpy(), ref[k].cpu().numpy()
        idx = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
        info = {"n_diff": int(len(idx)), "shape": list(a.shape)}
        if a.ndim == 2 and a.shape[0] % T == 0 and k != "Gs":
            rows, cols = idx[:, 0], idx[:, 1]
            info["rows"] = sorted(set(int(r) for r in rows))[:40]
            info["cols"] = sorted(set(int(c) for c in cols))
            info["seq_tile_lg_reg"] = sorted(set((int(r) // T % B, int(r) % T // 16, int(c) % 16 // 4, int(c) % 4) for r, c in zip(rows, cols)))[:40]
        info["samples"] = [{"at": [int(x) for x in p], "ref": float(b[tuple(p)]), "got": float(a[tuple(p)]), "ref_hex": hexf(b[tuple(p)]), "got_hex": hexf(a[tuple(p)])}
                           for p in idx[:12]]
        ev["buffers"][nm(k)] = info
    events.append(ev)
    if len(events) >= 40:
        break
print(json.dumps({"model": model, "lib": os.environ.get("CASTREC_LIB", "default"), "runs": it + 1, "B": B, "T": T, "n_slabs": n_slabs,
                  "kernels": [n for n, _, _ in eng.bwd], "nan_buffers_skipped": sorted(nan_bufs),
                  "runs_that_differ": len(events), "events": events}))
