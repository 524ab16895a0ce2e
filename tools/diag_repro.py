"""Runs one engine's step repeatedly on the same batch and reports which gradient tensors / buffers differ between runs."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E
import test_model_gpu as tm
model, T, D, B, n_slabs = os.environ.get("MODEL", "cast_3"), int(os.environ.get("T", 200)), 50, int(os.environ.get("B", 3)), int(os.environ.get("NSLABS", 5))
rs = np.random.RandomState(D + T)
hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.1, max_bins=200, num_context_blocks=1, lr=1e-3, seed=11)
eng = E.Engine(model, 9, 300, hp, B, training=True, n_slabs=n_slabs, attn_precision="bf16x3")
print([n for n, _, _ in eng.bwd])
g = torch.Generator().manual_seed(5)
eng.P.add_(0.05 * torch.randn(eng.P.numel(), generator=g).to(eng.P.device))
batch = tm.make_batch(rs, B, T, 300, 200)
ref_g, ref_b = None, None
for it in range(int(os.environ.get("NRUNS", 12))):
    eng.set_batch(*batch); eng.set_step(1); eng.Gflat.zero_()
    eng.launch_step(apply=False)
    torch.cuda.synchronize()
    gr = {k: v.clone() for k, v in eng.grads().items()}
    bufs = {k: v.clone() for k, v in eng._bufs.items() if v.dtype == torch.float32}
    if ref_g is None:
        ref_g, ref_b = gr, bufs
        continue
    bad = [(k, float((gr[k] - ref_g[k]).abs().max() / (ref_g[k].abs().max() + 1e-30))) for k in gr if not torch.equal(gr[k], ref_g[k])]
    badb = [(k, float((bufs[k] - ref_b[k]).abs().max())) for k in bufs if not torch.equal(bufs[k], ref_b[k]) and not torch.isnan(bufs[k]).any()]
    big = [(k, round(e, 6)) for k, e in bad if e > 1e-5]
    names = {v.data_ptr(): k for k, v in eng._bufs.items()}
    def nm(k):
        return "d(" + names.get(int(k[2:]), k) + ")" if k.startswith("d@") else k
    if len(bad) > 1:
        print("run", it, "grads differing:", len(bad), "of", len(gr), "| buffers differing:", sorted((nm(k), round(e, 6)) for k, e in badb))
        first = [k for k in bufs if not torch.equal(bufs[k], ref_b[k]) and not torch.isnan(bufs[k]).any() and nm(k).startswith("d(") and nm(k).endswith(".y)")]
        for k in bufs:
            if k in first:
                print("  ", nm(k))
                df = (bufs[k] - ref_b[k]).abs().reshape(B, T, -1).cpu().numpy()
                for b in range(B):
                    rows = np.flatnonzero(df[b].max(1) > 0)
                    if len(rows):
                        cols = np.flatnonzero(df[b].max(0) > 0)
                        print("   seq", b, "len", int((batch[0][b] != 0).sum()), "rows", rows.min(), "..", rows.max(), "n", len(rows), "cols", cols.min(), "..", cols.max(), "n", len(cols),
                              "max", float(df[b].max()), "tiles", sorted(set(int(r) // 16 for r in rows)))
    else:
        print("run", it, "same")
