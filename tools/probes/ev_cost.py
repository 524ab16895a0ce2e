import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import castrec_amd
from castrec_amd import synth
from castrec_amd.models import build_model
from castrec_amd.sampler import WarpSampler
B, T, N = 128, 200, 600
corpus = synth.preset("ml-1m")
args = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False, maxlen=T, hidden_units=50, num_blocks=2, num_heads=1,
                             dropout_rate=0.2, l2_emb=0.0, lr=1e-3, num_context_blocks=2, batch_size=B, input_context=False, max_norm=5.0)
model = build_model("cast_1", corpus.usernum, corpus.itemnum, 5, args)
smp = WarpSampler(args, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T, n_workers=1)
u, seq, pos, neg, ts, rat, hrs, dys, _ = smp.next_batch()
smp.close()
model.feed(u, seq, pos, neg, ts, hrs, dys); model.train_fed(fetch=False)
eng = model._train
host = torch.zeros(6, eng.M, dtype=torch.int32).pin_memory()
dst = torch.zeros(6, eng.M, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
def loop(label, record=False, h2d=False, every=1):
    for w in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        evs = []
        for i in range(N):
            if h2d:
                with torch.cuda.stream(side):
                    dst.copy_(host, non_blocking=True)
            eng.graph.launch()
            if record and i % every == 0:
                e = torch.cuda.Event(); e.record(); evs.append(e)
            if len(evs) > 8: evs.pop(0).synchronize()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-50s %.4f ms per step" % (label, dt / N * 1e3))
loop("graph only")
loop("graph + event record behind every step", record=True)
loop("graph + event record behind every 4th step", record=True, every=4)
loop("graph + pinned H2D copy on a side stream", h2d=True)
loop("graph + record + H2D", record=True, h2d=True)
