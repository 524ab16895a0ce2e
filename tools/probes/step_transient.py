"""Per-step device time of the first graph replays after capture (HIP events around every launch): how long the transient lasts
that a bench run with few warm-up steps includes."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E, synth
from castrec_amd.sampler import WarpSampler
B, T = 128, 200
corpus = synth.preset("ml-1m")
sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T)
hb = [smp.next_batch() for _ in range(16)]
smp.close()
staged = torch.from_numpy(np.stack([np.stack([a.reshape(-1) for a in (b[1], b[2], b[3], b[4], b[6], b[7])]) for b in hb]).astype(np.int32)).cuda()
hp = E.Hyper(maxlen=T, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=200, lr=1e-3)
eng = E.Engine("cast_1", corpus.usernum, corpus.itemnum, hp, B, training=True)
eng.use_id_ring(staged)
eng.capture()
eng.ids_all.copy_(staged[eng.step_number() % 16])
torch.cuda.synchronize()
N = 120
import time
big = torch.empty(32 << 20, dtype=torch.float32, device="cuda")


def replay(label, before):
    before()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        eng.graph.launch()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])
    print("%-44s steps 0..4: %s | mean 5..24: %.4f  25..59: %.4f  60..119: %.4f" % (label, " ".join("%.3f" % x for x in t[:5]), t[5:25].mean(), t[25:60].mean(), t[60:].mean()))


def idle():
    torch.cuda.synchronize(); time.sleep(1.0)


def busy():
    torch.cuda.synchronize(); time.sleep(1.0)
    for _ in range(300):
        big.mul_(1.0001)                                   # ~30 ms of streaming work on every CU


def spins():
    torch.cuda.synchronize(); time.sleep(1.0)
    for _ in range(40):
        torch.cuda._sleep(1_000_000)


replay("right after capture", lambda: None)
replay("after 1 s of idling", idle)
replay("after 1 s idle + 30 ms of streaming kernels", busy)
replay("after 1 s idle + 40 spin kernels", spins)
replay("after 1 s of idling", idle)
