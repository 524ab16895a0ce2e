"""Where the host time of the graph-resident data-parallel step goes (one rank, RCCL): per-section host clocks + whole-step time."""
import os, sys, time, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch, torch.distributed as dist
import castrec_amd
from castrec_amd import engine as E, dist as D_, synth
from castrec_amd.sampler import WarpSampler
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, T = 128, 200
corpus = synth.preset("ml-1m")
sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T)
u, seq, pos, neg, ts_, rat, hrs, dys, _ = smp.next_batch(); smp.close()
hp = E.Hyper(maxlen=T, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=200, lr=1e-3)
eng = E.Engine("cast_1", corpus.usernum, corpus.itemnum, hp, B, training=True)
eng.set_batch(seq, pos, neg, ts_, hrs, dys)
rep = D_.EngineReplica(eng, use_graph=True)
dp = D_.DataParallel(rep, 0, 1, sparse=(os.environ.get("SPARSE") == "1"), force_collectives=True)
eng.set_step(1); eng.Mom.zero_(); eng.Vel.zero_(); eng.Gflat.zero_()
bucket = rep.bucket(); spec = rep.sparse_spec(); n_item = spec["n_item"]
def timed(name, f, acc):
    t = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
for _ in range(300): dp.step_phases()
ok = dp.capture_step()
print("whole-step capture:", ok)
for mode in ("phases", "graphs only"):
    for _ in range(20): dp.step_phases()
    torch.cuda.synchronize()
    acc = {}; N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        if mode == "phases":
            timed("step_phases", dp.step_phases, acc)
        else:
            timed("A1", lambda: rep.phase(0), acc)
            w = None
            if mode == "graphs + sync all_reduce x2": timed("ar1", lambda: dist.all_reduce(bucket[:n_item]), acc)
            if mode == "graphs + async + wait": w = timed("ar1 async", lambda: dist.all_reduce(bucket[:n_item], async_op=True), acc)
            timed("A2", lambda: rep.phase(1), acc)
            if mode != "graphs only": timed("ar2", lambda: dist.all_reduce(bucket[n_item:]), acc)
            if w is not None: timed("wait", w.wait, acc)
            timed("B", lambda: rep.phase(2), acc)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("%-32s host %.1f us/step  total %.1f us/step  |" % (mode, host / N * 1e6, tot / N * 1e6), {k: round(v / N * 1e6, 1) for k, v in acc.items()})
dist.destroy_process_group()
