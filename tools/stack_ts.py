#!/usr/bin/env python
"""Per-wave phase timeline of the whole-stack forward kernel (cr_stack.hip) inside the headline training step
(cast_1, B = 128, T = 200, D = 50): stamps of the LAST cr_stack_fwd launch of the step (the trunk; in pair mode --
two workgroups per sequence, one launch per block -- its last block)."""
import ctypes as C, os, sys, types
os.environ["CASTREC_TIMELINE"] = "1"      # instrumented library: python -m castrec_amd.build --timeline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import castrec_amd
from castrec_amd import engine as E, lib as L, synth
from castrec_amd.sampler import WarpSampler

B, T = int(os.environ.get("B", 128)), int(os.environ.get("T", 200))
PREC = os.environ.get("PREC", "bf16x3")
corpus = synth.preset("ml-1m")
sargs = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
smp = WarpSampler(sargs, corpus, corpus.usernum, corpus.itemnum, batch_size=B, maxlen=T)
u, seq, pos, neg, ts_, rat, hrs, dys, _ = smp.next_batch()
smp.close()
hp = E.Hyper(maxlen=T, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=200, lr=1e-3)
eng = E.Engine("cast_1", corpus.usernum, corpus.itemnum, hp, B, training=True, attn_precision=PREC)
eng.set_batch(seq, pos, neg, ts_, hrs, dys)
fn = getattr(L._lib, "cr_debug_attn_ts"); fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = None
for _ in range(3):
    eng.launch_step()
torch.cuda.synchronize()
NS = 64
ts = torch.zeros(2 * B * 8 * NS, dtype=torch.int64, device="cuda")
WHICH = int(os.environ.get("WHICH", "7"))      # 7: cr_stack_fwd, 8: cr_stack_ffn_bwd
fn(ts.data_ptr(), WHICH)
eng.launch_step()
torch.cuda.synchronize()
fn(None, 0)
t = ts.cpu().numpy().reshape(2 * B, 8, NS).astype(np.float64)
t = t[t[:, 0, 63] > 0]                              # workgroups that ran (pair mode: 2 B, else B)
print("workgroups", len(t))
w0 = t[:, :, 0].min()
print("kernel span %.1f us (wall clock, first wave start -> last wave end)" % ((t[:, :, 63].max() - w0) * 10.0 / 1e3))
print("wave start ns p50 %.0f max %.0f" % tuple(np.percentile((t[:, :, 0] - w0) * 10.0, [50, 100])))
if WHICH == 8:
    nm = {2: "weights staged (barrier)", 3: "round 0 tile", 4: "barrier", 5: "round 0 weight gradients", 6: "barrier", 7: "round 1 tile", 8: "barrier",
          9: "round 1 weight gradients", 10: "barrier", 11: "slab stores", 63: "LayerNorm gradient fold + store"}
    for wave in (0, 3, 6, 7):
        print("-- wave", wave)
        prev = t[:, wave, 1]
        for k in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
            cur = t[:, wave, k]
            ok = (cur > 0) & (prev > 0)
            if ok.sum():
                dlt = (cur - prev)[ok]
                print("  %-34s median %7.0f clk (%5.2f us)  p90 %7.0f" % (nm[k], np.median(dlt), np.median(dlt) / 2.4e3, np.percentile(dlt, 90)))
            prev = np.where(cur > 0, cur, prev)
    sys.exit(0)
names = {0: "weights staged (barrier)", 1: "phase A first tile", 2: "phase A all tiles", 3: "W1/W2 staged (2 barriers)",
         4: "scores + softmax (first tile)", 5: "A V + o (first tile)", 6: "LN2 + FFN (first tile)", 8: "all tiles done"}
# full-length sequences, per workgroup of the pair and wave: us from the wave's start to each stamp (shader clock from the wall-clock ends)
lens = (seq != 0).sum(1)
full = np.where(lens >= 193)[0]
tt = ts.cpu().numpy().reshape(2, B, 8, NS).astype(np.float64)
if (tt[1, :, 0, 63] > 0).any():
    for y in range(2):
        life = (tt[y][full][:, :, 63] - tt[y][full][:, :, 0]) * 10.0 / 1e3
        print("== workgroup y = %d of a full-length sequence: wave life us p50 %.1f max %.1f" % (y, np.median(life), life.max()))
        print("   wave: weights staged | phase A first | phase A all | W1/W2 staged | scores+softmax | A V + o | LN2 + FFN | all tiles | end    (us from the first stamp, 2.28 GHz assumed)")
        for wave in range(8):
            row = []
            for k in (0, 1, 2, 3, 4, 5, 6, 8):
                cur = tt[y][full][:, wave, 2 + k]; base = tt[y][full][:, wave, 1]
                ok = (cur > 0) & (base > 0)
                row.append("%5.2f" % (np.median((cur - base)[ok]) / 2.28e3) if ok.sum() else "  -  ")
            print("   %d: %s | life %.1f" % (wave, " | ".join(row), np.median(life[:, wave])))
for wave in (0, 3, 7):
    print("-- wave", wave)
    prev = t[:, wave, 1]
    for b in range(2):
        for k in (0, 1, 2, 3, 4, 5, 6, 8):
            cur = t[:, wave, 2 + 10 * b + k]
            ok = (cur > 0) & (prev > 0)
            if ok.sum() == 0:
                continue
            dlt = (cur - prev)[ok]
            print("  block %d  %-34s median %7.0f clk (%5.2f us)  p90 %7.0f" % (b, names[k], np.median(dlt), np.median(dlt) / 2.4e3, np.percentile(dlt, 90)))
            prev = np.where(cur > 0, cur, prev)
