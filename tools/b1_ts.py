#!/usr/bin/env python
"""Per-wave phase timeline of the one-launch block backward (cr_stack_bwd1.hip) inside the headline training step
(cast_1, B = 128, T = 200, D = 50): stamps of the LAST cr_stack_block_bwd launch of the step."""
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
NS = 32
ts = torch.zeros(2 * B * 8 * NS, dtype=torch.int64, device="cuda")
fn(ts.data_ptr(), 9)
eng.launch_step()
torch.cuda.synchronize()
fn(None, 0)
t = ts.cpu().numpy().reshape(2, B, 8, NS).astype(np.float64)
w0 = t[:, :, :, 0][t[:, :, :, 0] > 0].min()
print("kernel span %.1f us (wall clock, first wave start -> last wave end)" % ((t[:, :, :, 31].max() - w0) * 10.0 / 1e3))
lens = (seq != 0).sum(1)
full = lens >= 193                                    # sequences with all 13 tiles live
print("sequences with 13 live tiles:", int(full.sum()), "of", B)
names = {1: "weights staged + barrier (phase 1 opens)", 2: "phase 1 rounds (chain [+ weight gradients])", 9: "phase 1 stores, LayerNorm fold",
         3: "phase 2 staging + barriers", 6: "first tile: attention loop", 8: "first tile: row chain + stores", 4: "second tile (loop + chain)",
         7: "barrier (waiting for the slowest wave)", 5: "phase 3: images, weight gradients, stores"}
for side, sname in ((0, "K side (blockIdx.y = 0)"), (1, "Q side (blockIdx.y = 1)")):
    tt = t[side]
    life = (tt[:, :, 31] - tt[:, :, 0]) * 10.0 / 1e3
    print("== %s: wave life us  all p50 %.1f max %.1f | full-length sequences p50 %.1f max %.1f" %
          (sname, np.median(life), life.max(), np.median(life[full]), life[full].max()))
    clk = tt[:, :, 29] - tt[:, :, 30]
    mhz = np.median(clk[full] / life[full])
    print("   shader clock over the wave lives: median %.0f MHz" % mhz)
    print("   per wave, full-length sequences, us from the wave's start: loads issued | weights staged | phase 1 opens | phase 1 done")
    for wave in range(8):
        rel = lambda k: np.median((tt[full, wave, k] - tt[full, wave, 30])) / mhz
        print("     wave %d: %5.2f | %5.2f | %5.2f | %5.2f" % (wave, rel(11), rel(12), rel(1), rel(2)))
    print("   phase 1 per wave, us from the phase's opening barrier: first chain done | barrier passed | products done | barrier | last chain done | barrier | products done")
    for wave in range(8):
        rel1 = lambda k: np.median((tt[full, wave, k] - tt[full, wave, 1])) / mhz
        print("     wave %d: " % wave + " | ".join("%5.2f" % rel1(k) for k in ((24, 25, 26, 27, 28, 10, 2) if side == 1 else (24, 28, 2))))
    print("   attention phase per wave, us from the staging's last barrier: first loop done | first chain done | second tile done | barrier passed")
    for wave in range(8):
        def rel3(k):
            v = tt[full, wave, k] - tt[full, wave, 3]
            v = v[tt[full, wave, k] > 0]
            return np.median(v) / mhz if len(v) else float("nan")
        print("     wave %d: " % wave + " | ".join("%5.2f" % rel3(k) for k in (6, 8, 4, 7)))
    for wave in (0, 3, 7):
        print(" -- wave %d, full-length sequences" % wave)
        prev = tt[full, wave, 0] * 0 + np.nan
        order = [30, 15, 13, 14, 11, 12, 1, 24, 25, 26, 27, 28, 10, 2, 9, 3, 6, 8, 4, 7, 16, 17, 18, 20, 21, 22, 23, 5, 29] if side == 1 else [30, 15, 13, 14, 11, 12, 1, 24, 28, 2, 3, 6, 8, 4, 7, 16, 19, 17, 18, 20, 21, 22, 23, 5, 29]
        names[29] = "end (scatter issue)"; names[24] = "phase 1: first round's chain"; names[25] = "phase 1: next rows requested, barrier"; names[26] = "phase 1: first round's weight-gradient products"; names[27] = "phase 1: barrier"; names[28] = "phase 1: last round's chain"; names[10] = "phase 1: phase-2 loads requested, barrier"; names[2] = "phase 1: last products (+ barrier, query side)"; names[16] = "phase 3: first images up (loads, puts, barrier)"; names[19] = "phase 3: first round's products + barrier"; names[17] = "phase 3: products done"; names[18] = "phase 3: weight-gradient stores issued"; names[5] = "phase 3: embedding backward (scatter issued / small table: slab stored)"; names[20] = "small table: barrier (phase 3's images dead)"; names[21] = "small table: ids + row images up, barrier"; names[22] = "small table: one-hot products"; names[23] = "small table: (unused stamp)"; names[15] = "prologue: argument lines arrived"; names[13] = "prologue: dropout keys"; names[14] = "prologue: weight loads issued"; names[11] = "prologue: tile loads issued"; names[12] = "prologue: weights arrived and staged"
        base = None
        for k in order:
            cur = tt[full, wave, k]
            if base is None:
                base = cur
                prev = cur
                continue
            ok = (cur > 0) & (prev > 0)
            if ok.sum():
                dlt = (cur - prev)[ok]
                print("    %-44s median %7.0f clk (%5.2f us)  p90 %7.0f" % (names[k], np.median(dlt), np.median(dlt) / mhz, np.percentile(dlt, 90)))
            prev = np.where(cur > 0, cur, prev)
