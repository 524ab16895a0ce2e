"""Per-wave timeline of one cr_wide kernel: build with CASTREC_EXTRA_FLAGS=-DWD_TS=<k>, k = 1 qkv_fwd, 2 ffn_fwd, 3 ffn_bwd,
4 qkv_bwd; phase durations in core clocks (stamps of the LAST launch of that kernel; clocks of different CUs are not comparable,
only differences inside a wave are used).
    CASTREC_EXTRA_FLAGS=-DWD_TS=1 python -m castrec_amd.build && python tools/wide_ts.py 6     (argument: panels of that kernel)"""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import castrec_amd  # noqa
from castrec_amd import engine as E, lib as L

D, H, T, B, NB = 128, 4, 200, 128, 4
hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=NB, num_heads=H, dropout_rate=0.2, max_bins=200, lr=1e-3, seed=42)
eng = E.Engine("sasrec", 6040, 3416, hp, B, training=True)
rs = np.random.RandomState(0)
seq = rs.randint(1, 3417, (B, T)); eng.set_batch(seq, seq, seq, seq * 0, seq * 0, seq * 0)
for _ in range(3):
    eng.launch_step()
torch.cuda.synchronize()
buf = np.zeros(512 * 8 * 64, np.uint64)
fn = L.lib.cr_wide_ts_read
fn.restype = C.c_int
assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
ts = buf.reshape(512, 8, 64).astype(np.int64)
nwg = 200
t = ts[:nwg]
npan = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print("wave lifetime to the end of the panels", "mean %7.0f  min %7.0f  max %7.0f" % ((t[:, :, 8 + 4 * npan] - t[:, :, 0]).mean(), (t[:, :, 8 + 4 * npan] - t[:, :, 0]).min(), (t[:, :, 8 + 4 * npan] - t[:, :, 0]).max()))
def stat(a): return "mean %7.0f  min %7.0f  max %7.0f" % (a.mean(), a.min(), a.max())
print("rows loaded + stats   ", stat(t[:, :, 1] - t[:, :, 0]))
print("LN, q_in store, split ", stat(t[:, :, 2] - t[:, :, 1]))
print("first put + barrier   ", stat(t[:, :, 8] - t[:, :, 2]))
for i in range(npan):
    b = 8 + 4 * i
    print("panel %d: issue+mma %s | epilogue %s | put %s | barrier %s" % (i, stat(t[:, :, b + 1] - t[:, :, b]), stat(t[:, :, b + 2] - t[:, :, b + 1]),
                                                                         stat(t[:, :, b + 3] - t[:, :, b + 2]), stat(t[:, :, b + 4] - t[:, :, b + 3])))
