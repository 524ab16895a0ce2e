import os, sys, subprocess
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch, castrec_amd, ctypes as C
from castrec_amd import ops as O, lib as L
B, T, D = 128, 200, 50
M = B * T
f = lambda *s: torch.randn(*s, device="cuda")
state = torch.zeros(16, device="cuda")
drop = O.Drop(0.2, 1, state)
ids = torch.randint(0, 5, (M,), device="cuda", dtype=torch.int32)
x, q_in, qkv, kv, qv, o, f_in, hid, y = f(M, D), f(M, D), f(M, 3 * D), f(M), f(M), f(M, D), f(M, D), f(M, D), f(M, D)
w = [f(D) for _ in range(4)] + [f(D, 3 * D), f(3 * D), f(D, D), f(D), f(D, D), f(D)]
bd = L.BlockDesc(M, D, w[0].data_ptr(), w[1].data_ptr(), w[4].data_ptr(), w[5].data_ptr(), w[2].data_ptr(), w[3].data_ptr(),
                 w[6].data_ptr(), w[7].data_ptr(), w[8].data_ptr(), w[9].data_ptr(), x.data_ptr(), q_in.data_ptr(), qkv.data_ptr(),
                 kv.data_ptr(), qv.data_ptr(), o.data_ptr(), f_in.data_ptr(), hid.data_ptr(), y.data_ptr(), ids.data_ptr(),
                 drop.rng(1), drop.rng(2))
s = torch.cuda.current_stream().cuda_stream
def t(name, fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) * 1e3 / reps, 1), "us")
t("ffn_fwd", lambda: L.call("cr_block_ln_ffn_fwd", C.byref(bd), s))
t("qkv_fwd", lambda: L.call("cr_block_ln_qkv_fwd", C.byref(bd), s))
'''
for dbg in (0,):
    env = dict(os.environ, CR_BLOCK_DBG=str(dbg))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("dbg=%d" % dbg, out.stdout.strip().replace("\n", " | "), out.stderr[-200:] if out.returncode else "")
