"""Plain bf16 at T 200 / D 50: each parameter's gradient error against the fp64 oracle (engine's gates and masks), relative to that parameter's
own largest gradient -- for the one-launch block backward and for the tile kernels (CASTREC_NO_STACK_BWD)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import castrec_amd.engine as E
from oracle import fpmodel as fm
from test_model_gpu import make_batch, oracle_drop, oracle_with_engine_gates

B, T, D, H = int(os.environ.get("PB", 3)), int(os.environ.get("PT", 200)), 50, 1
for prec in ("bf16", "bf16x3"):
    for name, env in (("block", None), ("tile", "CASTREC_NO_STACK_BWD")):
        rs = np.random.RandomState(5)
        hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3, seed=13)
        ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3)
        if env:
            os.environ[env] = "1"
        eng = E.Engine("cast_1", 9, 45, hp, B, training=True, n_slabs=7, attn_precision=prec)
        if env:
            del os.environ[env]
        if os.environ.get("PINIT") == "engine":                   # the perturbation of test_register_layout_kernels_equal_the_tile_kernels
            eng.P.add_(0.05 * torch.randn(eng.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
        else:
            P = fm.init_params("cast_1", 9, 45, ohp, seed=8)
            P = {k: v + 0.05 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}
            eng.load_params(P)
        P = {k: v.double().cpu() for k, v in eng.get_params().items()}
        batch = make_batch(rs, B, T, 45, 9)
        eng.set_batch(*batch)
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        drop = oracle_drop(E, 13, 1, 0.2, B, T, H)
        out, G = oracle_with_engine_gates(eng, B, T, drop, prec, lambda: fm.loss_and_grads("cast_1", P, ohp, fm.to_batch(*batch), drop))
        got = eng.grads()
        errs = sorted(((float((got[k].cpu().double() - G[k]).abs().max() / max(float(G[k].abs().max()), 1e-12)), k) for k in G if not k.endswith(".bk")), reverse=True)
        for k in ("ctx_time.1.w2", "trunk.0.w2", "ctx_time.1.w1"):
            print("      ", k, "own max %.4f  error %.4f;  hid max %s" % (float(G[k].abs().max()), float((got[k].cpu().double() - G[k]).abs().max()),
                  ["%s %.2f" % (n, float(b.abs().max())) for n, b in eng._bufs.items() if n.endswith(".hid")]))
        print(prec, name, "loss", float(eng.state[0] / eng.state[2]), float(out["loss"]), "  worst (error / own max):", ["%s %.3g" % (k, e) for e, k in errs[:8]])
