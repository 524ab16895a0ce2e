"""Parity at the FULL size of BASELINE configs[3] and [4]: the forms of the engine that only large
problems reach -- auto slabs at 25 600 rows of 128 columns, the 368 000-row table's gather / scatter under Zipf ids, the streaming
Adam sweep of a 10 GB table, row-sparse Adam, the occurrence index over 10^7 rows -- against the fp64 oracle, not only for speed.

C4: models/sasrec.py at main.py:63-65's Books settings (hidden 128, 4 blocks, 4 heads, maxlen 200), batch 128, 368 000 items.
C5: the 10 M-item table (10.24 GB fp32) under maxlen 512 / hidden 256; the oracle cannot hold that table in fp64, and
need not: the graph only ever sees the rows a batch looks up, so the oracle runs on the COMPACT vocabulary of the touched rows (id ->
1 + its rank among the steps' distinct ids, same row values) -- loss, every dense gradient and the touched rows' gradients must agree,
rows outside the batch must have no gradient, and after two optimiser steps the touched rows must hold the oracle's dense-Adam values
(a row touched in step 1 only drifts on its momentum in step 2: modules.py:154-157 + sasrec.py:120-121) while, under row-sparse Adam
(a documented deviation), that row stays where step 1 left it."""
import numpy as np
import pytest
import torch

from oracle import fpmodel as fm
from test_model_gpu import (E, TOL, _other_shapes, engine_relu_gates, oracle_drop, oracle_with_engine_gates, rel,  # noqa: F401
                            worst_grad_error)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_config_c4_full_size(E, prec):
    """BASELINE configs[3] at full size: B 128, V 368 000 (Books), Zipf ids (hot rows), the engine's own slab count, both arithmetics:
    loss, every gradient (the 47 M-entry table's included) and the forward rows against the fp64 oracle."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)              # (f32 above 64 columns: the unfused kernels, by design)
        # (f32: 4e-4 of a parameter's largest gradient at the small shapes; here a positional row's gradient is the sum of 128
        #  sequences' rows with cancellation -- 5e-4 observed on pos_emb -- so the bound is the north star's 1e-3)
        eng = _other_shapes(E, "sasrec", 128, 4, 200, 4, B=128, prec=prec, itemnum=368000, zipf=1.0, n_slabs=None, dropout=0.2,
                            grad_tol=1e-3 if prec == "f32" else None)
    assert eng.M == 25600 and eng.layout.entries["item_emb"][1] == (368001, 128)


def _zipf_ids(rs, V, shape, hot):
    """ids in 1..V: half of them from a hot set (collisions inside a batch and between the steps), half anywhere in the table"""
    a = rs.randint(1, V + 1, shape)
    pick = rs.random_sample(shape) < 0.5
    a[pick] = hot[rs.randint(0, len(hot), int(pick.sum()))]
    return a


@pytest.mark.parametrize("lazy", [False, True])
def test_config_c5_full_table(E, lazy):
    V, D, H, T, L_, B, rate = 10_000_000, 256, 4, 512, 2, 2, 0.1
    rs = np.random.RandomState(5)
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=L_, num_heads=H, dropout_rate=rate, lr=1e-3, seed=11)
    eng = E.Engine("sasrec", 9, V, hp, B, training=True, attn_precision="bf16x3", lazy_adam=lazy)
    lay = eng.layout
    assert lay.entries["item_emb"][1] == (V + 1, D) and eng.P.numel() > 2_560_000_000
    assert eng.use_index == (not lazy)
    # two steps' batches: ragged left padding, ids over the whole table with a shared hot set
    hot = rs.randint(1, V + 1, 300)
    batches = []
    for _ in range(2):
        seq, pos, neg = (_zipf_ids(rs, V, (B, T), hot) for _ in range(3))
        for b, n in enumerate([0, 200]):
            seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0
        z = np.zeros_like(seq)
        batches.append((seq, pos, neg, z, z, z))
    uniq = np.unique(np.concatenate([np.concatenate([a.reshape(-1) for a in bt[:3]]) for bt in batches]))
    uniq = uniq[uniq != 0]
    n_c = len(uniq)
    assert 1000 < n_c < 6 * B * T
    cmap = np.zeros(V + 1, np.int64)
    cmap[uniq] = 1 + np.arange(n_c)
    d_uniq = torch.from_numpy(uniq).to(eng.dev)
    item = lay.view(eng.P, "item_emb")
    # off the initial point: the touched rows get rows of a usable scale (glorot over 10^7 rows is 8e-4), the dense parameters a
    # perturbation (LayerNorm gains / biases off their defaults)
    g = torch.Generator().manual_seed(3)
    item[d_uniq] = (0.1 * torch.randn(n_c, D, generator=g)).to(eng.dev)
    nt = lay.n_table
    eng.P[nt:] += (0.05 * torch.randn(lay.n_dense, generator=g)).to(eng.dev)
    pe = lay.view(eng.P, "pos_emb")
    pe += (0.05 * torch.randn(T, D, generator=g)).to(eng.dev)
    untouched = torch.from_numpy(np.setdiff1d(rs.randint(1, V + 1, 4000), uniq)).to(eng.dev)
    before_untouched = item[untouched].clone()

    def compact_params():
        P = {}
        for k in lay.logical_names():
            v = lay.view(eng.P, k)
            P[k] = torch.cat([v[0:1], v[d_uniq]]).double().cpu() if k == "item_emb" else v.detach().double().cpu().clone()
        return P

    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=L_, num_heads=H, dropout_rate=rate, lr=1e-3)
    P = compact_params()
    opt = fm.AdamTF(P, lr=1e-3)
    tol = TOL["bf16x3"]
    after_step1 = None
    for step, bt in enumerate(batches, 1):
        seq, pos, neg = bt[:3]
        cb = fm.to_batch(cmap[seq], cmap[pos], cmap[neg], bt[3], bt[4], bt[5])
        drop = oracle_drop(E, 11, step, rate, B, T, H)
        eng.set_batch(*bt)
        eng.set_step(step)
        eng.Gflat.zero_()
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        out, G = oracle_with_engine_gates(eng, B, T, drop, "bf16x3", lambda: fm.loss_and_grads("sasrec", P, ohp, cb, drop))
        st = eng.state.cpu().numpy()
        assert st[2] == float(out["istarget"].sum())
        assert st[0] / st[2] == pytest.approx(float(out["loss"]), rel=tol["loss"])
        assert rel(eng.seq_emb, out["seq_emb"].reshape(B * T, -1)) < tol["act"]
        got = eng.grads()
        # the table: touched rows against the oracle's compact table, every other row exactly zero
        gi = got.pop("item_emb")
        touched_now = np.unique(np.concatenate([a.reshape(-1) for a in bt[:3]]))
        nz = torch.nonzero(gi.abs().sum(1)).reshape(-1).cpu().numpy()
        assert set(nz.tolist()) <= set(touched_now.tolist()) - {0}
        got["item_emb"] = torch.cat([gi[0:1], gi[d_uniq]])
        del gi
        worst = worst_grad_error(got, G, "bf16x3")
        assert worst[0] < 2 * worst[2], worst
        # the optimiser step on both sides
        eng.Gflat.zero_()
        eng.set_step(step)
        eng.launch_step(apply=True)
        torch.cuda.synchronize()
        prev = {k: v.clone() for k, v in P.items()}
        P = opt.step(P, G)
        now = compact_params()
        lr = hp.lr
        for k in P:
            if k.endswith(".bk"):
                continue
            diff = (now[k] - P[k]).abs()
            if k == "item_emb" and lazy and step == 2:
                # row-sparse Adam: rows of step 1 that step 2 does not touch stay put (dense Adam moves them on their momentum)
                t2 = np.isin(uniq, touched_now)
                stay = torch.from_numpy(np.r_[False, ~t2])
                assert torch.equal(now[k][stay], after_step1[stay]) and bool(stay.any())
                moved_dense = (P[k][stay] - prev[k][stay]).abs().max()
                assert float(moved_dense) > 0.05 * lr                       # ... which the oracle's dense update did move
                diff = diff[~stay]
                big = (G[k].abs() > 1e-5)[~stay]
            else:
                big = G[k].abs() > 1e-5
            if bool(big.any()):
                assert float(diff[big].max()) < 2e-5, (k, float(diff[big].max()))
            assert float(diff.max()) < 2.5 * lr, (k, float(diff.max()))
        if step == 1:
            after_step1 = {k: v.clone() for k, v in now.items()}["item_emb"]
            g_step1 = G["item_emb"].clone()
        if not lazy and step == 2:
            # dense Adam: a row only step 1 touched has moved on its momentum, by the oracle's amount
            t2 = np.isin(uniq, touched_now)
            only1 = torch.from_numpy(np.r_[False, ~t2])
            assert bool(only1.any())
            drift_e = (now["item_emb"][only1] - after_step1[only1])
            drift_o = (P["item_emb"][only1] - prev["item_emb"][only1])
            # (where step 1's gradient was significant: Adam turns the rounding residue of a ~0 gradient into a move of up to lr on
            #  either side -- 27 of 1 194 such rows in the first run of this test -- as everywhere in these tests)
            sig = g_step1[only1].abs() > 1e-5
            dd = (drift_e - drift_o).abs()
            assert float(drift_o.abs().max()) > 0.05 * lr and float(sig.double().mean()) > 0.5 and float(dd[sig].max()) < 2e-5, float(dd[sig].max())
        # continue from the engine's values (the allowed Adam differences would otherwise show as activation differences)
        for k in P:
            P[k] = now[k]
            if k.endswith(".bk"):
                opt.m[k].zero_(); opt.v[k].zero_()
                lay.view(eng.Mom, k).zero_(); lay.view(eng.Vel, k).zero_()
    # rows no batch touched: zero moments, so no step moves them -- dense or not
    assert torch.equal(item[untouched], before_untouched)
