"""Known-answer tests for oracle.fpmodel derived by hand from the reference text
 -- the fp path has no reference-generated vectors (TF absent)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import fpmodel as fm
from helpers import GOLDEN


def _hp(**kw):
    d = dict(maxlen=8, hidden_units=6, num_blocks=2, num_heads=1, dropout_rate=0.0, max_bins=10)
    d.update(kw)
    return fm.Hyper(**d)


def _rand_batch(B, T, itemnum, rs, max_bins=10, pad=3):
    seq = rs.randint(1, itemnum + 1, (B, T)); pos = rs.randint(1, itemnum + 1, (B, T)); neg = rs.randint(1, itemnum + 1, (B, T))
    for b in range(B):
        n = rs.randint(0, pad + 1)
        seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0
    time = rs.randint(0, max_bins + 1, (B, T)) * (seq != 0)
    hours = rs.randint(1, 25, (B, T)) * (seq != 0)
    days = rs.randint(1, 8, (B, T)) * (seq != 0)
    return fm.to_batch(seq, pos, neg, time, hours, days)


def _perturb(P, rs, scale=0.3):
    """Move LN gains / biases off their init so every path is exercised."""
    return {k: v + scale * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}


def test_param_inventory_matches_reference_checkpoint_sizes():
    with open(os.path.join(GOLDEN, "ckpt_sizes.json")) as f:
        g = json.load(f)
    hp = fm.Hyper(maxlen=g["maxlen"], hidden_units=g["hidden_units"], num_blocks=g["num_blocks"], max_bins=g["max_bins"])
    for model, nbytes in g["sizes"].items():
        n = sum(int(np.prod(s)) for _, s, _ in fm.param_spec(model, g["usernum"], g["itemnum"], hp))
        extra = g["untrained_ln_floats_cast"] if model.startswith("cast") else 0
        assert 4 * (3 * n + extra) + 12 == nbytes, model
    # sasrec trainable floats
    n = sum(int(np.prod(s)) for _, s, _ in fm.param_spec("sasrec", 6040, 3416, hp))
    assert n == 170850 + 10000 + 2 * 12950 + 100


def test_static_positional_row0_and_parity_quirk():
    pe = fm.positional_encoding(6, 4)
    assert pe[0].tolist() == [0, 1, 0, 1, 0, 1]
    # flat-index parity: with odd dim the sin/cos columns alternate per row (modules.py:34-35)
    pe5 = fm.positional_encoding(5, 3).numpy()
    k = 1 * 5 + 0          # row 1 col 0 has odd flat index -> cos
    assert pe5[1, 0] == pytest.approx(math.cos(1 / 10000 ** 0.0))
    assert pe5[1, 1] == pytest.approx(math.sin(1 / 10000 ** (2 / 5)))


def test_layernorm_constant_row_is_beta():
    g, b = torch.tensor([2.0, 3.0, 4.0]), torch.tensor([0.5, -1.0, 7.0])
    y = fm.normalize(torch.full((2, 3), 5.0), g, b)
    assert torch.allclose(y, b.expand(2, 3))


def test_embedding_row0_invariance_and_zero_grad():
    rs = np.random.RandomState(0)
    hp = _hp()
    P = _perturb(fm.init_params("sasrec", 5, 12, hp, seed=1), rs)
    batch = _rand_batch(3, hp.maxlen, 12, rs)
    out1, g1 = fm.loss_and_grads("sasrec", P, hp, batch)
    P2 = dict(P); P2["item_emb"] = P["item_emb"].clone(); P2["item_emb"][0] += 123.0
    out2, _ = fm.loss_and_grads("sasrec", P2, hp, batch)
    assert float(out1["loss"]) == float(out2["loss"])
    assert torch.all(g1["item_emb"][0] == 0)


def test_all_padding_row_gives_zero_logits_at_init():
    rs = np.random.RandomState(1)
    hp = _hp()
    P = fm.init_params("sasrec", 5, 12, hp, seed=2)          # beta = 0
    batch = _rand_batch(2, hp.maxlen, 12, rs, pad=0)
    for k in ("seq", "pos", "neg"):
        batch[k][0] = 0
    out = fm.forward("sasrec", P, hp, batch)
    T = hp.maxlen
    assert torch.all(out["seq_emb"][:T] == 0)
    assert torch.all(out["pos_logits"][:T] == 0) and torch.all(out["istarget"][:T] == 0)
    assert math.isfinite(float(out["loss"]))


def test_all_keys_masked_row_is_uniform_over_all_T():
    """modules.py:227-244: finite fill => softmax uniform over ALL keys incl. future ones."""
    hp = _hp(num_blocks=1)
    T, D = hp.maxlen, hp.hidden_units
    P = fm.init_params("sasrec", 5, 12, hp, seed=3)
    keys = torch.zeros(1, T, D, dtype=torch.float64)         # every key row sums to 0 -> all masked
    queries = torch.ones(1, T, D, dtype=torch.float64)
    _, attn = fm.multihead_attention(queries, keys, P, "trunk.0.", 1, 0.0, None, "s")
    assert torch.allclose(attn, torch.full((1, T, T), 1.0 / T, dtype=torch.float64))


def test_causality_and_query_mask():
    rs = np.random.RandomState(4)
    hp = _hp(num_blocks=1)
    T, D = hp.maxlen, hp.hidden_units
    P = _perturb(fm.init_params("sasrec", 5, 12, hp, seed=4), rs)
    x = torch.tensor(rs.standard_normal((2, T, D)))
    q = fm.normalize(x, P["trunk.0.ln1.gamma"], P["trunk.0.ln1.beta"])
    o1, a1 = fm.multihead_attention(q, x, P, "trunk.0.", 1, 0.0, None, "s")
    x2 = x.clone(); x2[:, 5:] += 1.0
    q2 = q.clone(); q2[:, 5:] += 1.0
    o2, a2 = fm.multihead_attention(q2, x2, P, "trunk.0.", 1, 0.0, None, "s")
    assert torch.allclose(o1[:, :5], o2[:, :5])
    assert torch.all(torch.triu(a1[0], diagonal=1) == 0)
    # zero query row => its attention row is zeroed (modules.py:248-253)
    q3 = q.clone(); q3[0, 2] = 0
    _, a3 = fm.multihead_attention(q3, x, P, "trunk.0.", 1, 0.0, None, "s")
    assert torch.all(a3[0, 2] == 0)


def test_head_split_order():
    """head j of sample n lives at row j*N + n of attention_weights (modules.py:208-213)."""
    rs = np.random.RandomState(5)
    hp = _hp(num_blocks=1, num_heads=2)
    T, D = hp.maxlen, hp.hidden_units
    P = _perturb(fm.init_params("sasrec", 5, 12, hp, seed=5), rs)
    x = torch.tensor(rs.standard_normal((3, T, D)))
    q = fm.normalize(x, P["trunk.0.ln1.gamma"], P["trunk.0.ln1.beta"])
    _, a = fm.multihead_attention(q, x, P, "trunk.0.", 2, 0.0, None, "s")
    assert a.shape == (6, T, T)
    d = D // 2
    for j in range(2):
        Q = (q @ P["trunk.0.wq"] + P["trunk.0.bq"])[1, :, j * d:(j + 1) * d]
        K = (x @ P["trunk.0.wk"] + P["trunk.0.bk"])[1, :, j * d:(j + 1) * d]
        s = (Q @ K.t()) / d ** 0.5
        s = s.masked_fill(torch.triu(torch.ones(T, T), 1) > 0, fm.NEG_FILL)
        assert torch.allclose(a[j * 3 + 1], torch.softmax(s, -1))


def test_adam_first_step_closed_form():
    P = {"w": torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)}
    G = {"w": torch.tensor([0.5, -0.25, 0.0], dtype=torch.float64)}
    opt = fm.AdamTF(P, lr=1e-3)
    P2 = opt.step(dict(P), G)
    lr_t = 1e-3 * math.sqrt(1 - 0.98) / (1 - 0.9)
    want = P["w"] - lr_t * (0.1 * G["w"]) / (torch.sqrt(0.02 * G["w"] ** 2) + 1e-8)
    assert torch.allclose(P2["w"], want, rtol=0, atol=1e-15)
    assert float(P2["w"][2]) == 3.0          # zero grad, zero moments => unchanged on step 1


@pytest.mark.parametrize("model", fm.MODELS)
def test_every_graph_runs_and_has_grads_for_all_listed_params(model):
    rs = np.random.RandomState(6)
    hp = _hp(num_heads=2, dropout_rate=0.25)
    P = _perturb(fm.init_params(model, 5, 12, hp, seed=6), rs, 0.1)
    batch = _rand_batch(3, hp.maxlen, 12, rs)
    batch["test_item"] = torch.arange(1, 12)
    keep = lambda site, shape: torch.tensor(np.random.RandomState(abs(hash(site)) % 2 ** 31).rand(*shape) > 0.25)
    out, grads = fm.loss_and_grads(model, P, hp, batch, drop=keep)
    assert math.isfinite(float(out["loss"])) and 0.0 <= float(out["auc"]) <= 1.0
    assert out["test_logits"].shape == (3, 11)
    for k, g in grads.items():
        assert torch.isfinite(g).all()
        assert float(g.abs().sum()) > 0, k        # every listed variable is on the loss path
    if model in ("cast_5", "cast_6"):
        assert float(out["seq_emb"].min()) >= 0   # final ReLU of mlp (modules.py:334, cast_5.py:149)


def test_zero_rows_behind_layernorm_explode_the_gradients_at_the_initial_point():
    """Evidence for tests/test_dist_gpu.py::perturb_start.  At the TensorFlow initial point (gamma = 1,
    beta = 0, zero biases) a position whose context embedding is the zero_pad row (time bin 0, cast_1.py:30-38) stays exactly
    zero through every block, every LayerNorm on it sees variance 0, and each backward through one multiplies by
    1 / sqrt(epsilon) = 1e4 (modules.py:74-78): the reference's own graph has d loss / d ctx.0.ln1.beta ~ 1e13 there.  With the
    parameters off that point the same gradients are O(1)."""
    rs = np.random.RandomState(100)
    B, T, D, H, items = 16, 24, 20, 1, 60
    hp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.0, max_bins=10)
    seq = rs.randint(1, items + 1, (B, T)); pos = rs.randint(1, items + 1, (B, T)); neg = rs.randint(1, items + 1, (B, T))
    time = rs.randint(0, 11, (B, T))
    time[:, -1] = 0
    z = np.zeros_like(seq)
    batch = fm.to_batch(seq, pos, neg, time, z, z)
    P = fm.init_params("cast_1", 9, items, hp, seed=4)
    _, G = fm.loss_and_grads("cast_1", P, hp, batch)
    assert float(G["ctx_time.0.ln1.beta"].abs().max()) > 1e9
    g = torch.Generator().manual_seed(5)
    P = {k: v + 0.05 * torch.randn(v.shape, generator=g, dtype=v.dtype) for k, v in P.items()}
    _, G = fm.loss_and_grads("cast_1", P, hp, batch)
    assert max(float(v.abs().max()) for v in G.values()) < 1e3
