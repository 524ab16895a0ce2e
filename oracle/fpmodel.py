"""Floating-point oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

torch-CPU restatement (float64 by default) of the reference's ``modules.py`` ops
and the eleven model graphs ``models/sasrec.py`` / ``models/cast_{1..9}.py``,
with gradients from torch autograd and the TensorFlow-1.15 Adam update.

PARITY UNPINNED at op level: the arithmetic itself lives in the un-vendored
``tensorflow-gpu==1.15.2`` (reference ``requirements.txt:2``), absent here, and
the reference's tests hold no vectors for these ops.  Each function cites the
reference call sites it restates; TF-1.15 semantics used (published behaviour):
``tf.layers.dense`` / ``conv1d(kernel_size=1)`` = x @ kernel + bias;
``tf.layers.dropout`` keeps with prob 1-rate and scales kept values by
1/(1-rate); ``tf.nn.moments`` = population variance; ``tf.nn.softmax`` over the
last axis; ``AdamOptimizer``: lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
p -= lr_t * m / (sqrt(v) + eps).

Parameter names are logical (``item_emb``, ``trunk.0.wq`` ...); the mapping to
the reference's TF variable scopes is in ``TF_NAMES``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

MODELS = ["cast_1", "cast_2", "cast_3", "cast_4", "cast_5", "cast_6", "cast_7", "cast_8", "cast_9",
          "sasrec", "sasrec_static"]            # main.py:28

NEG_FILL = float(-2 ** 32 + 1)                  # modules.py:227,239


@dataclass
class Hyper:
    maxlen: int = 50
    hidden_units: int = 50
    num_blocks: int = 2
    num_heads: int = 1
    dropout_rate: float = 0.5
    max_bins: int = 200
    num_context_blocks: int = 2
    lr: float = 1e-3
    l2_emb: float = 0.0


# ---------------------------------------------------------------------------
# parameter inventory
# ---------------------------------------------------------------------------
def _stack_spec(prefix, L, D):
    out = []
    for i in range(L):
        p = "%s.%d." % (prefix, i)
        out += [(p + "ln1.gamma", (D,), "ones"), (p + "ln1.beta", (D,), "zeros"),
                (p + "wq", (D, D), "glorot"), (p + "bq", (D,), "zeros"),
                (p + "wk", (D, D), "glorot"), (p + "bk", (D,), "zeros"),
                (p + "wv", (D, D), "glorot"), (p + "bv", (D,), "zeros"),
                (p + "ln2.gamma", (D,), "ones"), (p + "ln2.beta", (D,), "zeros"),
                (p + "w1", (D, D), "glorot"), (p + "b1", (D,), "zeros"),
                (p + "w2", (D, D), "glorot"), (p + "b2", (D,), "zeros")]
    out += [(prefix + ".lnf.gamma", (D,), "ones"), (prefix + ".lnf.beta", (D,), "zeros")]
    return out


def _mlp_spec(k, D):
    return [("mlp.w1", (k * D, k * D), "glorot"), ("mlp.b1", (k * D,), "zeros"),
            ("mlp.w2", (k * D, D), "glorot"), ("mlp.b2", (D,), "zeros")]


def param_spec(model, usernum, itemnum, hp: Hyper):
    """Ordered [(name, shape, init)] of the TRAINED variables of each graph.

    (cast_1 also creates one unused LayerNorm pair per context block,
    cast_1.py:45 -- it receives no gradient and is not listed.)"""
    D, T, L = hp.hidden_units, hp.maxlen, hp.num_blocks
    time_emb = [("time_emb", (hp.max_bins + 1, D), "glorot")]
    hd_emb = [("hours_emb", (25, D), "glorot"), ("days_emb", (8, D), "glorot")]
    item = [("item_emb", (itemnum + 1, D), "glorot")]
    pos = [("pos_emb", (T, D), "glorot")]
    trunk = _stack_spec("trunk", L, D)
    ctx_t = _stack_spec("ctx_time", L, D)
    if model == "sasrec":
        return item + pos + trunk
    if model == "sasrec_static":
        return item + trunk
    if model == "cast_1":
        return time_emb + ctx_t + item + trunk
    if model == "cast_2":
        return time_emb + ctx_t + item + _mlp_spec(2, D) + trunk
    if model in ("cast_3", "cast_5"):
        return hd_emb + time_emb + ctx_t + item + _mlp_spec(3, D) + trunk
    if model in ("cast_4", "cast_6"):
        return hd_emb + time_emb + ctx_t + item + _mlp_spec(4, D) + trunk
    if model == "cast_7":
        return hd_emb + item + _mlp_spec(3, D) + trunk
    if model == "cast_8":
        return hd_emb + _stack_spec("ctx_hours", L, D) + _stack_spec("ctx_days", L, D) + item + _mlp_spec(3, D) + trunk
    if model == "cast_9":
        Lc = hp.num_context_blocks
        return (hd_emb + _stack_spec("ctx_hours", Lc, D) + _stack_spec("ctx_days", Lc, D) + time_emb +
                _stack_spec("ctx_time", Lc, D) + item + pos + _mlp_spec(4, D) + trunk)
    raise ValueError(model)


def init_params(model, usernum, itemnum, hp: Hyper, seed=0, dtype=torch.float64):
    """glorot-uniform kernels / tables (TF default initializer of get_variable and
    tf.layers.dense/conv1d), zero biases, LayerNorm gamma=1 beta=0 (modules.py:75-76)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind in param_spec(model, usernum, itemnum, hp):
        if kind == "glorot":
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out[name] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim).to(dtype)
        elif kind == "ones":
            out[name] = torch.ones(shape, dtype=dtype)
        else:
            out[name] = torch.zeros(shape, dtype=dtype)
    return out


# ---------------------------------------------------------------------------
# modules.py ops
# ---------------------------------------------------------------------------
def positional_encoding(dim, length, dtype=torch.float64):
    """modules.py:27-37 -- NB: sin on even FLAT indices, cos on odd FLAT indices of
    the row-major [length*dim] vector, exponent 2*i/dim (not 2*(i//2)/dim)."""
    v = np.array([pos / np.power(10000, 2 * i / dim) for pos in range(length) for i in range(dim)])
    v[::2] = np.sin(v[::2])
    v[1::2] = np.cos(v[1::2])
    return torch.tensor(v.reshape(length, dim), dtype=dtype)


def embedding(table, ids, zero_pad=True, scale=True):
    """modules.py:148-164: returns (outputs, lookup_table-after-zero-pad)."""
    D = table.shape[1]
    if zero_pad:
        table = torch.cat([torch.zeros(1, D, dtype=table.dtype), table[1:]], 0)     # :154-156
    out = table[ids]                                                                 # :157
    if scale:
        out = out * (D ** 0.5)                                                       # :159-160
    return out, table


def normalize(x, gamma, beta, eps=1e-8):
    """modules.py:53-80."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return gamma * ((x - mean) / ((var + eps) ** 0.5)) + beta


def _dropout(x, rate, site, drop):
    if drop is None or rate == 0.0:
        return x
    keep = drop(site, tuple(x.shape)).to(x.dtype)
    return x * keep / (1.0 - rate)


def multihead_attention(queries, keys, P, pfx, num_heads, rate, drop, site):
    """modules.py:167-277 with causality=True. Returns (outputs, attention_weights[h*N,T,T])."""
    N, T, C = queries.shape
    Q = queries @ P[pfx + "wq"] + P[pfx + "bq"]                                      # :203
    K = keys @ P[pfx + "wk"] + P[pfx + "bk"]                                         # :204
    V = keys @ P[pfx + "wv"] + P[pfx + "bv"]                                         # :205
    Q_ = torch.cat(torch.split(Q, C // num_heads, dim=2), 0)                         # :208-213 (head j = rows jN..)
    K_ = torch.cat(torch.split(K, C // num_heads, dim=2), 0)
    V_ = torch.cat(torch.split(V, C // num_heads, dim=2), 0)
    out = Q_ @ K_.transpose(1, 2)                                                    # :216
    out = out / (K_.shape[-1] ** 0.5)                                                # :219
    key_masks = torch.sign(torch.abs(keys.sum(-1)))                                  # :222
    key_masks = key_masks.repeat(num_heads, 1)[:, None, :].expand(-1, T, -1)         # :223-225
    pad = torch.full_like(out, NEG_FILL)                                             # :227
    out = torch.where(key_masks == 0, pad, out)                                      # :228-229
    tril = torch.tril(torch.ones(T, T, dtype=out.dtype))                             # :233-235
    out = torch.where(tril[None] == 0, pad, out)                                     # :239-241
    out = torch.softmax(out, dim=-1)                                                 # :244
    query_masks = torch.sign(torch.abs(queries.sum(-1)))                             # :248-249
    query_masks = query_masks.repeat(num_heads, 1)[:, :, None]                       # :250-252
    out = out * query_masks                                                          # :253
    out = _dropout(out, rate, site, drop)                                            # :256-257
    attention_weights = out                                                          # :259
    out = out @ V_                                                                   # :262
    out = torch.cat(torch.split(out, N, dim=0), 2)                                   # :265-266
    out = out + queries                                                              # :269
    return out, attention_weights


# Optional ReLU gates (test infrastructure): {site: bool tensor}.  A test that compares GRADIENTS with an implementation
# whose activations differ by ~1e-5 (the bf16 split-precision attention) hands over that implementation's gates, exactly
# as it hands over its dropout masks: the few units that sit within 1e-5 of the kink would otherwise flip on one side
# only and change a row's gradient by a finite amount.  None (the default) = plain relu.
RELU_GATES = None
# Audit of a hand-over (set by handed_over_gates): per site (units compared, units whose handed-over gate differs from this
# oracle's own `pre > 0`, largest |pre| among those, largest |pre| of the site).  RELU_CARE[site] marks the units whose gate
# matters at all (a unit dropped by the dropout that follows, or a masked row, has no gradient whatever its gate).
RELU_AUDIT = None
RELU_CARE = None


def _relu(x, site):
    if RELU_GATES is not None and site in RELU_GATES:
        g = RELU_GATES[site].reshape(x.shape)
        if RELU_AUDIT is not None:
            xd = x.detach()
            care = RELU_CARE[site].reshape(x.shape) if (RELU_CARE is not None and site in RELU_CARE) else torch.ones_like(g)
            mism = ((xd > 0) != g) & care
            k = int(mism.sum())
            RELU_AUDIT[site] = (int(care.sum()), k, float(xd.abs()[mism].max()) if k else 0.0, float(xd.abs().max()))
        return x * g.to(x.dtype)
    return torch.relu(x)


class handed_over_gates(object):
    """with handed_over_gates(gates, care) as audit: ... -- the oracle runs with another implementation's ReLU gates; on
    exit the hand-over is checked (unless check=False: plain-bf16 runs, whose activations differ by 1e-2): the gates may
    differ from the oracle's own `pre > 0` only on a vanishing set of units that sit AT the kink -- fewer than `max_share`
    of a site's units (at least `min_units` allowed) and each with |pre| < `near` * max |pre|.  Anything else means the
    hand-over is hiding a real difference, and the parity claim of the test would be void."""

    def __init__(self, gates, care=None, check=True, max_share=1e-3, min_units=2, near=1e-4):
        self.gates, self.care, self.check = gates, care, check
        self.max_share, self.min_units, self.near = max_share, min_units, near
        self.audit = {}

    def __enter__(self):
        global RELU_GATES, RELU_AUDIT, RELU_CARE
        RELU_GATES, RELU_AUDIT, RELU_CARE = self.gates, (self.audit if self.gates is not None else None), self.care
        return self.audit

    def __exit__(self, et, ev, tb):
        global RELU_GATES, RELU_AUDIT, RELU_CARE
        RELU_GATES, RELU_AUDIT, RELU_CARE = None, None, None
        if et is None and self.check and self.gates is not None:
            assert set(self.audit) == set(self.gates), "gates handed over for sites the graph does not have: %s" % sorted(set(self.gates) - set(self.audit))
            for site, (n, k, worst, top) in self.audit.items():
                assert k <= max(self.min_units, self.max_share * n), "%s: %d of %d handed-over ReLU gates differ from the oracle's own" % (site, k, n)
                assert worst <= self.near * top, "%s: a differing gate sits at |pre| = %.3e, not at the kink (max |pre| %.3e)" % (site, worst, top)
        return False


def feedforward(x, P, pfx, rate, drop, site):
    """modules.py:280-318 (two kernel-size-1 conv1d = per-position dense)."""
    h = _relu(x @ P[pfx + "w1"] + P[pfx + "b1"], site + ".relu")                     # :300-302
    h = _dropout(h, rate, site + ".ffn1", drop)                                      # :303-304
    y = h @ P[pfx + "w2"] + P[pfx + "b2"]                                            # :306-308
    y = _dropout(y, rate, site + ".ffn2", drop)                                      # :309-310
    return y + x                                                                     # :313


def mlp(x, P):
    """modules.py:321-335 (ReLU on BOTH layers)."""
    h = _relu(x @ P["mlp.w1"] + P["mlp.b1"], "mlp.relu1")
    return _relu(h @ P["mlp.w2"] + P["mlp.b2"], "mlp.relu2")


def transformer_stack(x, mask, P, prefix, L, hp, drop, final_ln=True):
    """The block loop shared by every graph (sasrec.py:65-85, cast_1.py:42-60 ...)."""
    attn = None
    for i in range(L):
        p = "%s.%d." % (prefix, i)
        q = normalize(x, P[p + "ln1.gamma"], P[p + "ln1.beta"])                      # sasrec.py:69
        x, attn = multihead_attention(q, x, P, p, hp.num_heads, hp.dropout_rate, drop, p + "attn")  # :71-78
        x = feedforward(normalize(x, P[p + "ln2.gamma"], P[p + "ln2.beta"]), P, p,
                        hp.dropout_rate, drop, p[:-1])                               # :81-82
        x = x * mask                                                                 # :83
    if final_ln:
        x = normalize(x, P[prefix + ".lnf.gamma"], P[prefix + ".lnf.beta"])          # :85
    return x, attn


# ---------------------------------------------------------------------------
# model graphs
# ---------------------------------------------------------------------------
def forward(model, P, hp: Hyper, batch, drop=None):
    """Builds the graph of ``models/<model>.py`` and returns a dict with
    loss, auc, seq_emb [B*T,D], pos_logits, neg_logits, attention_weights,
    item_table (zero-padded), and test_logits [B,101] when batch has 'test_item'.

    ``drop``: None = is_training False; else callable(site, shape) -> keep mask.
    batch: LongTensors seq,pos,neg,time,hours,days of shape [B,T]."""
    seq_ids = batch["seq"]
    B, T = seq_ids.shape
    D, L = hp.hidden_units, hp.num_blocks
    dt = P["item_emb"].dtype
    mask = (seq_ids != 0).to(dt)[..., None]                                          # sasrec.py:23
    rate = hp.dropout_rate
    ctx_attn = None

    def ctx(table, ids, prefix, Lc):
        e, _ = embedding(P[table], ids, True, True)                                  # cast_1.py:30-38
        return transformer_stack(e, mask, P, prefix, Lc, hp, drop)                   # cast_1.py:42-60

    seq, item_table = embedding(P["item_emb"], seq_ids, True, True)                  # sasrec.py:27-36
    static_pe = positional_encoding(D, T, dt)

    if model in ("sasrec", "sasrec_static"):
        if model == "sasrec":
            pe, _ = embedding(P["pos_emb"], torch.arange(T)[None].expand(B, T), False, False)  # :40-50
        else:
            pe = static_pe                                                           # :52-55
        seq = seq + pe                                                               # :56
        seq = _dropout(seq, rate, "emb", drop)                                       # :59-61
        seq = seq * mask                                                             # :62
        seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        ret_attn = attn
    elif model == "cast_1":
        tseq, ctx_attn = ctx("time_emb", batch["time"], "ctx_time", L)
        seq = seq + static_pe                                                        # cast_1.py:86
        seq = seq + tseq                                                             # :87
        seq = _dropout(seq, rate, "emb", drop)                                       # :88-90
        seq = seq * mask                                                             # :91
        seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        ret_attn = ctx_attn                                                          # cast_1.py:47,159
    elif model == "cast_2":
        tseq, ctx_attn = ctx("time_emb", batch["time"], "ctx_time", L)
        seq = (seq + static_pe) * mask                                               # cast_2.py:85-86
        c = torch.cat([seq, tseq], 2)                                                # :89
        c = _dropout(c, rate, "concat1", drop)                                       # :90-92
        seq = mlp(c, P)                                                              # :95
        seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        ret_attn = ctx_attn
    elif model in ("cast_3", "cast_4", "cast_5", "cast_6"):
        hours, _ = embedding(P["hours_emb"], batch["hours"], True, True)             # cast_3.py:32-41
        days, _ = embedding(P["days_emb"], batch["days"], True, True)                # cast_3.py:43-52
        tseq, ctx_attn = ctx("time_emb", batch["time"], "ctx_time", L)
        seq = seq + static_pe
        if model == "cast_3":
            seq = (seq + tseq) * mask                                                # cast_3.py:113-114
            c = _dropout(torch.cat([seq, hours, days], 2), rate, "concat1", drop)    # :117-120
            seq = mlp(c, P)                                                          # :124
            seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        elif model == "cast_4":
            seq = seq * mask                                                         # cast_4.py:112
            c = _dropout(torch.cat([seq, tseq], 2), rate, "concat1", drop)           # :115-118
            c = _dropout(torch.cat([c, hours, days], 2), rate, "concat2", drop)      # :121-124
            seq = mlp(c, P)
            seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        elif model == "cast_5":
            seq = seq + tseq                                                         # cast_5.py:113-114 (no dropout, no mask)
            seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)        # :118-139
            c = _dropout(torch.cat([seq, hours, days], 2), rate, "concat1", drop)    # :143-146
            seq = mlp(c, P)                                                          # :149
        else:  # cast_6
            seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)        # cast_6.py:117-138
            c = _dropout(torch.cat([seq, tseq], 2), rate, "concat1", drop)           # :142-145
            c = _dropout(torch.cat([c, hours, days], 2), rate, "concat2", drop)      # :148-151
            seq = mlp(c, P)                                                          # :155
        ret_attn = ctx_attn
    elif model in ("cast_7", "cast_8"):
        hours, _ = embedding(P["hours_emb"], batch["hours"], True, True)
        days, _ = embedding(P["days_emb"], batch["days"], True, True)
        if model == "cast_8":
            hours, _ = transformer_stack(hours, mask, P, "ctx_hours", L, hp, drop)   # cast_8.py:56-74
            days, _ = transformer_stack(days, mask, P, "ctx_days", L, hp, drop)      # cast_8.py:78-95
        seq = (seq + static_pe) * mask                                               # cast_7.py:77-78
        c = _dropout(torch.cat([seq, hours, days], 2), rate, "concat1", drop)        # :81-84
        seq = mlp(c, P)                                                              # :87
        seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        ret_attn = attn
    elif model == "cast_9":
        Lc = hp.num_context_blocks
        hours, _ = embedding(P["hours_emb"], batch["hours"], True, True)
        days, _ = embedding(P["days_emb"], batch["days"], True, True)
        hours, _ = transformer_stack(hours, mask, P, "ctx_hours", Lc, hp, drop)      # cast_9.py:56-74
        days, _ = transformer_stack(days, mask, P, "ctx_days", Lc, hp, drop)         # :78-95
        tseq, _ = ctx("time_emb", batch["time"], "ctx_time", Lc)                     # :101-129
        pe, _ = embedding(P["pos_emb"], torch.arange(T)[None].expand(B, T), False, False)  # :149-159
        seq = seq + pe                                                               # :160
        c = _dropout(torch.cat([seq, tseq, hours, days], 2), rate, "concat1", drop)  # :163-168
        seq = mlp(c, P) * mask                                                       # :171-174
        seq, attn = transformer_stack(seq, mask, P, "trunk", L, hp, drop)
        ret_attn = attn
    else:
        raise ValueError(model)

    pos = batch["pos"].reshape(B * T)                                                # sasrec.py:87-88
    neg = batch["neg"].reshape(B * T)
    seq_emb = seq.reshape(B * T, D)                                                  # :91
    pos_emb, neg_emb = item_table[pos], item_table[neg]                              # :89-90
    pos_logits = (pos_emb * seq_emb).sum(-1)                                         # :100
    neg_logits = (neg_emb * seq_emb).sum(-1)                                         # :101
    istarget = (pos != 0).to(dt)                                                     # :104
    loss = (-torch.log(torch.sigmoid(pos_logits) + 1e-24) * istarget
            - torch.log(1 - torch.sigmoid(neg_logits) + 1e-24) * istarget).sum() / istarget.sum()   # :105-108
    auc = (((torch.sign(pos_logits - neg_logits) + 1) / 2) * istarget).sum() / istarget.sum()      # :113-115
    if hp.l2_emb != 0.0:
        # modules.py:149-153: every embedding's `lookup_table` VARIABLE (row 0 included: the regulariser sits on the
        # variable, not on the zero-padded concat) carries tf.contrib.layers.l2_regularizer(l2) = l2 * sum(w^2) / 2;
        # sasrec.py:109-110 adds the collection to the loss
        for k in ("item_emb", "pos_emb", "time_emb", "hours_emb", "days_emb"):
            if k in P:
                loss = loss + hp.l2_emb * 0.5 * (P[k] ** 2).sum()
    out = dict(loss=loss, auc=auc, seq_emb=seq_emb, pos_logits=pos_logits, neg_logits=neg_logits,
               attention_weights=ret_attn, item_table=item_table, istarget=istarget)
    if "test_item" in batch:
        te = item_table[batch["test_item"]]                                          # :93-94
        tl = (seq_emb @ te.t()).reshape(B, T, -1)[:, -1, :]                          # :95-97
        out["test_logits"] = tl
    return out


# ---------------------------------------------------------------------------
# training step (autograd + TF Adam), used for gradient / update parity and as the CPU baseline
# ---------------------------------------------------------------------------
class AdamTF:
    """tf.train.AdamOptimizer(lr, beta2=0.98) (sasrec.py:120): dense update of every variable."""

    def __init__(self, P, lr, beta1=0.9, beta2=0.98, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}

    def step(self, P, G):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k in P:
            g = G[k]
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            P[k] = P[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)
        return P


def loss_and_grads(model, P, hp, batch, drop=None):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = forward(model, leaves, hp, batch, drop)
    out["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return out, grads


def to_batch(seq, pos, neg, time=None, hours=None, days=None, test_item=None):
    f = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.long)
    b = dict(seq=f(seq), pos=f(pos), neg=f(neg))
    z = torch.zeros_like(b["seq"])
    b["time"] = f(time) if time is not None else z
    b["hours"] = f(hours) if hours is not None else z
    b["days"] = f(days) if days is not None else z
    if test_item is not None:
        b["test_item"] = f(test_item)
    return b


# logical name -> TF variable name in the reference's checkpoints (pinned by tests/test_tf_bundle.py against the
# bundle index files of the reference's own saved models, tests/golden/tf_index)
def tf_name(name):
    scope = {"trunk": "SASRec/num_blocks_%d", "ctx_time": "CONTEXT/timeseq_num_blocks_%d"}
    fixed = {"item_emb": "SASRec/input_embeddings/lookup_table", "pos_emb": "SASRec/dec_pos/lookup_table",
             "time_emb": "CONTEXT/time_embeddings/lookup_table",
             "hours_emb": "INPUT-CONTEXT/hours_embeddings/lookup_table", "days_emb": "INPUT-CONTEXT/days_embeddings/lookup_table",
             "trunk.lnf.beta": "SASRec/ln/Variable", "trunk.lnf.gamma": "SASRec/ln/Variable_1",
             "ctx_time.lnf.beta": "CONTEXT/ln/Variable", "ctx_time.lnf.gamma": "CONTEXT/ln/Variable_1",
             "mlp.w1": "SASRec/MLP/dense/kernel", "mlp.b1": "SASRec/MLP/dense/bias",
             "mlp.w2": "SASRec/MLP/dense_1/kernel", "mlp.b2": "SASRec/MLP/dense_1/bias"}
    if name in fixed:
        return fixed[name]
    parts = name.split(".")
    if parts[0] in scope and parts[1].isdigit():
        base = scope[parts[0]] % int(parts[1])
        leaf = ".".join(parts[2:])
        # the context blocks create one extra, unused LayerNorm pair first (`ln`, cast_1.py:45): theirs are ln_1 / ln_2
        l1, l2 = ("ln_1", "ln_2") if parts[0] == "ctx_time" else ("ln", "ln_1")
        m = {"ln1.beta": l1 + "/Variable", "ln1.gamma": l1 + "/Variable_1", "ln2.beta": l2 + "/Variable",
             "ln2.gamma": l2 + "/Variable_1", "wq": "self_attention/dense/kernel", "bq": "self_attention/dense/bias",
             "wk": "self_attention/dense_1/kernel", "bk": "self_attention/dense_1/bias",
             "wv": "self_attention/dense_2/kernel", "bv": "self_attention/dense_2/bias",
             "w1": "multihead_attention/conv1d/kernel", "b1": "multihead_attention/conv1d/bias",
             "w2": "multihead_attention/conv1d_1/kernel", "b2": "multihead_attention/conv1d_1/bias"}
        return base + "/" + m[leaf]
    return None
