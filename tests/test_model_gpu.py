"""End-to-end parity of the HIP training path (engine = C-ABI kernels) against the fp64 oracle on
identical inputs and parameters: loss, AUC, EVERY parameter gradient, Adam-updated parameters, test
logits -- for all eleven graphs, with and without dropout (the oracle is fed the exact masks of the
counter-based generator).  Tolerance: 1e-3 is the north-star bound for fp32 forward logits.
  attn_precision "f32"    (exact fp32 MFMA everywhere): observed ~1e-6, asserted 2e-4 (relative)
  attn_precision "bf16x3" (attention products on bf16 MFMA, hi + lo split operands, the engine's default):
                          activations (the quantity the 1e-3 bound is stated on): observed ~1e-5, asserted 2e-4; gradients,
                          against each parameter's own largest gradient: typically 1e-4, up to 9e-4 over 144 surveyed draws of
                          the unfused dropout case (dense layers on the split form as well), asserted 2e-3
  attn_precision "bf16"   (plain bf16 attention operands; BASELINE.json configs[1] names bf16, the reference is fp32):
                          asserted 3e-2 on activations and the loss, 2e-1 of the gradient scale on gradients (score errors of
                          2^-8 |s| are exponentiated by the softmax): the throughput option, not a parity claim"""
import math
import types

import numpy as np
import pytest
import torch

import dropout_ref as dr
from oracle import fpmodel as fm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import castrec_amd  # noqa: F401
    from castrec_amd import engine
    return engine


def make_batch(rs, B, T, itemnum, max_bins, all_zero_time_rows=True):
    seq = rs.randint(1, itemnum + 1, (B, T)); pos = rs.randint(1, itemnum + 1, (B, T)); neg = rs.randint(1, itemnum + 1, (B, T))
    for b in range(B):
        n = rs.randint(0, T - 2)
        seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0
    seq[0, :] = 0; pos[0, :] = 0; neg[0, :] = 0                  # an all-padding sequence
    seq[0, -1] = 3; pos[0, -1] = 4; neg[0, -1] = 5               # ... with a single real position
    time = rs.randint(0, max_bins + 1, (B, T)) * (seq != 0)
    time[:, -1] = 0                                              # most recent item is always bin 0 (sampler.py:61-64)
    if all_zero_time_rows and B > 2:
        time[2, :] = 0                                           # all interactions in one bin: no valid context key
    hours = rs.randint(1, 25, (B, T)) * (seq != 0); days = rs.randint(1, 8, (B, T)) * (seq != 0)
    return seq, pos, neg, time, hours, days


def oracle_drop(eng_mod, seed, step, rate, B, T, H):
    def drop(site, shape):
        sid = eng_mod.site_id(site)
        if site.endswith(".attn"):
            return torch.tensor(dr.attn_mask(seed, step, sid, rate, H, B, T))
        ncol = shape[-1]
        return torch.tensor(dr.rows_mask(seed, step, sid, rate, B * T, ncol).reshape(shape))
    return drop


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def same_run(a, b):
    """Two engines after the same batches in the same order: the dense parameters agree to rounding (the table gradients are
    float atomics, so runs of several steps are not bit-identical, and Adam turns a rounding-level change of a near-zero gradient
    into a step of ~lr on a few parameters: medians of 1e-8 .. 7e-6 over repeated runs), where a swapped or stale batch moves MOST
    parameters by ~lr (median 3e-3 when the first version of feed() let a copy overtake the step that still read its slot)."""
    d = (a.P - b.P).abs()
    st = (float(d.median()), float(torch.quantile(d[:1 << 20], 0.99)), float(d.max()))
    ok = st[0] <= 2e-4 and st[1] <= 1.5e-3 and st[2] <= 5e-3
    if not ok:
        print("same_run: median %.3g, 99 %% quantile %.3g, max %.3g of |difference|" % st)
    return ok


CASES = [(m, 0.0, True) for m in fm.MODELS] + [("sasrec", 0.3, True), ("cast_1", 0.3, True), ("cast_4", 0.25, True), ("cast_9", 0.2, True)]
# the unfused kernel path (used for hidden sizes > 64) stays covered at the same small shapes
CASES += [("sasrec", 0.0, False), ("cast_1", 0.3, False), ("cast_3", 0.0, False), ("cast_9", 0.2, False)]
CASES = [c + ("f32",) for c in CASES]
# the bf16-MFMA attention kernels (csrc/cr_attn_bf.hip), fused and unfused row phases, one and two heads
CASES += [("sasrec", 0.3, True, "bf16x3"), ("cast_1", 0.3, True, "bf16x3"), ("cast_5", 0.0, True, "bf16x3"), ("cast_9", 0.2, True, "bf16x3"),
          ("cast_1", 0.3, False, "bf16x3"), ("sasrec", 0.0, False, "bf16x3"), ("cast_1", 0.3, True, "bf16"), ("sasrec", 0.0, False, "bf16")]
TOL = {"f32": dict(grad=2e-4, act=2e-5, loss=2e-5, auc=1e-6), "bf16x3": dict(grad=2e-3, act=2e-4, loss=2e-4, auc=1e-6),
       "bf16": dict(grad=2e-1, act=3e-2, loss=2e-2, auc=None)}


def engine_relu_gates(eng, B, T, drop):
    """(gates, care): the ReLU gates of an engine's last forward, by the oracle's site names (fpmodel.RELU_GATES): read from
    the stored post-ReLU activations (FFN hidden: post-dropout, so a dropped unit's gate is unknown -- and irrelevant, its
    mask is 0: `care` is False there, and fpmodel.handed_over_gates audits the rest against the oracle's own pre > 0)."""
    gates, care = {}, {}
    live = (eng.ids["seq"] != 0).reshape(B * T, 1)              # every block ends with `*= mask` (sasrec.py:83): a padded row has no gradient
    for name, buf in eng._bufs.items():
        if name.endswith(".hid"):
            site = name[:-4]                                    # "trunk.0"
            g = buf > 0
            c = live.expand(B * T, buf.shape[1])
            if drop is not None:
                kept = drop(site + ".ffn1", (B * T, buf.shape[1])).reshape(B * T, -1).to(g.device)
                g = g | ~kept
                c = c & kept
            care[site + ".relu"] = c.reshape(B, T, -1).cpu()
            gates[site + ".relu"] = g.reshape(B, T, -1).cpu()
    if "mlp.h" in eng._bufs:
        gates["mlp.relu1"] = (eng._bufs["mlp.h"] > 0).reshape(B, T, -1).cpu()
        outb = eng._bufs["seq_emb"] if eng.model in ("cast_5", "cast_6") else eng._bufs["x0"]
        rowlive = (eng.ids["seq"] != 0).reshape(B * T, 1) if eng.model == "cast_9" else torch.ones(B * T, 1, dtype=torch.bool, device=outb.device)
        gates["mlp.relu2"] = ((outb > 0) | ~rowlive).reshape(B, T, -1).cpu()     # cast_9 masks the rows after the ReLU (their gradient is 0)
        care["mlp.relu2"] = rowlive.expand(B * T, outb.shape[1]).reshape(B, T, -1).cpu()
    return gates, care


def oracle_with_engine_gates(eng, B, T, drop, prec, fn):
    """fn() = the oracle run.  f32: the oracle's own gates.  Split precision: the engine's gates, audited (differences only
    at the kink, on < 1e-3 of the units: fpmodel.handed_over_gates).  Plain bf16 (activations differ by 1e-2; the
    throughput option, not a parity claim): handed over without the audit."""
    if prec == "f32":
        return fn()
    gates, care = engine_relu_gates(eng, B, T, drop)
    with fm.handed_over_gates(gates, care, check=(prec == "bf16x3")):
        return fn()


SEED_SHIFT = 0          # tools: shift to survey other draws


def worst_grad_error(got, G, prec):
    """(error, parameter name, threshold): largest element error of a parameter against that parameter's largest
    gradient (floored at 1e-3 of the global scale; plain bf16: against the global scale).  d loss / d bk == 0
    identically (rounding noise on both sides): not compared."""
    gmax = max(float(G[k].abs().max()) for k in G)
    names = [k for k in G if not k.endswith(".bk")]
    def emax(k):
        a = got[k].cpu().double().numpy(); b = G[k].numpy()
        return float(np.abs(a - b).max() / (gmax if prec == "bf16" else max(np.abs(b).max(), 1e-3 * gmax)))
    w = max((emax(k), k) for k in names)
    return w[0], w[1], TOL[prec]["grad"]


@pytest.mark.parametrize("model,rate,fused,prec", CASES)
def test_model_grads_and_adam_match_oracle(E, model, rate, fused, prec):
    tol = TOL[prec]
    # (zlib.crc32, not hash(): str hashes are salted per process, which made this test draw different data in every run)
    import zlib
    rs = np.random.RandomState(zlib.crc32(model.encode()) % 1000 + int(rate * 100) + SEED_SHIFT)
    B, T, D, H, itemnum, max_bins = 5, 24, 20, 2, 37, 12
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=rate, max_bins=max_bins,
                 num_context_blocks=1, lr=1e-3, seed=7)
    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=rate, max_bins=max_bins,
                   num_context_blocks=1, lr=1e-3)
    eng = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=7, fused=fused, attn_precision=prec)
    # oracle parameters: perturbed init so LN gains / biases are off their defaults
    P = fm.init_params(model, 9, itemnum, ohp, seed=3)
    P = {k: v + 0.1 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}
    assert sorted(P) == sorted(eng.layout.logical_names())
    eng.load_params(P)
    P = {k: v.double() for k, v in eng.get_params().items()}        # the fp32-rounded values the GPU holds
    P = {k: v.cpu() for k, v in P.items()}
    seq, pos, neg, time, hours, days = make_batch(rs, B, T, itemnum, max_bins)
    batch = fm.to_batch(seq, pos, neg, time, hours, days)
    opt = fm.AdamTF(P, lr=1e-3)
    for step in (1, 2):
        drop = oracle_drop(E, 7, step, rate, B, T, H) if rate > 0 else None
        eng.set_batch(seq, pos, neg, time, hours, days)
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        out, G = oracle_with_engine_gates(eng, B, T, drop, prec, lambda: fm.loss_and_grads(model, P, ohp, batch, drop))   # see _other_shapes
        st = eng.state.cpu().numpy()
        n = st[2]
        assert n == float(out["istarget"].sum())
        assert st[0] / n == pytest.approx(float(out["loss"]), rel=tol["loss"])
        if tol["auc"] is not None:                       # (a sign(p - n) count: plain bf16 may flip near-ties)
            assert st[1] / n == pytest.approx(float(out["auc"]), abs=tol["auc"])
        got = eng.grads()
        # d loss / d bk == 0 identically (adding a per-query constant to all scores leaves softmax unchanged),
        # so both sides hold rounding noise there: errors are measured against the global gradient scale.
        gmax = max(float(G[k].abs().max()) for k in G)
        worst = worst_grad_error(got, G, prec)                              # bk: asserted to vanish, below
        assert worst[0] < worst[2], worst
        for k in G:
            if k.endswith(".bk"):
                assert float(got[k].abs().max()) < max(1e-5, 0.1 * tol["grad"]) * gmax and float(G[k].abs().max()) < 1e-9 * gmax
        assert rel(eng.seq_emb, out["seq_emb"].reshape(B * T, -1)) < tol["act"]
        # now apply Adam on both sides (engine: re-run the step with the update; grads are recomputed)
        eng.Gt.zero_()
        eng.set_step(step)
        eng.launch_step(apply=True)
        torch.cuda.synchronize()
        prev = {k: v.clone() for k, v in P.items()}
        P = opt.step(P, G)
        now = eng.get_params()
        # Adam divides by sqrt(v): on the zero-gradient bk it amplifies rounding noise to O(lr) moves on
        # BOTH sides (in TensorFlow too), so bk is re-synchronised instead of compared.
        # One Adam step moves a weight by ~lr * g / (|g| + eps): insensitive to the gradient's rounding error
        # wherever |g| >> eps = 1e-8, ill-conditioned (sign flips of rounding noise become +-lr moves, in
        # TensorFlow too) where |g| ~ eps.  So: elements with a significant gradient must agree to 2% of a
        # step, the others only have to stay within one step.  (The Adam kernel itself is checked to 1e-6 in
        # test_ops_gpu.py.)
        lr = hp.lr
        worst_big = worst_all = (0.0, "")
        for k in P:
            if k.endswith(".bk"):
                continue
            diff = (now[k].cpu().double() - P[k]).abs()
            big = G[k].abs() > 1e-5
            if bool(big.any()):
                worst_big = max(worst_big, (float(diff[big].max()), k))
            worst_all = max(worst_all, (float(diff.max()), k))
        if prec != "bf16":                               # (plain bf16: small gradients may change sign, i.e. a whole +-lr step)
            assert worst_big[0] < 2e-5, worst_big
        assert worst_all[0] < 2.5 * lr, worst_all
        if prec == "f32":
            # smaller gradients too: wherever |g| > 1e-7 (ten times Adam's eps scale at step 1) the update must have the
            # oracle's sign and at least a third of its size -- a wrong sign on a small-gradient element would pass the
            # 2.5 * lr bound above
            for k in P:
                if k.endswith(".bk"):
                    continue
                # (from step 2 on the update follows the moments, not g: elements whose oracle update is itself below 5 % of
                # a step are near a zero crossing of m and carry no sign information)
                sel = (G[k].abs() > 1e-7) & ((P[k] - prev[k]).abs() > 0.05 * lr)
                if not bool(sel.any()):
                    continue
                up_e = (now[k].cpu().double() - prev[k])[sel]
                up_o = (P[k] - prev[k])[sel]
                assert bool((torch.sign(up_e) == torch.sign(up_o)).all()), k
                assert bool((up_e.abs() > up_o.abs() / 3).all()), k
        # continue from the engine's parameters so step 2 compares gradients at identical points (the allowed
        # 1e-4 Adam differences would otherwise show up as 1e-4 activation differences)
        for k in P:
            P[k] = now[k].cpu().double()
            if k.endswith(".bk"):
                opt.m[k].zero_(); opt.v[k].zero_()
        off = eng.layout
        for k in P:
            if k.endswith(".bk"):
                eng.layout.view(eng.Mom, k).zero_(); eng.layout.view(eng.Vel, k).zero_()
        loss, auc = eng.loss_auc()
        assert loss == pytest.approx(float(out["loss"]), rel=tol["loss"])
        P = {k: v.detach() for k, v in P.items()}


@pytest.mark.parametrize("model,D,H,T,L", [("cast_1", 50, 1, 40, 2),      # BASELINE configs 1-2 shape class (D=50, 1 head): the bench workload
                                           ("sasrec", 64, 2, 50, 2),      # BASELINE config 3 shape (Beauty: D=64, 2 heads)
                                           ("cast_2", 64, 2, 24, 1),
                                           ("sasrec", 128, 4, 40, 2),     # config 4 shape (Books: D=128, 4 heads) -> unfused path
                                           ("cast_9", 128, 4, 24, 1),
                                           ("sasrec", 256, 4, 300, 1),    # config 5 shape class (D=256, maxlen > 256) -> general attention
                                           ("cast_1", 128, 1, 20, 1)])    # head dim 128 > 64 -> general attention
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_other_baseline_shapes_match_oracle(E, model, D, H, T, L, prec):
    _other_shapes(E, model, D, H, T, L, prec=prec)


@pytest.mark.parametrize("model,D,H,T,L,B", [("sasrec", 4, 1, 19, 1, 3),      # smallest hidden size the fused kernels take
                                             ("cast_1", 5, 1, 33, 2, 3),      # the 4-column chunk that crosses column D is rotated
                                             ("sasrec", 12, 3, 21, 1, 5),     # head dim 4
                                             ("cast_3", 13, 1, 24, 1, 3),
                                             ("cast_1", 33, 1, 50, 1, 2),
                                             ("sasrec", 47, 1, 70, 2, 2),
                                             ("cast_1", 8, 1, 40, 1, 3),      # hidden sizes the block backward has as constants: 8 ..
                                             ("sasrec", 24, 1, 100, 2, 2),
                                             ("cast_1", 56, 1, 200, 1, 2),    # .. 56 (key side: the tile's Q rows requested late)
                                             ("cast_8", 63, 3, 31, 1, 3),     # head dim 21, odd everything
                                             ("sasrec", 60, 4, 129, 1, 2),    # head dim 15, T = 8*16 + 1
                                             ("cast_1", 64, 1, 255, 1, 1),    # upper edge of the LDS-resident envelope
                                             ("sasrec", 6, 3, 17, 1, 2),      # head dim 2 < 4 -> general attention under fused blocks
                                             ("sasrec", 66, 1, 18, 1, 2)])    # first hidden size past the fused kernels
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_awkward_hidden_sizes_and_row_counts(E, model, D, H, T, L, B, prec):
    """Hidden sizes that are not multiples of 4 or 16, row counts B*T that are not multiples of the 16-row strips /
    64-row tiles, single-sequence batches: every tail path of the staging, MFMA and store code (head dims below 8 run
    the fp32 attention kernels under either setting)."""
    _other_shapes(E, model, D, H, T, L, B, prec=prec)


@pytest.mark.parametrize("env", ["CASTREC_NO_TAILS", "CASTREC_TWO_PASS_ATTN_BWD", "CASTREC_NO_EMBED_FUSION", "CASTREC_NO_HEAD_LN",
                                 "CASTREC_NO_STACK_KERNEL", "CASTREC_NO_STACK_BWD", "CASTREC_NO_LNF_FUSION", "CASTREC_NO_HEAD_DELTA",
                                 "CASTREC_NO_BLOCK_BWD", "CASTREC_NO_HEAD_FUSION", "CASTREC_NO_INDEX"])
def test_alternative_kernel_paths_stay_green(E, env, monkeypatch):
    """The plain FFN forward entry (no tail), the two-pass attention backward at one head and the stand-alone
    embedding gather in front of a stack: the engine's default path no longer uses them, the C ABI still offers them."""
    monkeypatch.setenv(env, "1")
    if env == "CASTREC_NO_HEAD_DELTA":                     # two heads at D = 64 with the two-launch attention backward
        _other_shapes(E, "sasrec", 64, 2, 50, 2, prec="bf16x3")
        return
    _other_shapes(E, "cast_1", 50, 1, 40, 2, prec="f32")
    _other_shapes(E, "cast_1", 50, 1, 40, 2, prec="bf16x3")
    if env == "CASTREC_NO_HEAD_FUSION":                    # ... at the engine's own slab count, where the fused head would have run
        eng = _other_shapes(E, "cast_1", 50, 1, 40, 2, prec="bf16x3", n_slabs=None)
        assert "cr_head_fwd_bwd_ln" in [n for n, _, _ in eng.fwd]


@pytest.mark.parametrize("model,D,H,T,B", [("cast_1", 50, 1, 40, 3), ("sasrec", 50, 1, 200, 5), ("sasrec", 64, 2, 50, 7), ("cast_3", 20, 1, 24, 4),
                                            ("sasrec", 36, 1, 104, 6), ("cast_1", 60, 1, 200, 3)])
def test_prediction_head_as_the_tail_of_the_last_forward_launch(E, model, D, H, T, B):
    """cr_stack_fwd_head (round 5): the head and the final LayerNorm's backward run on the rows of the trunk's last forward launch -- no
    cr_head_fwd_bwd_ln launch -- at the engine's own slab count (two slabs per sequence for the LayerNorm's gradient): loss, every
    gradient and the forward rows against the oracle, over the exact-size, family and generic instantiations."""
    eng = _other_shapes(E, model, D, H, T, 2, B=B, prec="bf16x3", n_slabs=None)
    names = [n for n, _, _ in eng.fwd]
    assert names[-1] == "cr_stack_fwd_head" and "cr_head_fwd_bwd_ln" not in names and eng.n_slabs >= 2 * B


@pytest.mark.parametrize("env", ["CASTREC_NO_TAILS", "CASTREC_NO_EMBED_FUSION", "CASTREC_NO_HEAD_LN"])
@pytest.mark.parametrize("model", ["cast_1", "sasrec"])
def test_fused_entries_give_the_results_of_their_separate_calls(E, env, model, monkeypatch):
    """include/castrec.h promises that the fused entries (FFN tails, stack input composed in the first block's kernels,
    final-LayerNorm backward inside the head) give the results of the calls they replace: the same engine built with
    and without them, same parameters, same batch, dropout on.  Same mathematics; the fused forms round differently in the
    last bit in places (e.g. the row mean as sum * (1/D) against sum / D in the final LayerNorm) and the table scatter's float atomics
    reorder, so the comparison is to 2e-6 of the tensor's scale -- two orders below the parity bound."""
    rs = np.random.RandomState(17)
    B, T, D, itemnum, max_bins = 6, 40, 50, 45, 9
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=max_bins, seed=13)
    # exact fp32 attention on both sides: the bf16 split form resolves 2^-17, so last-bit input differences between the
    # two builds would show at 1e-5 there, above what this comparison is about
    a = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=4, attn_precision="f32")
    monkeypatch.setenv(env, "1")
    b = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=4, attn_precision="f32")
    monkeypatch.delenv(env)
    assert a.n_launches() < b.n_launches()
    a.P.add_(0.05 * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    b.P.copy_(a.P)
    batch = make_batch(rs, B, T, itemnum, max_bins)
    for eng in (a, b):
        eng.set_batch(*batch)
        eng.launch_step(apply=False)
    torch.cuda.synchronize()
    assert float((a.seq_emb - b.seq_emb).abs().max()) <= 2e-6 * float(b.seq_emb.abs().max())
    sa, sb = a.state.cpu().numpy(), b.state.cpu().numpy()
    assert sa[2] == sb[2] and sa[0] == pytest.approx(sb[0], rel=1e-6)
    ga, gb = a.grads(), b.grads()
    gmax = max(float(v.abs().max()) for v in gb.values())
    for k in gb:
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-6 * gmax, k


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("env,T,D,B,H", [("CASTREC_NO_STACK_BWD", 40, 50, 6, 1), ("CASTREC_NO_STACK_BWD", 200, 50, 3, 1), ("CASTREC_NO_STACK_BWD", 33, 24, 5, 1),
                                         # the one-launch block backward (cr_stack_bwd1.hip) against the three launches it replaces
                                         ("CASTREC_NO_BLOCK_BWD", 40, 50, 6, 1), ("CASTREC_NO_BLOCK_BWD", 200, 50, 3, 1), ("CASTREC_NO_BLOCK_BWD", 33, 24, 5, 1),
                                         ("CASTREC_NO_BLOCK_BWD", 224, 50, 2, 1), ("CASTREC_NO_BLOCK_BWD", 100, 63, 9, 1),
                                         ("CASTREC_NO_STACK_KERNEL", 40, 50, 6, 1), ("CASTREC_NO_STACK_KERNEL", 200, 50, 3, 1), ("CASTREC_NO_STACK_KERNEL", 33, 24, 5, 1),
                                         ("CASTREC_NO_STACK_BWD", 50, 64, 5, 2), ("CASTREC_NO_STACK_KERNEL", 50, 64, 5, 2),     # config C3's shape: two heads
                                         ("CASTREC_NO_STACK_BWD", 100, 64, 3, 1), ("CASTREC_NO_STACK_KERNEL", 100, 64, 3, 1)])
def test_register_layout_kernels_equal_the_tile_kernels(monkeypatch, E, prec, env, T, D, B, H):
    """The whole-stack forward (cr_stack.hip) and the register-layout row-phase backward (cr_stack_bwd.hip) against the
    kernels they replace (cr_block_* + cr_attn_fwd): same parameters, same batch, dropout on, n_slabs below AND above the
    batch size.  Same mathematics; the projections / feed-forward / weight gradients run on split bf16 products here and on
    fp32 MFMA there, so the two differ by the rounding of those products: 1e-4 of the tensor scale for the split form
    (measured 2e-5); plain bf16 rounds every operand to 8 bits, at different places on the two paths: 3e-2 on the
    activations; on the gradients 3e-2 with the same forward (measured 1e-2) and 1e-1 with another forward (measured 4.5e-2).  Until
    round 5 the bound was 1e-1 throughout, which hid a defect: test_plain_bf16_weight_gradients_at_the_headline_length."""
    rs = np.random.RandomState(5)
    itemnum, max_bins = 45, 9
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=max_bins, seed=13)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True, n_slabs=4 if B > 4 else 7, attn_precision=prec)
    monkeypatch.setenv(env, "1")
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True, n_slabs=4 if B > 4 else 7, attn_precision=prec)
    monkeypatch.delenv(env)
    names = lambda e: [n for n, _, _ in e.fwd + e.bwd]
    new = {"CASTREC_NO_STACK_KERNEL": ("cr_stack_fwd",), "CASTREC_NO_BLOCK_BWD": ("cr_stack_block_bwd",),
           "CASTREC_NO_STACK_BWD": ("cr_stack_ffn_bwd", "cr_stack_ffn_bwd_heads", "cr_stack_block_bwd")}[env]
    assert any(n in names(a) for n in new) and not any(n in names(b) for n in new)
    if env == "CASTREC_NO_BLOCK_BWD":
        assert "cr_stack_ffn_bwd" in names(b) or "cr_stack_ffn_bwd_ln" in names(b)      # ... and b runs the three-launch form
    a.P.add_(0.05 * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    b.P.copy_(a.P)
    batch = make_batch(rs, B, T, itemnum, max_bins)
    for eng in (a, b):
        eng.set_batch(*batch)
        eng.launch_step(apply=False)
    torch.cuda.synchronize()
    tol = 1e-4 if prec == "bf16x3" else 3e-2
    assert float((a.seq_emb - b.seq_emb).abs().max()) <= tol * float(b.seq_emb.abs().max())
    sa, sb = a.state.cpu().numpy(), b.state.cpu().numpy()
    assert sa[2] == sb[2] and sa[0] == pytest.approx(sb[0], rel=tol)
    ga, gb = a.grads(), b.grads()
    gmax = max(float(v.abs().max()) for v in gb.values())
    for k in gb:
        # (plain bf16 with another FORWARD: the activations themselves differ at the 8-bit level and the backward inherits that, 4.5e-2
        #  measured; with the same forward and another backward: 1e-2 measured)
        assert float((ga[k] - gb[k]).abs().max()) <= (10 * tol if prec == "bf16x3" else 1e-1 if env == "CASTREC_NO_STACK_KERNEL" else 3e-2) * gmax, k


def test_plain_bf16_weight_gradients_at_the_headline_length(monkeypatch, E):
    """Round 5 found the plain-bf16 block backward's dW2 wrong by 20-50 % of its largest entry at T = 200 / D = 50, run-dependent, in
    registers 0 / 1 of the out-column tiles 0 and 2, from the rows of tiles 4 / 5 only: the odd seventh tile of a round went through
    v_mfma_f32_16x16x16_bf16 with, as SrcC, the result of the v_mfma_f32_16x16x32_bf16 of tiles 4 / 5 issued just before it (two
    products per pair in the plain build, six in the split one, which never showed it).  A 1e-1 bound on plain bf16 hid it.  Now one
    shape along the chain (cr_rbwd.hpp wgrad_accum; the build refuses listings with such a chain: tests/test_isa.py).  Here: every
    weight gradient of the one-launch block backward against the tile kernels (fp32 MFMA on the same stored activations), relative
    to the parameter's OWN largest entry, with all positions live and with only the rows of tiles 4 / 5 live, and the same bits twice."""
    T, D, B = 200, 50, 3
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=9, seed=13)
    a = E.Engine("sasrec", 9, 45, hp, B, training=True, n_slabs=7, attn_precision="bf16")
    monkeypatch.setenv("CASTREC_NO_STACK_BWD", "1")
    b = E.Engine("sasrec", 9, 45, hp, B, training=True, n_slabs=7, attn_precision="bf16")
    monkeypatch.delenv("CASTREC_NO_STACK_BWD")
    assert "cr_stack_block_bwd" in [n for n, _, _ in a.bwd] and "cr_stack_block_bwd" not in [n for n, _, _ in b.bwd]
    a.P.add_(0.05 * torch.randn(a.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
    b.P.copy_(a.P)
    for lo, hi in ((0, T), (64, 96)):
        rs = np.random.RandomState(5)
        batch = [x.copy() for x in make_batch(rs, B, T, 45, 9)]
        for x in batch[:3]:
            x[:, :lo] = 0
            x[:, hi:] = 0
            x[:, lo:hi] = np.maximum(x[:, lo:hi], 1)
        got = []
        for e in (a, a, b):
            e.set_batch(*batch)
            e.set_step(1)
            e.Gflat.zero_()
            e.launch_step(apply=False)
            torch.cuda.synchronize()
            got.append({k: v.clone() for k, v in e.grads().items()})
        for k in got[0]:
            if k.endswith((".w1", ".w2", ".wq", ".wk", ".wv")):
                assert torch.equal(got[0][k], got[1][k]), k
                err = float((got[0][k] - got[2][k]).abs().max() / got[2][k].abs().max())
                assert err < 3e-2, (lo, hi, k, err)               # (measured <= 1.1e-2; the defect: 0.2-0.86)


def test_plain_bf16_at_the_headline_length_against_the_oracle(E):
    """Plain bf16 (BASELINE configs[1] names bf16) at the bench workload's length and width -- cast_1, T 200, D 50 -- against the fp64
    oracle, every parameter's gradient relative to that parameter's OWN largest entry (the small-shape cases compare with the global
    gradient scale at 2e-1, which is how a 47 % error of one weight gradient at this length stayed unseen until round 5): measured
    <= 2.7e-2 over two parameter draws (tools/probes/bf16_vs_oracle.py), asserted 6e-2; loss 2e-3."""
    B, T, D, H = 4, 200, 50, 1
    for init in ("oracle", "engine"):
        rs = np.random.RandomState(5)
        hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3, seed=13)
        ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3)
        eng = E.Engine("cast_1", 9, 45, hp, B, training=True, n_slabs=7, attn_precision="bf16")
        assert "cr_stack_block_bwd" in [n for n, _, _ in eng.bwd]
        if init == "engine":
            eng.P.add_(0.05 * torch.randn(eng.P.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)))
        else:
            P = fm.init_params("cast_1", 9, 45, ohp, seed=8)
            eng.load_params({k: v + 0.05 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()})
        P = {k: v.double().cpu() for k, v in eng.get_params().items()}
        batch = make_batch(rs, B, T, 45, 9)
        eng.set_batch(*batch)
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        drop = oracle_drop(E, 13, 1, 0.2, B, T, H)
        out, G = oracle_with_engine_gates(eng, B, T, drop, "bf16", lambda: fm.loss_and_grads("cast_1", P, ohp, fm.to_batch(*batch), drop))
        st = eng.state.cpu().numpy()
        assert st[0] / st[2] == pytest.approx(float(out["loss"]), rel=2e-3)
        got = eng.grads()
        errs = sorted(((float((got[k].cpu().double() - G[k]).abs().max() / max(float(G[k].abs().max()), 1e-12)), k)
                       for k in G if not k.endswith(".bk")), reverse=True)
        assert errs[0][0] < 6e-2, (init, errs[:5])


@pytest.mark.parametrize("model", ["cast_1", "cast_3"])
def test_block_backward_rows_are_reproducible_at_the_headline_length(E, model):
    """Every activation-gradient buffer of a step (d_o, dQ / dK / dV, both partials of every block input, what the consumers
    make of them) and the dense slabs hold the same bits on 40 replays of the same step at T = 200.  Round 3 lost this twice
    (profiles/r04_flake/README.md: with the SLP vectoriser on, the last statement of the LayerNorm backward became in-place
    v_pk_fma_f32 chains, and one of them went missing in one register of lanes 48..63).  What keeps that form out is structural
    and checked without a GPU -- the statement is three scalar instructions in inline assembly (cr_common.hpp cr_ln_bwd_tail), the
    build scans the device code for the chain and refuses the flags that could bring it back (tests/test_isa.py); this test is the
    plain reproducibility check, no longer a 400-replay statistical guard."""
    rs = np.random.RandomState(250)
    B, T, D, itemnum = 3, 200, 50, 300
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.1, max_bins=200, num_context_blocks=1, lr=1e-3, seed=11)
    eng = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=5)
    assert "cr_stack_block_bwd" in [n for n, _, _ in eng.bwd]
    eng.P.add_(0.05 * torch.randn(eng.P.numel(), generator=torch.Generator().manual_seed(5)).to(eng.P.device))
    batch = make_batch(rs, B, T, itemnum, 200)
    ref = None
    for it in range(40):
        eng.set_batch(*batch)
        eng.set_step(1)
        eng.Gflat.zero_()
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        cur = {k: v for k, v in eng._bufs.items() if v.dtype == torch.float32}
        cur["Gs"] = eng.Gs
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            assert not [k for k, v in ref.items() if torch.isnan(v).any()]
        else:
            bad = [k for k in cur if not torch.equal(cur[k], ref[k])]
            assert not bad, (it, bad)


@pytest.mark.parametrize("B,n_slabs", [(6, 16), (9, 4)])
def test_block_backward_is_bitwise_reproducible(E, B, n_slabs):
    """cr_stack_block_bwd: slabs written in a fixed order by one workgroup pair per sequence (and added to, sequence after
    sequence, when there are fewer slabs than sequences), no atomics in the dense path: the same bits on every run."""
    rs = np.random.RandomState(3)
    T, D, itemnum = 100, 50, 41
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3, seed=11)
    eng = E.Engine("cast_1", 9, itemnum, hp, B, training=True, n_slabs=n_slabs)
    assert "cr_stack_block_bwd" in [n for n, _, _ in eng.bwd]
    batch = make_batch(rs, B, T, itemnum, 9)
    got = []
    for _ in range(3):
        eng.set_batch(*batch)
        eng.set_step(1)
        eng.Gflat.zero_()
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        got.append(eng.Gs.clone())
    assert float(got[0].abs().max()) > 0
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])


@pytest.mark.parametrize("model,D,T", [("cast_1", 50, 100), ("sasrec", 50, 100), ("sasrec", 64, 50), ("cast_3", 20, 24), ("sasrec", 128, 40)])
def test_a_step_is_bitwise_reproducible_with_the_occurrence_index(E, model, D, T):
    """Round 5: no float atomics are left in a step's gradients -- the item (and learned positional) table's rows are gathered from
    the batch's occurrence index in list order (castrec.h "occurrence index") where rounds 1-4 scattered float atomics from the head
    and the embedding backward.  Same batch, same state: the table gradient (Engine.grads: cr_table_grad) holds the same bits on
    every run, and two engines hold the same PARAMETER bits after three optimiser steps (cr_adam_step's in-place gather)."""
    rs = np.random.RandomState(31)
    B, itemnum = 7, 60                                    # few items: hot rows (hundreds of occurrences -- the heavy units) and rows no batch touches
    H = 2 if D in (64, 128) else 1
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3, seed=11)
    a = E.Engine(model, 9, itemnum, hp, B, training=True)
    b = E.Engine(model, 9, itemnum, hp, B, training=True)
    assert a.use_index and a.bitwise_reproducible and a._adam[2][0]._obj.tg
    names = [n for n, _, _ in a.fwd + a.bwd]
    assert not any(n.endswith("_scatter") for n in names)
    b.P.copy_(a.P)
    batches = [make_batch(rs, B, T, itemnum, 9) for _ in range(3)]
    tabs = []
    for _ in range(3):
        a.set_batch(*batches[0])
        a.set_step(1)
        a.launch_step(apply=False)
        torch.cuda.synchronize()
        g = a.grads()
        tabs.append(torch.cat([g[k].reshape(-1) for k in sorted(g) if k in ("item_emb", "pos_emb")] + [a.Gs.reshape(-1)]))
    assert float(tabs[0].abs().max()) > 0 and torch.equal(tabs[0], tabs[1]) and torch.equal(tabs[0], tabs[2])
    a.set_step(1)
    for bt in batches:
        a.train_step(*bt)
        b.train_step(*bt)
    torch.cuda.synchronize()
    assert torch.equal(a.P, b.P) and torch.equal(a.Mom, b.Mom) and torch.equal(a.Vel, b.Vel)
    assert bool((a.P != E.Engine(model, 9, itemnum, hp, B, training=False).P).any())


def test_index_and_atomics_paths_agree(E, monkeypatch):
    """CASTREC_NO_INDEX=1 brings the float-atomic scatters of rounds 1-4 back: same gradients (to rounding: another order of the
    sums), same training run."""
    rs = np.random.RandomState(33)
    B, T, D, itemnum = 6, 40, 50, 70
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=9, lr=1e-3, seed=11)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    monkeypatch.setenv("CASTREC_NO_INDEX", "1")
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    monkeypatch.delenv("CASTREC_NO_INDEX")
    assert a.use_index and not b.use_index and b._adam[2][0]._obj.tg is None or not b._adam[2][0]._obj.tg
    b.P.copy_(a.P)
    bt = make_batch(rs, B, T, itemnum, 9)
    for e in (a, b):
        e.set_batch(*bt); e.set_step(1); e.Gflat.zero_(); e.launch_step(apply=False)
    torch.cuda.synchronize()
    ga, gb = a.grads(), b.grads()
    for k in ga:                                          # (fp32 sums in another order: rounding, relative to the parameter's gradient scale)
        if k.endswith(".bk"):                             # (identically zero: rounding residue on both sides)
            continue
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-4 * max(1e-6, float(gb[k].abs().max())), k
    for e in (a, b):
        e.Gflat.zero_(); e.set_step(1)
    for _ in range(4):
        bt = make_batch(rs, B, T, itemnum, 9)
        a.train_step(*bt); b.train_step(*bt)
    torch.cuda.synchronize()
    assert same_run(a, b)


def _other_shapes(E, model, D, H, T, L, B=3, prec="f32", itemnum=41, max_bins=9, zipf=None, kink_free=False, n_slabs=5, dropout=0.1, grad_tol=None):
    """kink_free: the feed-forward pre-activations are pushed away from the ReLU kink (small W1, biases of +-1 alternating by
    unit), the oracle runs with ITS OWN gates (no hand-over), and the test first proves that the engine's gates are the
    same everywhere: gradient parity of the split-precision kernels with nothing taken from the engine but the dropout masks."""
    tol = TOL[prec]
    rs = np.random.RandomState(D + T)
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=L, num_heads=H, dropout_rate=dropout, max_bins=max_bins,
                 num_context_blocks=1, lr=1e-3, seed=11)
    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=L, num_heads=H, dropout_rate=dropout, max_bins=max_bins,
                   num_context_blocks=1, lr=1e-3)
    eng = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=n_slabs, attn_precision=prec)      # n_slabs None: the engine's own choice
    assert eng.fused == (D <= 64)
    P = fm.init_params(model, 9, itemnum, ohp, seed=8)
    P = {k: v + 0.05 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}
    if kink_free:
        assert not any(k.startswith("mlp.") for k in P)
        for k in P:
            if k.endswith(".w1"):
                P[k] = 0.1 * P[k]
            elif k.endswith(".b1"):
                P[k] = torch.where(torch.arange(P[k].numel()) % 2 == 0, 1.0, -1.0).to(P[k].dtype).reshape(P[k].shape)
    eng.load_params(P)
    P = {k: v.double().cpu() for k, v in eng.get_params().items()}
    seq, pos, neg, time, hours, days = make_batch(rs, B, T, itemnum, max_bins)
    if zipf:                                             # long-tail ids: a few hot rows take most of the scatter / head atomics
        w = 1.0 / np.arange(1, itemnum + 1, dtype=np.float64) ** zipf
        cdf = np.cumsum(w / w.sum())
        draw = lambda: (np.searchsorted(cdf, rs.random_sample((B, T))).clip(0, itemnum - 1) + 1)
        live = seq != 0
        seq, pos, neg = draw() * live, draw() * live, draw() * live
    eng.set_batch(seq, pos, neg, time, hours, days)
    eng.launch_step(apply=False)
    torch.cuda.synchronize()
    drop = oracle_drop(E, 11, 1, dropout, B, T, H)
    # split-precision attention perturbs activations by ~1e-5: the oracle takes the engine's ReLU gates like it takes its
    # dropout masks, so that units within 1e-5 of the kink do not flip on one side only (see fpmodel.RELU_GATES)
    run = lambda: fm.loss_and_grads(model, P, ohp, fm.to_batch(seq, pos, neg, time, hours, days), drop)
    if kink_free:
        gates, care = engine_relu_gates(eng, B, T, drop)
        with fm.handed_over_gates(gates, care, max_share=0.0, min_units=0) as audit:      # premise: not ONE gate differs ...
            run()
        assert audit and all(a[0] > 0 for a in audit.values())
        assert all(float(g.float().mean()) > 0.2 and float(g.float().mean()) < 0.8 for g in gates.values())   # ... on a real on / off mix
        out, G = run()                                                                    # the oracle on its own gates
    else:
        out, G = oracle_with_engine_gates(eng, B, T, drop, prec, run)
    st = eng.state.cpu().numpy()
    assert st[0] / st[2] == pytest.approx(float(out["loss"]), rel=tol["loss"])
    got = eng.grads()
    worst = worst_grad_error(got, G, prec)
    assert worst[0] < (2 * worst[2] if grad_tol is None else grad_tol), worst
    assert rel(eng.seq_emb, out["seq_emb"].reshape(B * T, -1)) < tol["act"]     # forward parity (north-star bound 1e-3)
    return eng


def test_config_c1_exact_shape(E):
    """BASELINE configs[0]: ml-1m SASRec maxlen=50 hidden_units=50 num_blocks=2 num_heads=1 batch=128."""
    _other_shapes(E, "sasrec", 50, 1, 50, 2, B=128, prec="f32", itemnum=3416)
    _other_shapes(E, "sasrec", 50, 1, 50, 2, B=128, prec="bf16x3", itemnum=3416)


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_config_c2_exact_shape(E, prec):
    """BASELINE configs[1], the bench workload itself: ml-1m CAST(1) maxlen=200 hidden_units=50 num_blocks=2 num_heads=1, batch 128,
    3 416 items, 200 time bins, dropout 0.2, the engine's own slab count (128 sequences on 256 workgroups: one launch per block,
    the PAIR mode of the forward, one slab per sequence pair in the backward) -- loss, every gradient and the forward rows against
    the fp64 oracle (models/cast_1.py:29-91)."""
    _other_shapes(E, "cast_1", 50, 1, 200, 2, B=128, prec=prec, itemnum=3416, max_bins=200, n_slabs=None, dropout=0.2)


def test_config_c3_long_tail_vocabulary(E):
    """BASELINE configs[2]: Beauty, hidden 64, 2 heads, maxlen 50, the 57 289-item vocabulary with Zipf-distributed ids
    (hot rows: the float-atomic scatters of the embedding backward and of the head see heavy collisions)."""
    _other_shapes(E, "sasrec", 64, 2, 50, 2, B=128, prec="bf16x3", itemnum=57289, zipf=1.1)


@pytest.mark.parametrize("model,D,H,T,B", [("cast_1", 50, 1, 200, 161),      # the headline kernels' non-PAIR forms (D = 50 constant, 13 tiles)
                                            ("sasrec", 40, 1, 104, 170),      # generic hidden size, 7 tiles (NKT = 8 instantiation)
                                            ("sasrec", 64, 2, 50, 163)])      # two heads of 32 columns
def test_stack_forward_with_one_workgroup_per_sequence(E, model, D, H, T, B):
    """More than 160 sequences: cr_stack_fwd gives a sequence ONE workgroup and runs all blocks in one launch (two tiles per
    wave) -- the template family every other model test (B <= 130) never reaches (`python main.py --batch_size 256` does)."""
    _other_shapes(E, model, D, H, T, 2, B=B, prec="bf16x3", itemnum=300)


@pytest.mark.parametrize("D,H,T", [(50, 1, 200),      # register-layout kernels (cr_stack*.hip)
                                   (128, 4, 40)])     # wide row kernels (cr_wide.hip)
def test_split_precision_gradients_without_gate_handover(E, D, H, T):
    """Gradient parity of the bf16x3 kernel families against the oracle's OWN ReLU gates."""
    _other_shapes(E, "sasrec", D, H, T, 2, B=3, prec="bf16x3", kink_free=True)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_config_c4_full_shape(E, prec):
    """BASELINE configs[3]: Books, maxlen=200 hidden_units=128 num_blocks=4 num_heads=4 (unfused row phases, head dim 32)."""
    _other_shapes(E, "sasrec", 128, 4, 200, 4, B=4, prec=prec, itemnum=500)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_config_c5_shape(E, prec):
    """BASELINE configs[4]: maxlen=512 hidden_units=256 (4 heads of 64), small batch: the chunked attention kernels
    (bf16x3) / the general kernels (f32) under the unfused row phases."""
    _other_shapes(E, "sasrec", 256, 4, 512, 2, B=2, prec=prec, itemnum=3000)


@pytest.mark.parametrize("model,D,H,T,L,B,prec", [("sasrec", 192, 3, 40, 1, 3, "bf16x3"),     # 4 waves per workgroup, 3 panels per weight
                                                    ("cast_1", 128, 2, 37, 2, 3, "bf16x3"),     # 111 rows: a ragged last row block
                                                    ("cast_1", 128, 2, 37, 2, 3, "bf16"),       # plain bf16 operands
                                                    ("sasrec", 256, 4, 50, 1, 5, "bf16"),
                                                    ("cast_4", 128, 4, 24, 1, 3, "bf16x3")])    # wide blocks + the CAST mlp
def test_wide_row_kernels(E, model, D, H, T, L, B, prec):
    """cr_wide_* (hidden sizes 128 / 192 / 256, one launch per row phase) against the oracle; n_slabs = 5 < row blocks, so the
    backward kernels walk several row blocks per workgroup and ADD to their slab from the second one on."""
    _other_shapes(E, model, D, H, T, L, B, prec=prec)


@pytest.mark.parametrize("D,H", [(128, 4), (256, 4)])
def test_wide_dense_gradients_are_bitwise_reproducible(E, D, H):
    """the slabs of the wide kernels (LayerNorm column sums folded in a fixed wave order, weight gradients one slab per
    workgroup, added to in row-block order) hold the same bits on every run of the same step"""
    rs = np.random.RandomState(3)
    B, T, itemnum = 6, 40, 41
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=9, num_context_blocks=1, lr=1e-3, seed=11)
    eng = E.Engine("sasrec", 9, itemnum, hp, B, training=True, n_slabs=2)          # 240 rows on 2 slabs: several row blocks per workgroup
    seq, pos, neg, time, hours, days = make_batch(rs, B, T, itemnum, 9)
    got = []
    for _ in range(3):
        eng.set_batch(seq, pos, neg, time, hours, days)
        eng.set_step(1)
        eng.Gflat.zero_()
        eng.launch_step(apply=False)
        torch.cuda.synchronize()
        got.append(eng.Gs.clone())
    assert float(got[0].abs().max()) > 0
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])


@pytest.mark.parametrize("env", ["CASTREC_NO_WIDE", "CASTREC_WIDE_NO_WGRAD", "CASTREC_WIDE_NO_DELTA", "CASTREC_WIDE_NO_TAILS"])
def test_wide_alternative_paths_stay_green(E, env, monkeypatch):
    """the unfused chain (cr_layernorm_* / cr_gemm_rows / cr_eltwise) and the wide kernels with cr_gemm_wgrad forming the
    weight gradients: what the engine falls back to outside D = 128 / 192 / 256; the two-launch attention backward (no
    per-head delta from the FFN backward)"""
    monkeypatch.setenv(env, "1")
    _other_shapes(E, "sasrec", 128, 4, 40, 2, prec="bf16x3")


@pytest.mark.parametrize("model", ["cast_%d" % i for i in range(2, 10)])
def test_every_cast_graph_at_the_headline_shape(E, model):
    """cast_2 ... cast_9 once at D = 50, T = 200 (cast_1 is the bench workload and has its own tests)."""
    _other_shapes(E, model, 50, 1, 200, 2, B=3, prec="bf16x3", itemnum=300, max_bins=200)


@pytest.mark.gpu
@pytest.mark.parametrize("model,D,H,T,max_bins", [("cast_1", 50, 1, 200, 255), ("cast_1", 64, 2, 40, 255), ("cast_1", 40, 1, 72, 16),
                                                    ("cast_1", 48, 1, 120, 100), ("cast_1", 16, 1, 24, 9)])
def test_small_tables_through_the_one_hot_product(E, model, D, H, T, max_bins):
    """cr_stack_block_bwd's small-table backward = one-hot(ids)^T x rows on the matrix pipe (round 4): tables of 256 rows (all
    sixteen table-row tiles, both tiles of every wave), of fewer rows than one tile, the two-head instantiation at D = 64, an odd
    and an even count of row tiles, the generic and the constant hidden sizes.  Gradients of the table against the oracle like every
    other parameter (the product takes the rows as bf16 hi + lo: inside the tolerance of the arithmetic)."""
    eng = _other_shapes(E, model, D, H, T, 1, B=3, prec="bf16x3", max_bins=max_bins, n_slabs=8)
    names = [n for n, _, _ in eng.bwd]
    # the time table's backward is inside the context stack's block backward (two slabs per sequence pair: 8 >= 2 x 3)
    assert names.count("cr_stack_block_bwd") == 2 and "cr_embed_bwd" not in names, names


def test_context_tables_too_large_for_one_lds_image(E):
    """time_emb at --max_bins 200 and hidden 64 (one head: not a shape of the one-launch block backward) is 201 x 64 floats > the
    12 288 cr_embed_bwd's small-table mode keeps in LDS: it must fall back to the large-table scatter (same slab contract), fused and
    unfused (D = 128) alike."""
    _other_shapes(E, "cast_1", 64, 1, 40, 1, B=3, prec="f32", max_bins=200)
    _other_shapes(E, "cast_3", 128, 2, 24, 1, B=3, prec="f32", max_bins=200)


def test_id_range_is_checked(E):
    hp = E.Hyper(maxlen=8, hidden_units=16, num_blocks=1, num_heads=1, dropout_rate=0.0, max_bins=5, seed=1)
    eng = E.Engine("cast_1", 9, 20, hp, 2, training=False)
    ok = np.ones((2, 8), np.int64)
    eng.set_batch(ok, None, None, ok, ok, ok)
    with pytest.raises(ValueError, match="time ids outside"):
        eng.set_batch(ok, None, None, ok * 6, ok, ok)               # bin 6 in a 6-row table
    with pytest.raises(ValueError, match="seq ids outside"):
        eng.set_batch(ok * 21)
    with pytest.raises(ValueError, match="time ids outside"):
        eng.set_batch(ok, None, None, -ok, ok, ok)                  # unsorted timestamps give negative bins


@pytest.mark.parametrize("model", ["sasrec", "cast_1", "cast_5", "cast_8"])
def test_eval_logits_and_attention_weights(E, model):
    rs = np.random.RandomState(5)
    B, T, D, H, itemnum, max_bins = 4, 16, 10, 1, 30, 8
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.4, max_bins=max_bins, seed=1)
    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.4, max_bins=max_bins)
    eng = E.Engine(model, 9, itemnum, hp, B, training=False, want_attn=True)
    P = fm.init_params(model, 9, itemnum, ohp, seed=4)
    P = {k: v + 0.1 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}
    eng.load_params(P)
    P = {k: v.double().cpu() for k, v in eng.get_params().items()}
    seq, pos, neg, time, hours, days = make_batch(rs, B, T, itemnum, max_bins)
    cand = rs.randint(1, itemnum + 1, 101)
    batch = fm.to_batch(seq, pos, neg, time, hours, days, test_item=cand)
    out = fm.forward(model, P, ohp, batch, drop=None)              # is_training False
    eng.forward_eval(seq, time, hours, days)
    cd = torch.tensor(np.tile(cand, (B, 1)).astype(np.int32)).cuda()
    lg = eng.test_logits(cd)
    torch.cuda.synchronize()
    assert rel(lg, out["test_logits"]) < 1e-4                      # north-star bound: 1e-3
    assert rel(eng.attn_weights, out["attention_weights"]) < 1e-4


def test_graph_replay_equals_eager(E):
    rs = np.random.RandomState(6)
    B, T, D, itemnum = 8, 32, 50, 100
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=3)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b.P.copy_(a.P)
    b.capture()
    batch = make_batch(rs, B, T, itemnum, 20)
    a.train_step(*batch)
    b.train_step(*batch)
    torch.cuda.synchronize()
    nt = a.layout.n_table
    assert torch.equal(a.P[nt:], b.P[nt:])                         # dense part: slab reduction, bit-identical
    assert torch.allclose(a.P[:nt], b.P[:nt], rtol=0, atol=1e-6)   # item table: float atomics may reorder
    assert a.loss_auc() == pytest.approx(b.loss_auc(), rel=1e-6)   # loss sums use float atomics across workgroups
    for _ in range(3):                                             # replay keeps tracking eager training
        batch = make_batch(rs, B, T, itemnum, 20)
        a.train_step(*batch)
        b.train_step(*batch)
    torch.cuda.synchronize()
    la, lb = a.loss_auc(), b.loss_auc()
    assert la[0] == pytest.approx(lb[0], rel=1e-3) and b.step_number() == 5


def same_loss_auc(a, b, pos, rel=2e-3):
    """Loss to `rel`; the AUC (the fraction of target rows with pos logit > neg logit, sasrec.py:113-115) of two runs whose table
    gradients were summed by float atomics in different orders may differ by ONE row whose two logits all but tie: 1 / n_targets."""
    (la, aa), (lb, ab) = a.loss_auc(), b.loss_auc()
    n_t = max(int((pos != 0).sum()), 1)
    return la == pytest.approx(lb, rel=rel) and abs(aa - ab) <= 1.01 / n_t + rel * abs(ab)



@pytest.mark.parametrize("lazy,graph", [(False, False), (False, True), (True, True), (None, True)])
def test_id_ring_feeds_the_batches_a_copy_per_step_would(E, lazy, graph):
    """Engine.use_id_ring: the batch of step k sits in ring slot k mod n_slots and the step before moves it into the static id
    buffers (extra workgroups of the Adam launch; cr_ids_ring_next behind a row-sparse Adam).  Same batches, same order =
    the run that copies one batch per step."""
    rs = np.random.RandomState(16)
    B, T, D, itemnum, NS = (6, 32, 50, 90, 3) if lazy is False else (5, 31, 50, 90, 3)      # (5 x 31 rows: a slot that is no multiple of 16 bytes)
    lazy = bool(lazy)
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=5)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True, lazy_adam=lazy)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True, lazy_adam=lazy)
    b.P.copy_(a.P)
    batches = [make_batch(rs, B, T, itemnum, 20) for _ in range(NS)]
    ring = torch.from_numpy(np.stack([b.pack_slot(*bt) for bt in batches])).cuda()          # a slot: the six id rows (+ the batch's occurrence index)
    b.use_id_ring(ring)
    if graph:
        b.capture()
    first = b.step_number()
    b.set_batch(*batches[first % NS])
    for k in range(first, first + 2 * NS + 1):                      # more steps than slots: the ring wraps
        a.train_step(*batches[k % NS])
        if graph:
            b.graph.launch()
        else:
            b.launch_step()
    torch.cuda.synchronize()
    assert b.step_number() == first + 2 * NS + 1
    assert torch.equal(b.ids_all.cpu().reshape(-1), ring[b.step_number() % NS][:6 * B * T].cpu())       # the coming step's batch is in place
    nt = a.layout.n_table
    assert same_run(a, b) and same_loss_auc(a, b, batches[(first + 2 * NS) % NS][1])
    b.P.copy_(a.P); b.Mom.copy_(a.Mom); b.Vel.copy_(a.Vel)
    b.use_id_ring(None)                                             # ... and back to a batch per call
    batch = make_batch(rs, B, T, itemnum, 20)
    if graph:
        b.capture()
    a.train_step(*batch)
    b.train_step(*batch)
    torch.cuda.synchronize()
    assert torch.equal(b.ids_all.cpu(), a.ids_all.cpu()) and torch.equal(a.P[nt:], b.P[nt:])


def test_several_fed_steps_per_graph_launch(E):
    """Engine.capture(n_steps = 4) / enable_feed(steps_per_graph = 4): train_fed() runs four steps in ONE graph launch whenever four
    batches wait (each step's tail moves its successor's batch on the device), one step otherwise or where max_steps says so -- the
    same batches in the same order as a step per call, whatever the pattern."""
    rs = np.random.RandomState(19)
    B, T, D, itemnum = 5, 32, 50, 80
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=6)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b.P.copy_(a.P)
    b.capture()
    b.enable_feed(n_slots=16, steps_per_graph=4)
    assert b.graph_multi is not None and b.graph_steps == 4
    batches = [make_batch(rs, B, T, itemnum, 20) for _ in range(14)]       # (few steps: float-atomic table gradients let runs drift apart)
    for bt in batches:
        a.train_step(*bt)
    it = iter(batches)
    ran = []
    for _ in range(5):
        b.feed(*next(it))
    ran.append(b.train_fed())                                      # five wait: four steps, the fifth batch moved by the last one's tail
    ran.append(b.train_fed())                                      # one waits: one step
    for _ in range(4):
        b.feed(*next(it))
    ran.append(b.train_fed())                                      # exactly four wait: four steps, nobody moves a successor
    for _ in range(5):
        b.feed(*next(it))
    ran.append(b.train_fed(max_steps=3))                           # (an epoch's end in three steps): one step
    ran.append(b.train_fed())                                      # four wait: four steps in one launch again
    torch.cuda.synchronize()
    assert ran == [4, 1, 4, 1, 4] and next(it, None) is None and b.step_number() == a.step_number() == 15
    assert same_run(a, b) and same_loss_auc(a, b, batches[-1][1])


def test_deep_feeding_around_multi_step_launches_keeps_slot_reuse_ordered(E):
    """Twelve batches fed ahead in the 16-slot ring (the Engine API's bound; main.py keeps five), single steps and four-step launches
    mixed so that the ring wraps: every step number that is a multiple of four carries an event (a four-step launch files its
    end-of-launch event under each such step it covers), so feed() always finds a recorded step at or behind a slot's last reader
    before it overwrites the slot -- and the run equals a step per call."""
    rs = np.random.RandomState(23)
    B, T, D, itemnum = 5, 32, 50, 80
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=6)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b.P.copy_(a.P)
    b.capture()
    b.enable_feed(n_slots=16, steps_per_graph=4)
    batches = [make_batch(rs, B, T, itemnum, 20) for _ in range(26)]      # (float-atomic table gradients let longer runs drift: same_run)
    for bt in batches:
        a.train_step(*bt)
    it = iter(batches)
    fed = ran = 0

    def top_up(n):
        nonlocal fed
        while fed - ran < n and fed < len(batches):
            b.feed(*next(it)); fed += 1
    pattern = [1, 1, 1, 4, 1, 1, 1, 4, 4, 1, 4]                               # singles in front of multi-step launches, as ADVICE round 4 drew it
    for want in pattern:
        top_up(12)
        got = b.train_fed(max_steps=None if want == 4 else 1)
        assert got == want, (got, want)
        ran += got
        marks = [j for j in b._feed_done]
        assert all(j % 4 == 0 or b._feed_done[j] is not None for j in marks)
        # every window of four steps behind the last launched one holds a recorded step
        last = b._feed_next - b._feed_have - 1
        assert any(j in b._feed_done for j in range(max(last - 3, b._feed_first), last + 1)) or last < b._feed_first + 3, (last, sorted(b._feed_done))
    while ran < len(batches):
        top_up(12)
        ran += b.train_fed()
    torch.cuda.synchronize()
    assert ran == len(batches) and b.step_number() == a.step_number()
    assert same_run(a, b) and same_loss_auc(a, b, batches[-1][1])


@pytest.mark.parametrize("graph,n_slots", [(False, 4), (True, 4), (True, 8)])
def test_fed_batches_train_like_batches_set_per_step(E, graph, n_slots):
    """Engine.enable_feed / feed / train_fed: pinned host batches sent ahead over a copy stream into the id ring.  Every
    feeding pattern (one ahead, two ahead, none ahead) runs the batches in order; an out-of-range id is refused."""
    rs = np.random.RandomState(17)
    B, T, D, itemnum = 5, 32, 50, 80
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=1, dropout_rate=0.2, max_bins=20, seed=6)
    a = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b = E.Engine("cast_1", 9, itemnum, hp, B, training=True)
    b.P.copy_(a.P)
    if graph:
        b.capture()
    b.enable_feed(n_slots=n_slots)
    batches = [make_batch(rs, B, T, itemnum, 20) for _ in range(12)]
    for bt in batches:
        a.train_step(*bt)
    it = iter(batches)
    b.feed(*next(it))
    for _ in range(4):                                             # one batch ahead
        b.feed(*next(it))
        b.train_fed()
    b.train_fed()                                                  # the queue runs dry: the next batch takes the direct way
    b.feed(*next(it)); b.feed(*next(it))                           # two ahead
    for _ in range(3):
        b.feed(*next(it))
        b.train_fed()
    b.train_fed(); b.train_fed()
    b.feed(*next(it)); b.train_fed()                               # none ahead
    b.feed(*next(it)); b.train_fed()
    torch.cuda.synchronize()
    assert next(it, None) is None and b.step_number() == a.step_number() == 13
    assert same_run(a, b) and same_loss_auc(a, b, batches[-1][1])
    bad = list(batches[0]); bad[0] = bad[0].copy(); bad[0][1, -1] = itemnum + 1
    with pytest.raises(ValueError):
        b.feed(*bad)
    for _ in range(5 if n_slots == 8 else 3):
        b.feed(*batches[0])
    with pytest.raises(RuntimeError):                              # no more than 5 (ring of 8) / n_slots - 1 batches wait
        b.feed(*batches[0])


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_trained_reference_weights_at_the_headline_shape(E, prec):
    """The TRAINED variables of the reference's cast_1 ml-1m run (tests/golden/cast_1_ml1m_weights.npz, extracted
    from its TensorFlow checkpoint by castrec_amd/tf_bundle.py): eval logits, loss and every gradient against the
    oracle at T=200, D=50, one head -- the bench workload's exact shape, with real weight statistics."""
    import os
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cast_1_ml1m_weights.npz"))
    rs = np.random.RandomState(11)
    B, T, D, H, itemnum, max_bins = 6, 200, 50, 1, 3416, 200
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=max_bins, seed=5)
    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=max_bins)
    P = {k: torch.tensor(w[k]) for k in w.files}
    seq = rs.randint(1, itemnum + 1, (B, T)); pos = rs.randint(1, itemnum + 1, (B, T)); neg = rs.randint(1, itemnum + 1, (B, T))
    for b, n in enumerate([0, 3, 60, 150, 190, 199]):                    # ragged left padding, one nearly empty sequence
        seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0
    time = np.minimum(rs.randint(0, 40, (B, T)).cumsum(1)[:, ::-1] // 8, max_bins) * (seq != 0)   # non-increasing bins, last = 0
    zeros = np.zeros_like(seq)
    # inference graph
    tol = TOL[prec]
    ev = E.Engine("cast_1", 6040, itemnum, hp, B, training=False, attn_precision=prec)
    ev.load_params(P)
    Pd = {k: v.double().cpu() for k, v in ev.get_params().items()}
    cand = rs.randint(1, itemnum + 1, 101)
    out = fm.forward("cast_1", Pd, ohp, fm.to_batch(seq, pos, neg, time, zeros, zeros, test_item=cand), drop=None)
    ev.forward_eval(seq, time, zeros, zeros)
    lg = ev.test_logits(torch.tensor(np.tile(cand, (B, 1)).astype(np.int32)).cuda())
    torch.cuda.synchronize()
    assert rel(lg, out["test_logits"]) < (1e-4 if prec == "f32" else 3e-4)   # north-star bound: 1e-3
    # training step with dropout (oracle fed with the engine's masks): loss and every gradient
    eng = E.Engine("cast_1", 6040, itemnum, hp, B, training=True, n_slabs=5, attn_precision=prec)
    eng.load_params(P)
    eng.set_batch(seq, pos, neg, time, zeros, zeros)
    eng.launch_step(apply=False)
    torch.cuda.synchronize()
    drop = oracle_drop(E, 5, 1, 0.2, B, T, H)
    o2, G = oracle_with_engine_gates(eng, B, T, drop, prec,
                                     lambda: fm.loss_and_grads("cast_1", Pd, ohp, fm.to_batch(seq, pos, neg, time, zeros, zeros), drop))
    st = eng.state.cpu().numpy()
    assert st[0] / st[2] == pytest.approx(float(o2["loss"]), rel=tol["loss"])
    got = eng.grads()
    worst = worst_grad_error(got, G, prec)
    assert worst[0] < 2 * worst[2], worst


@pytest.mark.parametrize("model", ["sasrec", "cast_3"])
def test_l2_emb_regulariser_matches_oracle(E, model):
    """--l2_emb (modules.py:149-153, sasrec.py:109-110): l2 * sum(w^2) / 2 over every lookup-table VARIABLE (row 0
    included) is added to the loss, i.e. l2 * w to the gradients.  Rows a batch never touches then still move (by
    lr * sign(w) on the first Adam steps), which is what tells a missing term apart: three steps against the oracle."""
    rs = np.random.RandomState(3)
    B, T, D, H, itemnum, max_bins, l2 = 4, 12, 16, 1, 200, 6, 0.05
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=1, num_heads=H, dropout_rate=0.0, max_bins=max_bins, lr=1e-3, seed=2, l2_emb=l2)
    ohp = fm.Hyper(maxlen=T, hidden_units=D, num_blocks=1, num_heads=H, dropout_rate=0.0, max_bins=max_bins, lr=1e-3, l2_emb=l2)
    eng = E.Engine(model, 9, itemnum, hp, B, training=True, n_slabs=3, attn_precision="f32")
    P = fm.init_params(model, 9, itemnum, ohp, seed=5)
    P = {k: v + 0.05 * torch.tensor(rs.standard_normal(tuple(v.shape))) for k, v in P.items()}
    eng.load_params(P)
    P = {k: v.double().cpu() for k, v in eng.get_params().items()}
    batch_np = make_batch(rs, B, T, itemnum, max_bins)
    batch = fm.to_batch(*batch_np)
    opt = fm.AdamTF(P, lr=1e-3)
    for step in (1, 2, 3):
        out, G = fm.loss_and_grads(model, P, ohp, batch)
        P = opt.step(P, G)
        eng.train_step(*batch_np)
        torch.cuda.synchronize()
        loss, _ = eng.loss_auc()
        assert loss == pytest.approx(float(out["loss"]), rel=2e-5)           # the penalty is part of the reported loss
    now = eng.get_params()
    untouched = np.setdiff1d(np.arange(1, itemnum + 1), np.unique(np.concatenate([a.reshape(-1) for a in batch_np[:3]])))
    assert len(untouched) > 50
    for k in ("item_emb",) + (("time_emb", "hours_emb", "days_emb") if model == "cast_3" else ("pos_emb",)):
        assert float((now[k].cpu().double() - P[k]).abs().max()) < 2e-5, k          # incl. rows the batch never touched
