"""The occurrence index of a batch (include/castrec.h "occurrence index"; csrc/cr_index.cpp, cr_tgrad.hpp): the host builder against a
plain restatement (no GPU), and -- on the GPU -- the gather it drives against the scatter it replaces (autodiff of modules.py:157
through sasrec.py:27, of sasrec.py:89-90 and of the learned positional lookup sasrec.py:40-50)."""
import ctypes as C

import numpy as np
import pytest

import castrec_amd  # noqa: F401
from castrec_amd import lib as L


def build_index(M, V, T_pos, seq, pos, neg, times=1, ng=32, ent=16):
    lay = L.IndexLayout()
    L.check(L.lib.cr_batch_index_layout(M, V, T_pos, ng, ent, C.byref(lay)), "layout")
    b = L.lib.cr_index_builder_create(M, V, T_pos, ng, ent)
    assert b
    out = np.full(lay.total_words, -7, np.int32)
    for _ in range(times):                                # (a builder's work arrays are clean again after every build)
        L.check(L.lib.cr_index_build(b, seq.ctypes.data, pos.ctypes.data, neg.ctypes.data, out.ctypes.data), "build")
    L.lib.cr_index_builder_destroy(b)
    return lay, out


def reference_lists(M, V, T_pos, seq, pos, neg):
    """flat row -> its occurrence words, in the order the header fixes"""
    ref = {}
    for k, ids in enumerate((seq, pos, neg)):
        for m in range(M):
            if ids[m]:
                ref.setdefault(int(ids[m]), []).append((k << 30) | m)
    for t in range(T_pos):
        ref[V + t] = [(3 << 30) | (b * T_pos + t) for b in range(M // T_pos)]
    return ref


def decode_plan(lay, out):
    """The plan as the device walks it: flat row -> its occurrence words in summation order (groups in q order, slices in j order);
    every structural promise of the header is asserted on the way."""
    nb, nrows, nocc, magic, used, off_occ, off_bits, z = (int(x) for x in out[:8])
    ng, ent = lay.ng, lay.ent
    assert magic == 0x43524959 and z == 0 and nb <= lay.cap_blocks and used <= lay.total_words
    assert off_occ == (8 + 4 * ng * nb + 3) // 4 * 4 and off_bits == off_occ + (nocc + 3) // 4 * 4 and used == off_bits + (lay.bitmap_words + 3) // 4 * 4
    recs = out[8:8 + 4 * ng * nb].reshape(nb, ng, 4)
    occ = out[off_occ:off_occ + nocc]
    got, slices = {}, {}
    for w in range(nb):
        g = 0
        while g < ng:
            row, st, c, info = (int(x) for x in recs[w, g])
            if c == 0:
                assert row == 0 and st == 0 and info == 0
                g += 1
                continue
            q, k, j, n = info & 63, (info >> 6) & 127, (info >> 13) & 511, (info >> 22) & 1023
            assert q == 0 and 1 <= k <= ng - g and n >= 1 and j < n
            words = []
            for qq in range(k):                           # the row's k consecutive groups
                r2, s2, c2, i2 = (int(x) for x in recs[w, g + qq])
                assert r2 == row and 0 <= c2 <= ent and (i2 & 63) == qq and ((i2 >> 6) & 127) == k and (i2 >> 13) == (info >> 13)
                assert s2 == st + len(words)
                words += [int(x) & 0xffffffff for x in occ[s2:s2 + c2]]
            assert len(words) > 0
            if n == 1:
                assert row not in got and row not in slices
                got[row] = words
            else:
                assert g == 0 and all(int(recs[w, gg, 2]) == 0 for gg in range(k, ng))       # a slice has its workgroup to itself
                slices.setdefault(row, {})[j] = (w, n, words)
            g += k
    for row, sl in slices.items():
        n = next(iter(sl.values()))[1]
        assert sorted(sl) == list(range(n)) and [sl[j][0] for j in range(n)] == list(range(sl[0][0], sl[0][0] + n))   # consecutive workgroups
        assert row not in got
        got[row] = sum((sl[j][2] for j in range(n)), [])
    bits = out[off_bits:off_bits + lay.bitmap_words].view(np.uint32)
    assert nrows == len(got) and nocc == sum(len(v) for v in got.values())
    return got, bits, nb


@pytest.mark.parametrize("M,V,T_pos,pad,seed,ng,ent", [(3200, 60, 200, 50, 0, 32, 16), (96, 5000, 0, 10, 1, 64, 8), (64, 2, 0, 0, 2, 16, 16),
                                                       (640, 40, 0, 640, 3, 32, 16), (512, 300, 64, 100, 4, 64, 8), (4096, 7, 0, 0, 5, 16, 16),
                                                       (2048, 3, 0, 0, 6, 64, 8)])
def test_host_builder_equals_the_restatement(M, V, T_pos, pad, seed, ng, ent):
    """Every listed row's plan sums exactly its occurrences in the fixed order (seq by ascending m, then pos, then neg; a positional row
    its B rows): at most `ent` per lane group, a row's groups consecutive in one workgroup, a row beyond a workgroup in slices of
    whole consecutive workgroups; the bitmap marks exactly the listed rows; a second build on the same builder gives the same words."""
    rs = np.random.RandomState(seed)
    seq, pos, neg = (rs.randint(0, V, M).astype(np.int32) for _ in range(3))
    seq[:pad] = 0; pos[:pad] = 0; neg[:pad] = 0           # left padding (all ids 0: nothing listed)
    lay, out = build_index(M, V, T_pos, seq, pos, neg, ng=ng, ent=ent)
    lay2, out2 = build_index(M, V, T_pos, seq, pos, neg, times=3, ng=ng, ent=ent)
    used = int(out[4])
    assert np.array_equal(out[:used], out2[:used]) and (out[:used] != -7).all()
    got, bits, nb = decode_plan(lay, out)
    ref = reference_lists(M, V, T_pos, seq, pos, neg)
    assert got == ref
    assert [r for r in range(V + T_pos) if (bits[r >> 5] >> (r & 31)) & 1] == sorted(ref)
    assert 0 not in ref                                   # the zero-pad row has no gradient (modules.py:154-156)
    groups = sum(-(-len(v) // ent) for v in ref.values())
    assert nb <= 2 * (-(-groups // ng)) + 2               # the packing: more than half full (multi-group rows), full otherwise


def test_an_id_outside_the_table_is_refused_and_corrupts_nothing():
    M, V = 32, 10
    seq = np.arange(M, dtype=np.int32) % V
    bad = seq.copy(); bad[5] = V
    lay = L.IndexLayout()
    L.check(L.lib.cr_batch_index_layout(M, V, 0, 32, 16, C.byref(lay)), "layout")
    b = L.lib.cr_index_builder_create(M, V, 0, 32, 16)
    out = np.zeros(lay.total_words, np.int32)
    assert L.lib.cr_index_build(b, bad.ctypes.data, seq.ctypes.data, seq.ctypes.data, out.ctypes.data) == -1
    assert b"outside" in L.lib.cr_last_error()
    good = np.zeros(lay.total_words, np.int32)
    assert L.lib.cr_index_build(b, seq.ctypes.data, seq.ctypes.data, seq.ctypes.data, good.ctypes.data) == 0
    L.lib.cr_index_builder_destroy(b)
    _, want = build_index(M, V, 0, seq, seq, seq)
    assert np.array_equal(good[:good[4]], want[:want[4]])


def test_geometry_of_the_gather():
    ng, ent = C.c_int(), C.c_int()
    for D, want in ((50, (32, 16)), (64, (64, 8)), (128, (32, 8)), (256, (16, 8)), (20, (64, 8)), (33, (16, 16)), (6, (64, 16)), (2, (64, 16))):
        assert L.lib.cr_tgrad_geometry(D, C.byref(ng), C.byref(ent)) == 1 and (ng.value, ent.value) == want, D
    for D in (0, 65, 67, 130, 260, 512):
        assert L.lib.cr_tgrad_geometry(D, C.byref(ng), C.byref(ent)) == 0, D


# ---- GPU: the gather against the scatter -------------------------------------------------------------------
def _table_grad_reference(V, T_pos, D, seq, pos, neg, rows, rows2, scale, emb, coef):
    """fp64: what float atomics summed in rounds 1-4"""
    M = len(seq)
    g = np.zeros((V + T_pos, D))
    r = rows.astype(np.float64) + (rows2.astype(np.float64) if rows2 is not None else 0.0)
    np.add.at(g, seq, scale * r)
    np.add.at(g, pos, coef[0][:, None].astype(np.float64) * emb)
    np.add.at(g, neg, coef[1][:, None].astype(np.float64) * emb)
    g[0] = 0.0
    if T_pos:
        g[V:] = r.reshape(M // T_pos, T_pos, D).sum(0)
    return g


@pytest.mark.gpu
@pytest.mark.parametrize("D,V,T_pos,two", [(50, 90, 0, True), (50, 90, 25, False), (64, 3000, 0, True), (20, 12, 25, True), (33, 40, 0, False),
                                           (128, 500, 0, False), (256, 700, 25, True), (6, 9, 0, True), (24, 2, 0, False)])
def test_table_grad_gather_equals_the_scatter(D, V, T_pos, two):
    """cr_table_grad over every (lanes per row, vector width) form: hot rows (thousands of occurrences: the heavy path), cold rows, a
    learned positional table, one or two row partials -- against the fp64 scatter; twice: the same bits (no atomics)."""
    import torch
    rs = np.random.RandomState(D + V)
    B, T = 40, 25
    M = B * T
    p = np.r_[0.0, 1.0 / np.arange(1, V) ** 1.2] if V > 2 else np.r_[0.0, 1.0]
    p /= p.sum()
    seq, pos, neg = (rs.choice(V, M, p=p).astype(np.int32) for _ in range(3))
    for b in range(B):
        n = rs.randint(0, T - 2)
        seq[b * T:b * T + n] = 0; pos[b * T:b * T + n] = 0; neg[b * T:b * T + n] = 0
    ng, ent = C.c_int(), C.c_int()
    assert L.lib.cr_tgrad_geometry(D, C.byref(ng), C.byref(ent))
    lay, ix = build_index(M, V, T_pos, seq, pos, neg, ng=ng.value, ent=ent.value)
    rows = rs.standard_normal((M, D)).astype(np.float32); rows2 = rs.standard_normal((M, D)).astype(np.float32) if two else None
    emb = rs.standard_normal((M, D)).astype(np.float32); coef = rs.standard_normal((2, M)).astype(np.float32)
    scale = float(np.sqrt(D))
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_ix, d_rows, d_emb, d_coef = t(ix), t(rows), t(emb), t(coef)
    d_rows2 = t(rows2) if two else None
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    part = torch.full((lay.cap_blocks, (D + 3) // 4 * 4), float("nan"), device=dev)
    tickets = torch.zeros(lay.cap_blocks, dtype=torch.int32, device=dev)
    g = L.TgradDesc(d_ix.data_ptr(), None, 0, 0, 0, step.data_ptr(), lay, d_rows.data_ptr(), d_rows2.data_ptr() if two else None, D, scale,
                    d_emb.data_ptr(), D, d_coef.data_ptr(), D, part.data_ptr(), tickets.data_ptr())
    outs = []
    for _ in range(2):
        out = torch.full(((V + T_pos), D), 7.0, device=dev)
        L.call("cr_table_grad", C.byref(g), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    want = _table_grad_reference(V, T_pos, D, seq, pos, neg, rows, rows2, scale, emb, coef)
    listed = np.zeros(V + T_pos, bool)
    listed[np.unique(np.r_[seq, pos, neg])] = True
    listed[0] = False
    listed[V:] = True
    assert (outs[0][~listed] == 7.0).all()                # rows without a unit are not touched
    err = np.abs(outs[0][listed] - want[listed]).max()
    assert err <= 2e-5 * max(1.0, np.abs(want).max()), err
    # the same index out of a ring slot chosen by the step number
    slots, off = 3, 16
    ring = torch.zeros(slots, off + lay.total_words, dtype=torch.int32, device=dev)
    ring[2, off:] = d_ix
    step.fill_(5)                                         # 5 mod 3 = 2
    g2 = L.TgradDesc(None, ring.data_ptr(), slots, off + lay.total_words, off, step.data_ptr(), lay, d_rows.data_ptr(),
                     d_rows2.data_ptr() if two else None, D, scale, d_emb.data_ptr(), D, d_coef.data_ptr(), D, part.data_ptr(), tickets.data_ptr())
    out = torch.full(((V + T_pos), D), 7.0, device=dev)
    L.call("cr_table_grad", C.byref(g2), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), outs[0])
    assert int(tickets.abs().sum()) == 0                  # every launch leaves the slices' tickets zero


@pytest.mark.gpu
@pytest.mark.parametrize("D,H,stream", [(50, 1, "0"), (64, 2, "0"), (128, 4, "0"), (128, 4, "1"), (33, 1, "0")])
def test_dense_adam_moves_rows_without_a_unit_on_their_momentum(D, H, stream, monkeypatch):
    """TensorFlow's Adam is dense on a looked-up table (modules.py:154-157 + sasrec.py:120-121): a row step 1 touched and step 2 does not
    still moves in step 2 -- m, v decay, p goes on by lr_t m / (sqrt(v) + eps).  Under the occurrence index that update comes from
    cr_adam_step's sweep of the rows WITHOUT a unit (bitmap), the listed rows' from the gather: every row against the closed form."""
    import torch
    from castrec_amd import engine as E
    monkeypatch.setenv("CASTREC_ADAM_STREAM", stream)     # (the streaming sweep is chosen by size: forced here on a small table)
    rs = np.random.RandomState(D)
    B, T, V = 4, 20, 300
    hp = E.Hyper(maxlen=T, hidden_units=D, num_blocks=1, num_heads=H, dropout_rate=0.0, lr=1e-3, seed=3)
    eng = E.Engine("sasrec", 9, V, hp, B, training=True)
    assert eng.use_index
    lay = eng.layout
    item = lay.view(eng.P, "item_emb")
    p0 = item.clone()
    halves = [np.arange(1, 120), np.arange(100, 220)]      # step 2 drops rows 1..99, keeps 100..119, adds 120..219; 220.. never touched
    snaps = []
    for step, ids in enumerate(halves, 1):
        seq, pos, neg = (ids[rs.randint(0, len(ids), (B, T))] for _ in range(3))
        z = np.zeros_like(seq)
        eng.set_batch(seq, pos, neg, z, z, z)
        eng.set_step(step)
        eng.launch_step(apply=True)
        torch.cuda.synchronize()
        snaps.append((item.clone(), lay.view(eng.Mom, "item_emb").clone(), lay.view(eng.Vel, "item_emb").clone(), np.unique(np.r_[seq, pos, neg])))
    (p1, m1, v1, t1), (p2, m2, v2, t2) = snaps
    only1 = torch.from_numpy(np.setdiff1d(t1, t2)).to(eng.dev)
    never = torch.from_numpy(np.setdiff1d(np.arange(1, V + 1), np.union1d(t1, t2))).to(eng.dev)
    assert len(only1) > 20 and len(never) > 20
    assert torch.equal(p2[never], p0[never]) and not bool(m2[never].any())          # zero moments: no step moves them
    lr_t = 1e-3 * np.sqrt(1 - 0.98 ** 2) / (1 - 0.9 ** 2)
    m_want, v_want = 0.9 * m1[only1].double(), 0.98 * v1[only1].double()
    p_want = p1[only1].double() - lr_t * m_want / (v_want.sqrt() + 1e-8)
    assert float(m1[only1].abs().max()) > 0 and float((p_want - p1[only1].double()).abs().max()) > 3e-4
    assert float((m2[only1].double() - m_want).abs().max()) <= 1e-6 * float(m_want.abs().max())
    assert float((v2[only1].double() - v_want).abs().max()) <= 1e-6 * float(v_want.abs().max())
    assert float((p2[only1].double() - p_want).abs().max()) <= 2e-7
