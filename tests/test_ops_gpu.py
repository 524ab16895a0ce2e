"""Per-op parity of the HIP kernels (through the C ABI) against float64 torch-CPU restatements.
Tolerances: fp32 kernels vs fp64 reference, relative to the tensor scale (stated per test)."""
import math

import numpy as np
import pytest
import torch

import dropout_ref as dr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import castrec_amd  # noqa: F401
    from castrec_amd import ops as O
    assert torch.cuda.is_available()
    return O


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dtype).cuda().contiguous()


def relerr(got, want):
    want = np.asarray(want, np.float64)
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))


def new_state(step=0):
    st = torch.zeros(16, dtype=torch.float32, device="cuda")          # CR_STATE_FLOATS
    st[4:5].view(torch.int32)[0] = step
    return st


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 50, 50), (333, 200, 150), (130, 50, 200), (17, 7, 3)])
@pytest.mark.parametrize("trans_b", [False, True])
def test_gemm_rows_plain(ops, M, N, K, trans_b):
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((M, K)); B = rs.standard_normal((K, N))
    Bd = dev(B.T.copy() if trans_b else B)
    Ad, Cd = dev(A), torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_rows([ops.gemm_desc(Ad, K, Bd, K if trans_b else N, Cd, N, M, N, K, trans_b=trans_b)])
    torch.cuda.synchronize()
    assert relerr(Cd, A @ B) < 2e-6


def test_gemm_rows_epilogue_batched_and_dropout(ops):
    rs = np.random.RandomState(3)
    M, N, K = 150, 50, 50
    st = new_state(step=5)
    drop = ops.Drop(0.3, seed=77, state=st, row_offset=64)
    A = rs.standard_normal((M, K)); B = rs.standard_normal((K, N)); bias = rs.standard_normal(N)
    R = rs.standard_normal((M, N)); ids = rs.randint(0, 3, M).astype(np.int32); C0 = rs.standard_normal((M, N))
    Ad, Bd, bd, Rd, idd = dev(A), dev(B), dev(bias), dev(R), dev(ids, torch.int32)
    C1 = dev(C0); C2 = torch.empty(M, N + 6, device="cuda"); C3 = torch.empty(M, N, device="cuda")
    descs = [
        ops.gemm_desc(Ad, K, Bd, N, C1, N, M, N, K, bias=bd, relu=True, rng=drop.rng(9), residual=Rd, ldr=N,
                      mask_ids=idd, accumulate=True),
        ops.gemm_desc(Ad, K, Bd, N, C2, N + 6, M, N, K, bias=bd),                  # strided output
        ops.gemm_desc(Ad, K, Bd, N, C3, N, M, N, K, relu=True),
    ]
    ops.gemm_rows(descs)
    torch.cuda.synchronize()
    keep = dr.rows_mask(77, 5, 9, 0.3, M, N, row_offset=64)
    _, scale = dr.thresh_scale(0.3)
    v = np.maximum(A @ B + bias, 0) * keep * float(scale) + R
    v = v * (ids != 0)[:, None]
    assert relerr(C1, C0 + v) < 3e-6
    assert relerr(C2[:, :N], A @ B + bias) < 3e-6
    assert relerr(C3, np.maximum(A @ B, 0)) < 3e-6
    assert 0.6 < keep.mean() < 0.8


@pytest.mark.parametrize("M,N,K,ns", [(1000, 50, 50, 7), (257, 200, 150, 64), (64, 50, 50, 1), (50, 64, 64, 128)])
def test_gemm_wgrad_slabs(ops, M, N, K, ns):
    rs = np.random.RandomState(M + ns)
    A = rs.standard_normal((M, K)); G = rs.standard_normal((M, N))
    stride = K * N + N + 5
    slabs = torch.full((ns, stride), float("nan"), device="cuda")
    Ad, Gd = dev(A), dev(G)          # descriptors hold raw pointers: keep the tensors alive
    ops.gemm_wgrad([ops.wgrad_desc(Ad, K, Gd, N, slabs, slabs[0, K * N:], M, N, K)], stride, ns)
    torch.cuda.synchronize()
    tot = slabs.double().sum(0).cpu().numpy()
    assert relerr(tot[:K * N].reshape(K, N), A.T @ G) < 3e-6
    assert relerr(tot[K * N:K * N + N], G.sum(0)) < 3e-6


def test_gemm_wgrad_batched(ops):
    rs = np.random.RandomState(9)
    M = 300
    probs = [(50, 50), (50, 50), (30, 70)]
    ns, stride = 16, 20000
    slabs = torch.zeros(ns, stride, device="cuda")
    descs, refs, off, alive = [], [], 0, []
    for (K, N) in probs:
        A = rs.standard_normal((M, K)); G = rs.standard_normal((M, N))
        Ad, Gd = dev(A), dev(G)
        alive += [Ad, Gd]
        descs.append(ops.wgrad_desc(Ad, K, Gd, N, slabs[0, off:], slabs[0, off + K * N:], M, N, K))
        refs.append((off, K, N, A.T @ G, G.sum(0)))
        off += K * N + N
    ops.gemm_wgrad(descs, stride, ns)
    torch.cuda.synchronize()
    tot = slabs.double().sum(0).cpu().numpy()
    for off, K, N, w, b in refs:
        assert relerr(tot[off:off + K * N].reshape(K, N), w) < 3e-6
        assert relerr(tot[off + K * N:off + K * N + N], b) < 3e-6


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D", [(10, 50), (513, 64), (33, 200), (7, 6)])
def test_layernorm_fwd_bwd(ops, M, D):
    rs = np.random.RandomState(D)
    x = rs.standard_normal((M, D)) * 2 + 0.5
    x[1] = 0.0                                   # zero row -> key mask 0, LN(0) = beta
    g = rs.standard_normal(D); b = rs.standard_normal(D); dy = rs.standard_normal((M, D)); dx0 = rs.standard_normal((M, D))
    xt = torch.tensor(x, requires_grad=True); gt = torch.tensor(g, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
    mean = xt.mean(-1, keepdim=True); var = ((xt - mean) ** 2).mean(-1, keepdim=True)
    yt = gt * ((xt - mean) / (var + 1e-8) ** 0.5) + bt
    yt.backward(torch.tensor(dy))
    xd, gd, bd = dev(x), dev(g), dev(b)
    y = torch.empty(M, D, device="cuda"); xnz = torch.empty(M, device="cuda"); ynz = torch.empty(M, device="cuda")
    ops.layernorm_fwd(xd, D, gd, bd, y, D, M, D, x_nonzero=xnz, y_nonzero=ynz)
    ns, stride = 5, 2 * D + 3
    slabs = torch.full((ns, stride), float("nan"), device="cuda")
    dx = dev(dx0)
    dyd = dev(dy)
    ops.layernorm_bwd(xd, D, gd, dyd, D, dx, D, slabs, slabs[0, D:], stride, ns, M, D, accumulate=True)
    dx2 = torch.empty(M, D, device="cuda")
    ops.layernorm_bwd(xd, D, gd, dyd, D, dx2, D, slabs, slabs[0, D:], stride, ns, M, D, accumulate=False)
    torch.cuda.synchronize()
    assert relerr(y, yt.detach().numpy()) < 3e-6
    assert xnz.cpu().numpy().tolist() == (np.abs(x.sum(-1)) > 0).astype(np.float32).tolist()
    assert float(xnz[1]) == 0.0 and float(ynz[0]) == 1.0
    assert relerr(dx2, xt.grad.numpy()) < 2e-5          # zero row: rstd = 1e4 amplifies rounding
    assert relerr(dx, dx0 + xt.grad.numpy()) < 2e-5
    tot = slabs.double().sum(0).cpu().numpy()
    assert relerr(tot[:D], gt.grad.numpy()) < 5e-6
    assert relerr(tot[D:2 * D], bt.grad.numpy()) < 5e-6


# ------------------------------------------------------------------------------------------------
def test_embed_fwd_bwd_large_table(ops):
    rs = np.random.RandomState(2)
    B, T, D, V = 6, 10, 50, 40
    M = B * T
    ids = rs.randint(0, V, (B, T)).astype(np.int32); ids[0, :4] = 0
    table = rs.standard_normal((V, D)); pos = rs.standard_normal((T, D)); add = rs.standard_normal((M, D + 3))
    dout = rs.standard_normal((M, 2 * D))
    st = new_state(step=2)
    drop = ops.Drop(0.25, seed=5, state=st)
    out = torch.zeros(M, 2 * D, device="cuda")
    idd = dev(ids.reshape(-1), torch.int32)
    tabd, posd, addd, doutd = dev(table), dev(pos), dev(add), dev(dout)
    fd = ops.embed_fwd(idd, tabd, T, out, 2 * D, col_off=D, zero_pad=True, scale=D ** 0.5, pos_table=posd,
                       addend=addd, ld_add=D + 3, rng=drop.rng(4), mask_ids=idd)
    torch.cuda.synchronize()
    keep = dr.rows_mask(5, 2, 4, 0.25, M, D); _, scale = dr.thresh_scale(0.25)
    tz = table.copy(); tz[0] = 0
    ref = tz[ids.reshape(-1)] * D ** 0.5 + np.tile(pos, (B, 1)) + add[:, :D]
    ref = ref * keep * float(scale) * (ids.reshape(-1) != 0)[:, None]
    assert relerr(out[:, D:], ref) < 2e-6
    assert float(out[:, :D].abs().max()) == 0.0
    # backward
    tg = torch.zeros(V, D, device="cuda"); pg = torch.full((T, D), float("nan"), device="cuda")
    da = torch.full((M, D + 3), float("nan"), device="cuda")
    ops.embed_bwd(fd, doutd, table_grad=tg, pos_grad=pg, d_addend=da)
    torch.cuda.synchronize()
    g = dout[:, D:] * keep * float(scale) * (ids.reshape(-1) != 0)[:, None]
    tg_ref = np.zeros((V, D)); np.add.at(tg_ref, ids.reshape(-1), g * D ** 0.5); tg_ref[0] = 0
    assert relerr(tg, tg_ref) < 3e-6
    assert relerr(pg, g.reshape(B, T, D).sum(0)) < 3e-6
    assert relerr(da[:, :D], g) < 1e-6


def test_embed_bwd_small_table_slabs(ops):
    rs = np.random.RandomState(4)
    B, T, D, V = 16, 20, 50, 9
    M = B * T
    ids = rs.randint(0, V, M).astype(np.int32)
    table = rs.standard_normal((V, D)); dout = rs.standard_normal((M, D))
    out = torch.empty(M, D, device="cuda")
    idd, tabd, doutd = dev(ids, torch.int32), dev(table), dev(dout)
    fd = ops.embed_fwd(idd, tabd, T, out, D, zero_pad=True, scale=2.0)
    ns, stride = 12, V * D + 11
    slabs = torch.full((ns, stride), float("nan"), device="cuda")
    ops.embed_bwd(fd, doutd, table_grad=slabs, slab_stride=stride, n_slabs=ns)
    torch.cuda.synchronize()
    ref = np.zeros((V, D)); np.add.at(ref, ids, dout * 2.0); ref[0] = 0
    assert relerr(slabs[:, :V * D].double().sum(0).reshape(V, D), ref) < 3e-6


def test_eltwise_ops(ops):
    rs = np.random.RandomState(6)
    M, N = 37, 50
    x = rs.standard_normal((M, N)); aux = rs.standard_normal((M, N)); ids = rs.randint(0, 2, M).astype(np.int32)
    st = new_state(step=1)
    drop = ops.Drop(0.5, seed=9, state=st)
    from castrec_amd import lib as L
    y = torch.zeros(M, 3 * N, device="cuda")
    xd, auxd, idd = dev(x), dev(aux), dev(ids, torch.int32)
    ops.eltwise(L.ELT_COPY, xd, N, y[:, N:], 3 * N, M, N)
    ops.eltwise(L.ELT_ADD, xd, N, y[:, 2 * N:], 3 * N, M, N, aux=auxd, ldaux=N)
    ops.eltwise(L.ELT_DROPOUT, xd, N, y, 3 * N, M, N, rng=drop.rng(3), mask_ids=idd)
    z = dev(aux)
    ops.eltwise(L.ELT_RELU_BWD, xd, N, z, N, M, N, aux=auxd, ldaux=N, accumulate=True)
    torch.cuda.synchronize()
    keep = dr.rows_mask(9, 1, 3, 0.5, M, N)
    assert torch.equal(y[:, N:2 * N], xd)
    assert relerr(y[:, 2 * N:], x + aux) < 1e-7
    assert relerr(y[:, :N], x * keep * 2.0 * (ids != 0)[:, None]) < 1e-7
    assert relerr(z, aux + x * (aux > 0)) < 1e-7


# ------------------------------------------------------------------------------------------------
def attn_core_ref(Q, K, V, kvalid, qvalid, resid, H, keep=None, rate=0.0):
    """modules.py:208-269 from the projected Q/K/V on (float64 torch)."""
    N, T, C = Q.shape
    d = C // H
    Q_ = torch.cat(torch.split(Q, d, dim=2), 0); K_ = torch.cat(torch.split(K, d, dim=2), 0); V_ = torch.cat(torch.split(V, d, dim=2), 0)
    out = Q_ @ K_.transpose(1, 2) / d ** 0.5
    km = kvalid.repeat(H, 1)[:, None, :].expand(-1, T, -1)
    pad = torch.full_like(out, float(-2 ** 32 + 1))
    out = torch.where(km == 0, pad, out)
    tril = torch.tril(torch.ones(T, T, dtype=out.dtype))
    out = torch.where(tril[None] == 0, pad, out)
    out = torch.softmax(out, -1)
    out = out * qvalid.repeat(H, 1)[:, :, None]
    if keep is not None:
        out = out * torch.tensor(keep, dtype=out.dtype) / (1.0 - rate)
    w = out
    out = out @ V_
    out = torch.cat(torch.split(out, N, dim=0), 2)
    return out + resid, w


ATTN_CASES = [  # B, T, H, d, rate
    (3, 8, 1, 6, 0.0), (2, 50, 1, 50, 0.0), (2, 200, 1, 50, 0.2), (3, 37, 2, 32, 0.0), (2, 50, 4, 32, 0.5),
    (1, 256, 1, 64, 0.0), (2, 100, 2, 25, 0.3), (5, 16, 1, 50, 0.0), (5, 24, 1, 20, 0.0), (4, 20, 1, 50, 0.0), (4, 20, 1, 20, 0.2),
    # outside the LDS-resident MFMA envelope (T > 256 or head dim > 64) -> general-shape kernels (cr_attn_wide.hip)
    (2, 300, 1, 50, 0.2), (2, 512, 2, 64, 0.0), (2, 40, 1, 100, 0.3), (2, 130, 1, 256, 0.0), (1, 1024, 1, 8, 0.0),
]


@pytest.mark.parametrize("B,T,H,d,rate", ATTN_CASES)
def test_attention_fwd_bwd(ops, B, T, H, d, rate):
    rs = np.random.RandomState(B * 1000 + T + d)
    Cc = H * d
    M = B * T
    ld = Cc + 5
    Q = rs.standard_normal((B, T, Cc)); K = rs.standard_normal((B, T, Cc)); V = rs.standard_normal((B, T, Cc))
    resid = rs.standard_normal((B, T, Cc)); dout = rs.standard_normal((B, T, Cc))
    kvalid = (rs.rand(B, T) > 0.2).astype(np.float64); qvalid = (rs.rand(B, T) > 0.1).astype(np.float64)
    kvalid[0, :max(2, T // 4)] = 0                       # left padding: first queries see no valid key -> uniform rows
    if B > 1:
        kvalid[1, :] = 0                                 # a sample with NO valid key at all (CAST context, all bins 0)
    dout[0, 0] = 0                                       # a uniform row with zero incoming gradient (flag 2 path)
    st = new_state(step=3)
    drop = ops.Drop(rate, seed=11, state=st)
    keep = dr.attn_mask(11, 3, 21, rate, H, B, T) if rate > 0 else None
    Qt, Kt, Vt = (torch.tensor(a, requires_grad=True) for a in (Q, K, V))
    ref, w = attn_core_ref(Qt, Kt, Vt, torch.tensor(kvalid), torch.tensor(qvalid), torch.tensor(resid), H, keep, rate)
    ref.backward(torch.tensor(dout))

    def padded(a):
        buf = torch.full((M, ld), float("nan"), device="cuda")
        buf[:, :Cc] = dev(a.reshape(M, Cc))
        return buf
    Qd, Kd, Vd, Rd, dOd = padded(Q), padded(K), padded(V), padded(resid), padded(dout)
    out = torch.full((M, ld), float("nan"), device="cuda")
    wts = torch.full((H * B, T, T), float("nan"), device="cuda")
    kvd, qvd = dev(kvalid.reshape(-1)), dev(qvalid.reshape(-1))
    desc = ops.attn_desc(Qd, Kd, Vd, ld, kvd, qvd, Rd, ld, out, ld, B, T, H, d,
                         rng=drop.rng(21), attn_weights=wts)
    ops.attn_fwd(desc)
    dQ = torch.full((M, ld), float("nan"), device="cuda"); dK = torch.full((M, ld), float("nan"), device="cuda")
    dV = torch.full((M, ld), float("nan"), device="cuda")
    stats = torch.empty(H * B * T * 4, device="cuda")
    ops.attn_bwd(desc, dOd, ld, dQ, dK, dV, ld, stats)
    torch.cuda.synchronize()
    errs = dict(w=relerr(wts, w.detach().numpy()), out=relerr(out[:, :Cc], ref.detach().numpy().reshape(M, Cc)),
                dV=relerr(dV[:, :Cc], Vt.grad.numpy().reshape(M, Cc)), dQ=relerr(dQ[:, :Cc], Qt.grad.numpy().reshape(M, Cc)),
                dK=relerr(dK[:, :Cc], Kt.grad.numpy().reshape(M, Cc)))
    print("attn errs", errs)
    assert errs["w"] < 5e-6 and errs["out"] < 5e-6, errs
    assert errs["dV"] < 1e-5 and errs["dQ"] < 1e-5 and errs["dK"] < 1e-5, errs
    # uniform rows really are 1/T over ALL keys (future ones included)
    row = w.detach().numpy()[0, 0]
    if qvalid[0, 0] and keep is None:
        assert np.allclose(row, 1.0 / T)
    # single-pass backward (forward statistics saved, delta supplied, dQ returned as two partial sums)
    if H == 1 and T <= 256 and 4 <= d <= 64:
        rst = torch.full((H * B * T * 4,), float("nan"), device="cuda")
        out2 = torch.full((M, ld), float("nan"), device="cuda")
        desc2 = ops.attn_desc(Qd, Kd, Vd, ld, kvd, qvd, Rd, ld, out2, ld, B, T, H, d, rng=drop.rng(21), row_stats=rst)
        ops.attn_fwd(desc2)
        delta = (dOd[:, :Cc] * (out2[:, :Cc] - Rd[:, :Cc])).sum(1).contiguous()
        dQa = torch.full((M, ld), float("nan"), device="cuda"); dQb = torch.full((M, ld), float("nan"), device="cuda")
        dK2 = torch.full((M, ld), float("nan"), device="cuda"); dV2 = torch.full((M, ld), float("nan"), device="cuda")
        ops.attn_bwd(desc2, dOd, ld, dQa, dK2, dV2, ld, stats, delta=delta, dQ_part=dQb)
        torch.cuda.synchronize()
        e2 = dict(out=relerr(out2[:, :Cc], ref.detach().numpy().reshape(M, Cc)),
                  dV=relerr(dV2[:, :Cc], Vt.grad.numpy().reshape(M, Cc)),
                  dQ=relerr((dQa + dQb)[:, :Cc], Qt.grad.numpy().reshape(M, Cc)),
                  dK=relerr(dK2[:, :Cc], Kt.grad.numpy().reshape(M, Cc)))
        print("single-pass errs", e2)
        assert e2["out"] < 5e-6 and e2["dV"] < 1e-5 and e2["dQ"] < 2e-5 and e2["dK"] < 2e-5, e2


def test_attention_dead_rows_and_rejects(ops):
    from castrec_amd import lib as L
    rs = np.random.RandomState(8)
    B, T, H, d = 2, 24, 1, 50
    M = B * T
    Q, K, V, R = (dev(rs.standard_normal((M, d))) for _ in range(4))
    ids = np.ones(M, np.int32); ids[:10] = 0
    kv = dev((ids != 0).astype(np.float32)); qv = dev(np.ones(M))
    out = torch.empty(M, d, device="cuda")
    idd = dev(ids, torch.int32)
    desc = ops.attn_desc(Q, K, V, d, kv, qv, R, d, out, d, B, T, H, d, dead_ids=idd)
    ops.attn_fwd(desc)
    torch.cuda.synchronize()
    assert torch.equal(out[:10], R[:10])                 # dead rows: A = 0 -> residual only
    bad = ops.attn_desc(Q, K, V, d, kv, qv, R, d, out, d, 1, 1100, 1, 50)
    with pytest.raises(RuntimeError, match="T=1100"):
        ops.attn_fwd(bad)
    bad = ops.attn_desc(Q, K, V, 300, kv, qv, R, d, out, d, 1, 8, 1, 300)
    with pytest.raises(RuntimeError, match="head dim 300"):
        ops.attn_fwd(bad)


# ------------------------------------------------------------------------------------------------
def test_head_fwd_bwd_and_test_logits(ops):
    rs = np.random.RandomState(10)
    B, T, D, V = 4, 12, 50, 30
    M = B * T
    s = rs.standard_normal((M, D)) * 0.5; table = rs.standard_normal((V, D)) * 0.5
    pos = rs.randint(0, V, M).astype(np.int32); neg = rs.randint(1, V, M).astype(np.int32)
    pos[:5] = 0; neg[:5] = 0
    st_ = torch.tensor(s, requires_grad=True); tb = torch.tensor(table, requires_grad=True)
    tz = torch.cat([torch.zeros(1, D, dtype=torch.float64), tb[1:]], 0)
    pl = (tz[pos.astype(np.int64)] * st_).sum(-1); nl = (tz[neg.astype(np.int64)] * st_).sum(-1)
    ist = torch.tensor((pos != 0).astype(np.float64))
    loss_sum = (-torch.log(torch.sigmoid(pl) + 1e-24) * ist - torch.log(1 - torch.sigmoid(nl) + 1e-24) * ist).sum()
    loss_sum.backward()
    auc_sum = float((((torch.sign(pl - nl) + 1) / 2) * ist).sum())
    state = new_state(step=7)
    ds = torch.full((M, D), float("nan"), device="cuda"); tg = torch.zeros(V, D, device="cuda")
    plg = torch.empty(M, device="cuda"); nlg = torch.empty(M, device="cuda")
    sd, tabd, posd, negd = dev(s), dev(table), dev(pos, torch.int32), dev(neg, torch.int32)
    ops.head_fwd_bwd(sd, D, tabd, posd, negd, M, D, state, d_seq_emb=ds, ldd=D,
                     table_grad=tg, pos_logits=plg, neg_logits=nlg)
    cand = rs.randint(0, V, (B, 101)).astype(np.int32)
    candd = dev(cand, torch.int32)
    lg = torch.empty(B, 101, device="cuda")
    ops.test_logits(sd, D, tabd, candd, B, T, D, lg)
    torch.cuda.synchronize()
    got = state.cpu().numpy()
    assert got[0] == pytest.approx(float(loss_sum), rel=2e-6)
    assert got[1] == pytest.approx(auc_sum) and got[2] == float(ist.sum())
    # snapshot taken by the last workgroup to finish: the totals, the step counter, ticket re-armed (castrec.h state block)
    assert tuple(got[8:11]) == tuple(got[0:3]) and got[8:12].view(np.uint32)[3] == 7 and got[12:13].view(np.uint32)[0] == 0
    assert relerr(plg, pl.detach().numpy()) < 2e-6 and relerr(nlg, nl.detach().numpy()) < 2e-6
    assert relerr(ds, st_.grad.numpy()) < 3e-6
    assert relerr(tg, tb.grad.numpy()) < 3e-6
    last = s.reshape(B, T, D)[:, -1]
    tzn = tz.detach().numpy()
    assert relerr(lg, np.einsum("bd,bjd->bj", last, tzn[cand.astype(np.int64)])) < 2e-6


@pytest.mark.parametrize("external_stats", [False, True, "self_advancing"])
def test_adam_tf_three_steps_with_slabs(ops, external_stats):
    """external_stats: {loss_sum, auc_sum, n_target} come from a separate buffer (the all-reduced bucket tail of the
    data-parallel path) while state[0..2] hold this rank's local values, which must then be ignored."""
    from oracle import fpmodel as fm
    rs = np.random.RandomState(12)
    nt, nd, ns = 300, 130, 6
    p0 = rs.standard_normal(nt + nd)
    P = {"w": torch.tensor(p0)}
    opt = fm.AdamTF(P, lr=1e-3)
    p = dev(p0); m = torch.zeros(nt + nd, device="cuda"); v = torch.zeros(nt + nd, device="cuda")
    tg = torch.zeros(nt, device="cuda"); slabs = torch.zeros(ns, nd, device="cuda")
    state = new_state()
    for step in range(3):
        g = rs.standard_normal(nt + nd); n_target = float(rs.randint(5, 50))
        self_adv = external_stats == "self_advancing"
        if self_adv:
            # no cr_step_begin: state[4] already holds this step's number, sums and step come from the snapshot
            # (state[8..11], normally written by the head kernel), and Adam ends the step itself
            if step == 0:
                state[4:5].view(torch.int32)[0] = 1
            assert float(state[:4].abs().max()) == 0.0           # left clean by the previous step's Adam
        else:
            ops.step_begin(state)
        state[0] = 3.5 * n_target; state[1] = 0.25 * n_target; state[2] = n_target
        stats, snap = None, None
        if self_adv:
            state[8:11] = state[:3]; state[11:12].view(torch.int32)[0] = step + 1
            stats, snap = state[8:11], state[11:12]
        elif external_stats:
            stats = state[:3].clone()
            state[0] = -1.0; state[1] = -2.0; state[2] = 1.0
        tg.copy_(dev(g[:nt] * n_target))                            # un-normalised gradients
        parts = rs.dirichlet(np.ones(ns), nd).T * (g[nt:] * n_target)[None]
        slabs.copy_(dev(parts))
        ops.adam_step(p, m, v, tg, slabs, nt, nd, ns, 1e-3, state, stats=stats, step_snapshot=snap)
        torch.cuda.synchronize()
        P = opt.step(P, {"w": torch.tensor(g)})
        assert relerr(p, P["w"].numpy()) < 1e-6
        assert float(tg.abs().max()) == 0.0                         # table grad zeroed for the next step
        s = state.cpu().numpy()
        assert s[5] == pytest.approx(3.5, rel=1e-6) and s[6] == pytest.approx(0.25, rel=1e-6)
        assert int(state[4:5].view(torch.int32)[0]) == step + (2 if self_adv else 1)     # self-advancing: the NEXT step's number


def test_graph_capture_replay(ops):
    from castrec_amd import lib as L
    M, N = 64, 50
    x = dev(np.arange(M * N, dtype=np.float32).reshape(M, N)); y = torch.zeros(M, N, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = ops.Graph()
        g.begin()
        ops.eltwise(L.ELT_ADD, x, N, y, N, M, N, aux=x, ldaux=N, accumulate=True)     # y += 2x
        g.end()
        for _ in range(3):
            g.launch()
    s.synchronize()
    assert torch.equal(y, 6 * x)


@pytest.mark.parametrize("D", [64, 128, 256, 52])
def test_test_logits_vector_path(ops, D):
    """cr_test_logits for hidden sizes that are multiples of 4 takes the 16-lanes-per-row form (the read-only gather
    the bench holds against the HBM read roofline): same values as the scalar path, row 0 reads as zeros."""
    rs = np.random.RandomState(D)
    B, T, V, nc = 5, 7, 97, 101
    table = rs.standard_normal((V, D)); semb = rs.standard_normal((B * T, D))
    cand = rs.randint(0, V, (B, nc)); cand[0, :3] = 0
    t0 = table.copy(); t0[0] = 0
    want = np.einsum("bd,bnd->bn", semb.reshape(B, T, D)[:, -1], t0[cand])
    out = torch.full((B, nc), float("nan"), device="cuda")
    ops.test_logits(dev(semb), D, dev(table), dev(cand, torch.int32), B, T, D, out)
    torch.cuda.synchronize()
    assert relerr(out, want) < 2e-6


def test_adam_lazy_rows_against_dense(ops):
    """Row-sparse Adam on the leading item table (castrec.h cr_adam_desc.lazy_ids; a documented deviation for tables like
    config C5's).  Step 1 from zero moments: identical to dense everywhere.  Step 2 with other rows: rows touched in both
    steps or only now are identical; rows touched only in step 1 stay put under lazy Adam while dense Adam moves them by
    their decayed momentum alone -- exactly lr_t * (b1 m) / (sqrt(b2 v) + eps); the dense section is never affected."""
    rs = np.random.RandomState(4)
    V, D, n_dense, lr, b1, b2, eps = 40, 12, 70, 1e-2, 0.9, 0.98, 1e-8
    n_table = V * D + 5 * D                                  # item table + a small positional table swept densely
    p0 = rs.standard_normal(n_table + n_dense).astype(np.float32)
    ids1 = np.array([3, 7, 7, 0, 9, 12, 3], np.int32); ids2 = np.array([7, 9, 20, 21, 0], np.int32)

    def grads(ids, seed):
        g = np.zeros(n_table, np.float32)
        r = np.random.RandomState(seed)
        for i in set(ids.tolist()) - {0}:
            g[i * D:(i + 1) * D] = r.standard_normal(D)
        g[V * D:] = r.standard_normal(5 * D)
        return g, r.standard_normal((3, n_dense)).astype(np.float32)

    def run(lazy):
        P, M, Vv = dev(p0.copy()), torch.zeros(n_table + n_dense, device="cuda"), torch.zeros(n_table + n_dense, device="cuda")
        st = new_state(step=1)
        flags = torch.zeros(V, dtype=torch.int32, device="cuda")
        snaps = []
        for step, (ids, seed) in enumerate(((ids1, 1), (ids2, 2)), 1):
            g, slabs = grads(ids, seed)
            st[0], st[1], st[2] = 5.0, 2.0, 4.0
            st[4:5].view(torch.int32)[0] = step
            kw = dict(lazy_ids=dev(ids, torch.int32), lazy_rows=V, lazy_D=D, lazy_flags=flags) if lazy else {}
            ops.adam_step(P, M, Vv, dev(g), dev(slabs), n_table, n_dense, 3, lr, st, **kw)
            torch.cuda.synchronize()
            snaps.append((P.cpu().numpy().copy(), M.cpu().numpy().copy(), Vv.cpu().numpy().copy()))
        return snaps
    dense, lazy = run(False), run(True)
    np.testing.assert_array_equal(dense[0][0], lazy[0][0])              # step 1: nothing to tell them apart
    pd, pl = dense[1][0], lazy[1][0]
    np.testing.assert_array_equal(pd[V * D:], pl[V * D:])               # positional table + dense section: always dense
    row = lambda a, i: a[i * D:(i + 1) * D]
    for i in (7, 9, 20, 21):                                            # touched now (7, 9 also before): identical
        np.testing.assert_array_equal(row(pd, i), row(pl, i))
    for i in (3, 12):                                                   # touched only in step 1: momentum drift under dense Adam
        np.testing.assert_array_equal(row(pl, i), row(lazy[0][0], i))
        m1, v1 = row(dense[0][1], i).astype(np.float64), row(dense[0][2], i).astype(np.float64)
        lr_t = lr * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
        drift = lr_t * (b1 * m1) / (np.sqrt(b2 * v1) + eps)
        np.testing.assert_allclose(row(dense[0][0], i) - row(pd, i), drift, rtol=2e-5, atol=1e-9)
    untouched = [i for i in range(1, V) if i not in (3, 7, 9, 12, 20, 21)]
    for i in untouched + [0]:
        np.testing.assert_array_equal(row(pl, i), row(p0, i)); np.testing.assert_array_equal(row(pd, i), row(p0, i))


GEMM_BF_TOL = {1: 1e-4, 2: 3e-2}


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 50, 50), (333, 200, 150), (130, 128, 128), (257, 512, 128), (70, 9, 8), (1000, 128, 512)])
@pytest.mark.parametrize("trans_b", [False, True])
def test_gemm_rows_bf16_mfma(ops, M, N, K, trans_b, prec):
    """cr_gemm_rows with precision CR_PREC_BF16X3 / CR_PREC_BF16 (csrc/cr_gemm_bf.hip): every tail (rows, columns, K not a
    multiple of 8 / 32 / 64), padded leading dimensions with NaN beyond the matrices, epilogue included."""
    rs = np.random.RandomState(M + N + K)
    lda, ldb, ldc = K + 3, (K if trans_b else N) + 5, N + 2
    A = rs.standard_normal((M, K)); B = rs.standard_normal((N, K) if trans_b else (K, N)); bias = rs.standard_normal(N)
    res = rs.standard_normal((M, N)); ids = rs.randint(0, 3, M)

    def pad(a, ld):
        buf = torch.full((a.shape[0], ld), float("nan"), device="cuda")
        buf[:, :a.shape[1]] = dev(a)
        return buf
    Ad, Bd, Cd, Rd = pad(A, lda), pad(B, ldb), torch.full((M, ldc), float("nan"), device="cuda"), pad(res, N)
    want = np.maximum(A @ (B.T if trans_b else B) + bias, 0) + res
    want = want * (ids != 0)[:, None]
    d = ops.gemm_desc(Ad, lda, Bd, ldb, Cd, ldc, M, N, K, bias=dev(bias), trans_b=trans_b, relu=True, residual=Rd, ldr=N,
                      mask_ids=dev(ids, torch.int32), precision=prec)
    ops.gemm_rows([d])
    torch.cuda.synchronize()
    assert relerr(Cd[:, :N], want) < GEMM_BF_TOL[prec]
    assert torch.isnan(Cd[:, N:]).all()


@pytest.mark.parametrize("prec", [1, 2])
@pytest.mark.parametrize("M,N,K,ns", [(1000, 50, 50, 7), (257, 200, 150, 64), (64, 128, 128, 1), (25600 // 8, 128, 256, 25), (50, 64, 64, 128)])
def test_gemm_wgrad_bf16_mfma(ops, M, N, K, ns, prec):
    rs = np.random.RandomState(M + N + K + ns)
    A = rs.standard_normal((M, K)); G = rs.standard_normal((M, N))
    stride = K * N + N + 7
    slabs = torch.full((ns, stride), float("nan"), device="cuda")
    d = ops.wgrad_desc(dev(A), K, dev(G), N, slabs, slabs[0, K * N:], M, N, K, precision=prec)
    ops.gemm_wgrad([d], stride, ns)
    torch.cuda.synchronize()
    dW = slabs[:, :K * N].double().sum(0).reshape(K, N).cpu().numpy()
    db = slabs[:, K * N:K * N + N].double().sum(0).cpu().numpy()
    assert relerr(dW, A.T @ G) < GEMM_BF_TOL[prec]
    assert relerr(db, G.sum(0)) < GEMM_BF_TOL[prec]
    assert torch.isnan(slabs[:, K * N + N:]).all()
