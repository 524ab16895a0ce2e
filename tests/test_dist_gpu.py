"""Two data-parallel ranks of the REAL engine (HIP kernels, captured graphs) against one process on the whole batch.

Both ranks share the box's single card, so the exchange runs over gloo with the flat bucket bounced through host
memory (RCCL refuses two ranks on one device); everything else is the production path of castrec_amd.dist:
row sharding, dropout keyed by the global row index, un-normalised shard gradients + target counts in one bucket,
Adam dividing by the reduced count.  The N-GPU result must equal the 1-GPU result on batch B_global."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, D, ITEMS, USERS, STEPS = 16, 24, 20, 60, 9, int(os.environ.get("CASTREC_TEST_STEPS", "3"))


def make_batch(step, items=ITEMS):
    rs = np.random.RandomState(100 + step)
    seq = rs.randint(1, items + 1, (B, T)); pos = rs.randint(1, items + 1, (B, T)); neg = rs.randint(1, items + 1, (B, T))
    for b in range(B):
        n = rs.randint(0, T - 2)                               # ragged: the two shards hold different target counts
        seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0
    time = rs.randint(0, 11, (B, T)) * (seq != 0)
    z = np.zeros_like(seq)
    return seq, pos, neg, time, z, z


def hyper(E, D=D, H=1):
    return E.Hyper(maxlen=T, hidden_units=D, num_blocks=2, num_heads=H, dropout_rate=0.2, max_bins=10, seed=4)


def perturb_start(eng):
    """Moves the parameters off the TensorFlow initial point (seeded, the same on every engine of a test).  At gamma = 1,
    beta = 0, zero biases, every row whose context embedding is the zero_pad row (time bin 0: cast_1.py:30-38) stays EXACTLY
    zero through all blocks, each LayerNorm sees variance 0 and its backward multiplies by 1 / sqrt(1e-8) = 1e4: the fp64
    oracle has d loss / d ctx_time.0.ln1.beta = 3e13 at that start (D = 20 and D = 128 alike), and Adam on such gradients
    sits on sign boundaries that re-associated sums flip.  With beta off zero only the first LayerNorm of a zero row is
    degenerate, and its input gradient goes to the discarded zero_pad row."""
    g = torch.Generator().manual_seed(5)
    eng.P.add_(0.05 * torch.randn(eng.P.numel(), generator=g).to(eng.P.device))


def _worker(rank, world, port, q, items=ITEMS, sparse=False, D=D, H=1):
    import torch.distributed as dist
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    from castrec_amd.dist import DataParallel, EngineReplica, HostBounce, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_rows(B, rank, world)
    hp = hyper(E, D, H)
    hp.seed = 4                                                # same dropout key on every rank ...
    eng = E.Engine("cast_1", USERS, items, hp, hi - lo, training=True, n_slabs=8, batch_global=B, row_offset=lo * T)
    perturb_start(eng)
    if rank != 0:
        eng.P.mul_(1.5)                                        # ... but a different start: the wrapper must broadcast rank 0's
    rep = HostBounce(EngineReplica(eng, use_graph=True))
    dp = DataParallel(rep, rank, world, sparse=sparse)
    assert dp.sparse == bool(sparse)
    rep.adopt_params()
    m1 = None
    for s in range(STEPS):
        dp.step(make_batch(s, items))
        if s == 0:                                             # Adam's first moment after ONE step = 0.1 x the exchanged gradient
            m1 = {k: eng.layout.view(eng.Mom, k).cpu().numpy().copy() for k in eng.layout.logical_names()}
    torch.cuda.synchronize()
    q.put((rank, {k: v.cpu().numpy() for k, v in eng.get_params().items()}, eng.Gflat[eng.layout.n_total:eng.layout.n_total + 3].cpu().numpy(), m1))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("items,sparse,D,H", [(ITEMS, False, D, 1), (ITEMS, True, D, 1), (5000, True, D, 1),
                                              (ITEMS, False, 128, 4)])       # the wide row kernels (cr_wide.hip) under the same wrapper
def test_two_ranks_on_the_card_equal_one_process_on_the_whole_batch(items, sparse, D, H):
    """dense: the whole bucket in one all-reduce.  sparse: the item table's touched rows as an all-gather of
    de-duplicated (row id, gradient row) pairs -- 60 items: every row is hot and shared by both ranks; 5000 items:
    V >> the 3 * 8 * 24 row slots a rank touches (the regime of config C5)."""
    import torch.multiprocessing as mp
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, items, sparse, D, H)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = E.Engine("cast_1", USERS, items, hyper(E, D, H), B, training=True, n_slabs=8)
    perturb_start(one)
    for s in range(STEPS):
        one.train_step(*make_batch(s, items))
        if s == 0:
            m1 = {k: one.layout.view(one.Mom, k).cpu().numpy().copy() for k in one.layout.logical_names()}
    torch.cuda.synchronize()
    # (1) the exchanged GRADIENT of the first step, read off Adam's first moment (m = 0.1 g after one step: linear, nothing
    # amplifies a rounding difference): the two-rank run's equals the whole-batch run's to the re-association of fp32 sums
    for k in m1:
        if k.endswith(".bk"):
            continue                                           # the key bias: a zero-gradient direction (softmax is shift invariant), rounding noise only
        dm = np.abs(res[0][3][k] - m1[k]).max()
        assert dm <= 5e-6 * max(np.abs(m1[k]).max(), 1e-12), (k, float(dm), float(np.abs(m1[k]).max()))
    ref = {k: v.cpu().numpy() for k, v in one.get_params().items()}
    # both ranks hold the same replica after the exchange
    for k in ref:
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)
    # (2) and the PARAMETERS after three steps are the single-process result: same inputs, same dropout masks, global
    # normalisation.  Sums are re-associated (per-shard slabs, then the all-reduce; the table gradients are float atomics whose
    # order changes from run to run), and from the second step on a rounding-level parameter difference can flip a ReLU gate or
    # move a gradient element whose contributions nearly cancel, which Adam's m / (sqrt(v) + eps) turns into a step of a
    # fraction of lr.  Seen over 19 runs of the D = 128 case (tools/probes/flake.sh): 14 runs with <= 3 elements above 5e-6; one
    # with 46 elements of trunk.0.w1 up to 3.8e-5 (one hidden unit's column), one with 242 elements of item_emb (two table
    # rows, 3 % of it: items drawn as positive AND negative of one position, whose two contributions cancel) up to 1.13e-4 -- a
    # different tensor each time (a third run: two rows of the 21-row time_emb, 7 % of it, up to 1.10e-4), none of them off in
    # (1): whole rows or columns whose gradient is orders below the tensor's largest, where Adam's normalisation amplifies what (1)
    # bounds relative to that largest.  The bounds: three quarters of every tensor within 5e-6, no element beyond 5e-4 (a sixth of
    # what three Adam steps can move one); the sharp statement about the exchange is (1).
    for k in ref:
        if k.endswith(".bk"):
            continue                                           # zero-gradient direction, see test_e2e_gpu
        d = np.abs(res[0][1][k] - ref[k])
        assert np.quantile(d, 0.75) <= 5e-6 and d.max() <= 5e-4, (k, float(np.quantile(d, 0.75)), float(d.max()), int((d > 5e-6).sum()))
    loss_one = one.loss_auc()[0]
    st = res[0][2]
    assert st[0] / st[2] == pytest.approx(loss_one, rel=1e-5)  # bucket tail: loss_sum, auc_sum, n_target of the WHOLE batch
    assert st[2] == float((make_batch(STEPS - 1, items)[1] != 0).sum())


@pytest.mark.parametrize("D_,H_", [(20, 1), (128, 4)])
def test_row_buffers_do_not_depend_on_the_sharding(D_, H_):
    """Every activation and gradient row of a step (every [M, D] / [3, M, D] buffer of the engine) holds the same BITS whether the
    batch is one engine's or split over two replicas with batch_global / row_offset: dropout keyed by the global row, nothing
    in a row's arithmetic depends on its neighbours in the batch.  What data parallelism changes is only the order of the sums
    over rows (slabs, the bucket all-reduce)."""
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    hp = lambda: E.Hyper(maxlen=T, hidden_units=D_, num_blocks=2, num_heads=H_, dropout_rate=0.2, max_bins=10, seed=4)
    seq, pos, neg, time, z, _ = make_batch(0)
    whole = E.Engine("cast_1", USERS, ITEMS, hp(), B, training=True, n_slabs=8)
    parts = []
    for r in range(2):
        e = E.Engine("cast_1", USERS, ITEMS, hp(), B // 2, training=True, n_slabs=8, batch_global=B, row_offset=r * (B // 2) * T)
        e.P.copy_(whole.P)
        parts.append(e)
    whole.set_batch(seq, pos, neg, time, z, z)
    whole.launch_step(apply=False)
    for r, e in enumerate(parts):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        e.set_batch(seq[sl], pos[sl], neg[sl], time[sl], z[sl], z[sl])
        e.launch_step(apply=False)
    torch.cuda.synchronize()
    M, Mh, checked = B * T, B * T // 2, 0
    for name, a in whole._bufs.items():
        if a.dim() != 2 or a.shape[0] not in (M, 3 * M):
            continue
        for r, e in enumerate(parts):
            b = e._bufs.get(name)
            if b is None:
                continue
            for k in range(a.shape[0] // M):
                assert torch.equal(a[k * M + r * Mh: k * M + (r + 1) * Mh], b[k * Mh: (k + 1) * Mh]), (name, r, k)
                checked += 1
    assert checked >= 20


def _rccl_worker(rank, world, port, q):
    import torch.distributed as dist
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    from castrec_amd.dist import DataParallel, EngineReplica, shard_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    lo, hi = shard_rows(B, rank, world)
    eng = E.Engine("cast_1", USERS, ITEMS, hyper(E), hi - lo, training=True, n_slabs=8, batch_global=B, row_offset=lo * T)
    rep = EngineReplica(eng, use_graph=True)
    dp = DataParallel(rep, rank, world, sparse=(rank >= 0 and os.environ.get("CASTREC_TEST_SPARSE") == "1"))
    for s in range(STEPS):
        dp.step(make_batch(s))
    torch.cuda.synchronize()
    q.put((rank, {k: v.cpu().numpy() for k, v in eng.get_params().items()}, dist.get_backend(), dist.get_world_size()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("sparse", ["0", "1"])
def test_two_gpus_over_rccl(sparse, monkeypatch):
    """The same comparison with one rank per GPU and the exchange on RCCL (torch.distributed 'nccl'): needs two
    visible devices, so it is skipped on the one-GPU test box and runs wherever the driver has a multi-GPU node."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    monkeypatch.setenv("CASTREC_TEST_SPARSE", sparse)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] == "nccl" and res[0][3] == 2
    one = E.Engine("cast_1", USERS, ITEMS, hyper(E), B, training=True, n_slabs=8)
    for s_ in range(STEPS):
        one.train_step(*make_batch(s_))
    torch.cuda.synchronize()
    ref = {k: v.cpu().numpy() for k, v in one.get_params().items()}
    for k in ref:
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)
        if not k.endswith(".bk"):
            d = np.abs(res[0][1][k] - ref[k])                  # (tail elements: see the one-card test above)
            # (the count of such elements is a small-number statistic: 3 of 16 384 seen at D = 128 -- allowed: 1 + 5e-4 of the tensor;
        #  the bound that matters is the maximum, 1e-4: thirty times below what three Adam steps could move an element)
        assert d.max() <= 1e-4 and (d > 5e-6).sum() <= 1 + 5e-4 * d.size, (k, float(d.max()), int((d > 5e-6).sum()))


@pytest.mark.parametrize("sparse,one_graph", [(False, False), (True, False), (False, True), (True, True)])
def test_graph_resident_step_with_the_collectives_on_one_rank(sparse, one_graph):
    """The production step structure of data parallelism (castrec_amd.dist.DataParallel.step_phases: graph A1 -> table exchange
    started -> graph A2 -> small all-reduce -> exchange finished -> graph B = Adam) over RCCL with ONE rank on the box's card
    (force_collectives), against the plain single-graph step: same parameters after three steps up to the order in which the
    dense slabs are summed (cr_reduce_slabs + Adam on the flat bucket against Adam's own slab sum).  sparse: the item rows
    travel through cr_rows_pack -> all-gather -> cr_rows_add (packed, zeroed in the table, added back: exact)."""
    import torch.distributed as dist
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    from castrec_amd.dist import DataParallel, EngineReplica
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        items = 5000
        a = E.Engine("cast_1", USERS, items, hyper(E), B, training=True, n_slabs=8)
        b = E.Engine("cast_1", USERS, items, hyper(E), B, training=True, n_slabs=8)
        perturb_start(a)
        b.P.copy_(a.P)
        assert 0 < b.bwd_table_done < len(b.bwd)               # something of the backward is left to run beside the exchange
        a.capture()
        a.set_step(1); a.Mom.zero_(); a.Vel.zero_(); a.Gflat.zero_()
        rep = EngineReplica(b, use_graph=True)
        dp = DataParallel(rep, 0, 1, sparse=sparse, force_collectives=True)
        assert dp.sparse == sparse
        b.set_step(1); b.Mom.zero_(); b.Vel.zero_(); b.Gflat.zero_()
        if one_graph:                                          # the whole step, collectives included, as ONE HIP graph
            # captured and validated inside the first step, on that step's batch (an all-padding batch would move nothing and the
            # validation -- one replayed step against one eager step from the same state -- would refuse it)
            dp.request_capture()
        for st in range(STEPS):
            batch = make_batch(st, items)
            a.train_step(*batch)
            dp.step(batch)
        torch.cuda.synchronize()
        if one_graph:
            assert dp.step_form == "one graph", dp.step_form_why
            assert dp._replay_report["moved"] and dp._replay_report["matches_eager"], dp._replay_report
        pa, pb = a.get_params(), b.get_params()
        for k in pa:
            if k.endswith(".bk"):
                continue
            d = (pa[k] - pb[k]).abs().max().item()
            assert d <= 1e-5, (k, d)
        assert a.loss_auc()[0] == pytest.approx(b.loss_auc()[0], rel=1e-5)
    finally:
        dist.destroy_process_group()



def test_whole_step_graph_over_the_id_ring_on_one_rank():
    """bench.py's data-parallel form: the batches wait in the id ring (slot = ids + occurrence index), a step ends by moving the next
    slot's ids into the static id buffers, and the whole step -- collectives included -- is ONE HIP graph that capture_step validates
    by replaying one step against an eager step from a state snapshot.  Round 5: the snapshot did not hold the id buffers, the replay
    ran on the NEXT batch, the validation refused the graph and bench.py's forced-collectives run silently measured the three-graph
    form.  One rank over RCCL (force_collectives), the engine's own slab count (so the prediction head rides the last forward launch):
    the graph must be accepted and three steps must equal the plain engine's on the same batches."""
    import torch.distributed as dist
    import castrec_amd  # noqa: F401
    from castrec_amd import engine as E
    from castrec_amd.dist import DataParallel, EngineReplica
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        items = 5000
        a = E.Engine("cast_1", USERS, items, hyper(E), B, training=True)
        b = E.Engine("cast_1", USERS, items, hyper(E), B, training=True)
        perturb_start(a)
        b.P.copy_(a.P)
        batches = [make_batch(st, items) for st in range(4)]
        staged = torch.from_numpy(np.stack([b.pack_slot(*bt) for bt in batches])).cuda()
        b.use_id_ring(staged)
        b.load_slot(staged[0])
        dp = DataParallel(EngineReplica(b, use_graph=True), 0, 1, force_collectives=True)
        assert dp.capture_step(), dp.step_form_why
        assert dp.step_form == "one graph" and dp._replay_report["matches_eager"], (dp.step_form_why, dp._replay_report)
        b.set_step(1); b.Mom.zero_(); b.Vel.zero_(); b.Gflat.zero_()
        b.load_slot(staged[b.step_number() % len(batches)])
        a.set_step(1); a.Mom.zero_(); a.Vel.zero_(); a.Gflat.zero_()
        for st in range(3):
            a.train_step(*batches[(st + 1) % len(batches)])              # (slot = step number mod slots; the first step is number 1)
            dp.step_phases()
        torch.cuda.synchronize()
        pa, pb = a.get_params(), b.get_params()
        for k in pa:
            if not k.endswith(".bk"):
                assert (pa[k] - pb[k]).abs().max().item() <= 1e-5, k
    finally:
        dist.destroy_process_group()
