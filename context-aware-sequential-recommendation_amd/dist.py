"""Data parallelism: one process per GPU, full parameter replica per rank, per-user minibatch rows
sharded over ranks, ONE flat all-reduce per step (RCCL over xGMI through torch.distributed 'nccl').

The reference has no distributed code; the scheme follows SURVEY 8e:
  * the global batch is one batch of the single reference sampler stream; rank r takes rows
    [r*B/G, (r+1)*B/G) -- an N-GPU run sees exactly the inputs of a 1-GPU run with batch B_global;
  * dropout masks are keyed by the GLOBAL row index, so results do not depend on G;
  * the loss is normalised by the number of targets of the WHOLE batch (sasrec.py:104-108): ranks
    produce un-normalised gradients + their local target count, both travel in the same bucket and
    Adam divides by the reduced count;
  * bucket = [item/pos table grads | dense grads | loss_sum, auc_sum, n_target, pad] -- 0.9 MB for
    SASRec/CAST at D=50: one latency-bound ring all-reduce per step (xGMI links are point to point, so
    many small collectives would each pay the ring latency);
  * large item tables (config C5: 10 M x 256 fp32 = 10.24 GB) are NOT all-reduced densely: a step
    touches at most 3*B_local*T rows per rank (seq, pos, neg ids), so the ranks all-gather their de-duplicated
    (row id, gradient row) pairs and every rank adds them into its (otherwise zero) table gradient in rank
    order -- the same sums in the same order everywhere, so replicas stay bit-identical -- while the small
    rest of the bucket (positional table, dense gradients, loss statistics) takes the one all-reduce.
    Chosen per engine from the byte counts (sparse_exchange_bytes vs dense_allreduce_bytes below).

`Replica` is the minimal protocol the wrapper needs; castrec_amd.engine.Engine implements it on the GPU,
tests/test_dist_cpu.py drives the same wrapper over gloo with an oracle-backed replica."""
import os

import torch


def shard_rows(batch_global, rank, world):
    """Rows [lo, hi) of the global batch owned by `rank` (requires world | batch_global)."""
    if batch_global % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (batch_global, world))
    per = batch_global // world
    return rank * per, (rank + 1) * per


def init_from_env(backend=None):
    """torch.distributed bootstrap from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            if torch.cuda.is_available():              # gloo over host memory (HostBounce): ranks may share a card
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    return rank, local_rank, world


def dense_allreduce_bytes(n_floats, world):
    """Bytes a rank sends in a ring all-reduce of n_floats fp32 (reduce-scatter + all-gather)."""
    return 2.0 * (world - 1) / world * n_floats * 4


def sparse_exchange_bytes(n_slots, D, world):
    """Bytes a rank receives in the all-gather of n_slots (int32 row id, fp32 gradient row of D) pairs per rank."""
    return (world - 1) * n_slots * (4 + 4 * D)


class RowExchange:
    """Device form of exchange_table_rows (HIP kernels cr_rows_pack / cr_rows_add, csrc/cr_dist.hip): pack this rank's touched
    rows into fixed slots (de-duplicated through a row-flag table, no sort), ONE all-gather of [n, D + 1] floats (the row id
    travels as the int bits of column 0), then one add launch per rank, in rank order.  start() may be issued while the rest of
    the backward still runs (the item rows are final after the last embedding backward); finish() adds the gathered rows."""

    def __init__(self, V, D, n_slots, world, device, group=None):
        from . import lib as L
        self.L, self.V, self.D, self.n, self.world, self.group = L, V, D, n_slots, world, group
        self.flags = torch.zeros(V, dtype=torch.int32, device=device)
        self.packed = torch.empty(n_slots, D + 1, dtype=torch.float32, device=device)
        self.gathered = torch.empty(world, n_slots, D + 1, dtype=torch.float32, device=device)
        self.tag = torch.zeros(1, dtype=torch.int32, device=device)      # device word: advanced on the stream, so a graph replay advances it too
        self.work = None

    def start(self, table, ids):
        import torch.distributed as dist
        assert table.is_cuda and table.is_contiguous() and ids.dtype == torch.int32 and ids.numel() == self.n
        self.tag.add_(1)                                # a tag no flag holds yet (flags start at 0, tags at 1)
        st = torch.cuda.current_stream().cuda_stream
        self.L.call("cr_rows_pack", table.data_ptr(), ids.data_ptr(), self.n, self.D, self.V, self.flags.data_ptr(), self.tag.data_ptr(),
                    self.packed.data_ptr(), 1, st)
        self.work = dist.all_gather_into_tensor(self.gathered.view(-1), self.packed.view(-1), group=self.group, async_op=True)
        self._table = table

    def finish(self):
        self.work.wait()                                # the current stream waits for the all-gather
        st = torch.cuda.current_stream().cuda_stream
        for r in range(self.world):                     # fixed order: the sums match on every rank
            self.L.call("cr_rows_add", self._table.data_ptr(), self.gathered[r].data_ptr(), self.n, self.D, self.V, st)
        self.work = None
        return self.gathered[:, :, 0].view(torch.int32)                  # [world, n] row ids of every rank (0 = empty slot)


def exchange_table_rows(table, ids, world, group=None):
    """Sums over ranks the rows of `table` ([V, D] view of this rank's table-gradient bucket) that this step touched.
    ids: 1-D integer tensor of this rank's contributing row ids (duplicates and the zero-pad id 0 allowed; every row
    outside it must be zero).  On return `table` equals what a dense all-reduce would have produced, bit-identical on
    all ranks: a row's partial sums are added in rank order and a rank's list holds a row once.
    (Host tensors -- the gloo transport of the CPU tests and of ranks sharing a card; device tensors go through RowExchange.)"""
    import torch.distributed as dist
    srt = torch.sort(ids.to(torch.int64)).values
    first = torch.ones_like(srt, dtype=torch.bool)
    first[1:] = srt[1:] != srt[:-1]
    idu = torch.where(first, srt, torch.zeros_like(srt))          # later duplicates -> row 0, whose gradient is always zero
    rows = table.index_select(0, idu) * first.unsqueeze(1).to(table.dtype)
    ids32 = idu.to(torch.int32)
    all_ids = [torch.empty_like(ids32) for _ in range(world)]
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(all_ids, ids32, group=group)
    dist.all_gather(all_rows, rows, group=group)
    table.index_fill_(0, idu, 0.0)                                # this rank's share travels in `rows` like everyone's
    for r in range(world):                                        # fixed order: the sums match on every rank
        table.index_add_(0, all_ids[r].to(torch.int64), all_rows[r])
    return all_ids


class DataParallel:
    """Drives one replica per rank: local backward -> gradient exchange -> identical Adam on every rank.

    sparse: None = choose from the byte counts when the replica offers sparse_spec(); True / False force it."""

    def __init__(self, replica, rank, world, process_group=None, sparse=None, force_collectives=False):
        """force_collectives: run the collectives with a single rank as well (bench.py CASTREC_FORCE_DIST=1: the cost of the
        data-parallel step structure on one GPU)."""
        self.replica, self.rank, self.world, self.pg = replica, rank, world, process_group
        self.force = bool(force_collectives)
        self.sparse = False
        self._rows = None
        self._step_graph = None
        spec = replica.sparse_spec() if hasattr(replica, "sparse_spec") else None
        if world == 1 and self.force and spec is not None and sparse:
            self.sparse = True
        if world > 1 and spec is not None and sparse is not False:
            dense = dense_allreduce_bytes(spec["n_item"], world)
            sp = sparse_exchange_bytes(spec["n_slots"], spec["D"], world)
            self.sparse = True if sparse else (sp < 0.5 * dense)   # only when it at least halves the traffic
            self.exchange_bytes = dict(dense_allreduce=dense, sparse_allgather=sp)
        if world > 1 and getattr(replica, "lazy_adam", False) and not self.sparse:
            raise ValueError("lazy Adam under data parallelism needs the sparse table exchange (every replica must update the same rows)")
        if world > 1:
            import torch.distributed as dist
            dist.broadcast(replica.param_vector(), 0, group=process_group)       # same start everywhere

    def request_capture(self, validate=True):
        """Asks for the whole-step graph (capture_step).  The capture -- and the one rehearsed and one replayed step that validate
        it -- run inside the next step(), once that step's batch sits in the replica's static id buffers: a capture made before
        any batch exists would rehearse an all-padding batch, whose step moves no parameter and whose replicas agree trivially."""
        self._capture_pending = bool(validate) if validate else None
        self._capture_wanted = True

    def step(self, batch_global):
        """batch_global: tuple of [B_global, T] int arrays (seq, pos, neg, time, hours, days)."""
        lo, hi = shard_rows(len(batch_global[0]), self.rank, self.world)
        shard = tuple(a[lo:hi] for a in batch_global)
        if getattr(self.replica, "has_phases", False):
            self.replica.set_batch(shard)
            if getattr(self, "_capture_wanted", False):
                self._capture_wanted = False
                self.capture_step(validate=self._capture_pending is not None)
            self.step_phases()
            return
        bucket = self.replica.backward_to_flat(shard)
        self.exchange(bucket)
        self.replica.adam_from_flat()

    def step_phases(self, eager=False):
        """Graph-resident step of a device replica (EngineReplica with use_graph): graph A1 {forward, backward up to the last
        launch that adds to the table gradient} -> the table's exchange STARTS (asynchronous collective: it waits for A1, the
        stream goes on) -> graph A2 {rest of the backward, slab collapse} -> all-reduce of the small part of the bucket -> the
        table's exchange is waited for (sparse: the gathered rows are added, one launch per rank) -> graph B {Adam}."""
        import torch.distributed as dist
        if self._step_graph is not None and not eager:
            self._step_graph.launch()
            return
        rep = self.replica
        bucket = rep.bucket()
        rep.phase(0, eager)
        live = self.world > 1 or self.force
        w = None
        if live:
            spec = rep.sparse_spec()
            n_item, D = spec["n_item"], spec["D"]
            if self.sparse:
                if self._rows is None:
                    self._rows = RowExchange(n_item // D, D, spec["n_slots"], self.world, bucket.device, self.pg)
                self._rows.start(bucket[:n_item].view(-1, D), spec["ids"]())
            else:
                w = dist.all_reduce(bucket[:n_item], group=self.pg, async_op=True)
        rep.phase(1, eager)
        if live:
            dist.all_reduce(bucket[n_item:], group=self.pg)                       # positional table, dense grads, loss statistics
            if self.sparse:
                all_ids = self._rows.finish()
                if hasattr(rep, "set_lazy_ids"):
                    rep.set_lazy_ids(all_ids.reshape(-1))
            else:
                w.wait()
        rep.phase(2, eager)

    def capture_step(self, validate=True):
        """Tries to capture the WHOLE step -- the three phases and the collectives between them -- into one HIP graph (the
        collectives of torch.distributed's RCCL backend are stream-ordered and capturable; the sparse exchange's tag is a device
        word).  One graph launch per step instead of three plus the host side of two collectives.

        Must run with a REAL batch in the replica's static id buffers (request_capture() defers it into the next step()).  The
        replica's state (parameters, Adam moments, step counter, gradient bucket) is saved first and put back at the end, so the
        rehearsals leave no trace: (1) one eager step on that batch -- every lazy initialisation (communicator, RowExchange
        buffers) happens here -- whose parameters are the reference; (2) the capture; (3) with `validate`, ONE replayed step from
        the same state, which must reproduce (1) and leave the replicas identical (_replayed_step_agrees).  Returns False, and
        keeps the three-graph form, when the capture is refused or the replay does not agree; `step_form` / `step_form_why` say
        which form runs, and why."""
        rep = self.replica
        self.step_form, self.step_form_why = "three graphs", "no phases to capture"
        if not getattr(rep, "has_phases", False) or not hasattr(rep, "capture"):
            return False
        snap = rep.snapshot() if hasattr(rep, "snapshot") else None
        captured, ref = True, None
        try:
            self.step_phases(eager=True)
            if rep.param_vector().is_cuda:
                torch.cuda.synchronize()
            ref = rep.param_vector().clone()
            if snap is not None:
                rep.restore(snap)
            self._step_graph = rep.capture(lambda: self.step_phases(eager=True))
        except Exception as e:                           # noqa: BLE001 -- whatever refuses the capture: the other form runs
            self._step_graph = None
            captured = False
            self.step_form_why = "capture refused: %s" % e
        if self.world > 1:
            # one verdict for all ranks: a rank alone on the other form would leave the rest waiting in a collective
            import torch.distributed as dist
            flag = torch.tensor([1.0 if captured else 0.0], device=self.replica.param_vector().device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            captured = bool(flag.item() > 0.5)
        ok = captured
        if ok and validate:
            if snap is not None:
                rep.restore(snap)
            ok = self._replayed_step_agrees(ref)
        if snap is not None:
            rep.restore(snap)
        if not ok:
            self._step_graph = None
            if captured:
                self.step_form_why = "the replayed step did not agree: %s" % getattr(self, "_replay_report", "?")
        else:
            self.step_form, self.step_form_why = "one graph", ("validated by one replayed step" if validate else "not validated")
        if os.environ.get("CASTREC_DP_VERBOSE", "1") != "0" and self.rank == 0:
            import sys
            print("[castrec_amd.dist] data-parallel step: %s (%s)" % (self.step_form, self.step_form_why), file=sys.stderr)
        return ok

    def _replayed_step_agrees(self, ref=None, bitwise=None):
        """One REPLAYED step of the whole-step graph on every rank, from the state the caller restored.  It must (a) move the
        parameters, (b) leave them finite, (c) reproduce `ref` -- the parameters one eager step on the same batch and state gave:
        bit for bit where the replica's step is reproducible (`bitwise`; default: the replica's own `bitwise_reproducible`), else
        in all but a few elements (float atomics reorder the table gradient's last bits, and Adam's first steps turn a sign flip
        of a ~0 gradient into a move of 2 lr; a collective that did not run or reduced a stale buffer moves nearly EVERY
        element) -- and (d) leave the replicas identical over the ranks (sum and sum of squares all-gathered and compared).
        The collective sequence is the same on every rank whatever happens locally: a launch that raises contributes NaN, then
        all_gather and the verdict's all_reduce run all the same (a rank that skipped one would hang its peers)."""
        p = self.replica.param_vector()
        before = p.clone()
        launched = True
        try:
            self._step_graph.launch() if hasattr(self._step_graph, "launch") else self._step_graph.replay()
            if p.is_cuda:
                torch.cuda.synchronize()
        except Exception:                                # noqa: BLE001
            launched = False
        report = {}
        report["launched"] = launched
        report["moved"] = launched and bool((p != before).any())
        report["finite"] = launched and bool(torch.isfinite(p).all())
        if ref is not None and launched:
            if bitwise is None:
                # (bit for bit only with one rank: with more, the eager and the captured collectives may sum the ranks' buckets in
                #  another order -- channels, algorithm -- which is rounding, not a fault)
                bitwise = bool(getattr(self.replica, "bitwise_reproducible", False)) and self.world == 1
            if bitwise:
                report["matches_eager"] = bool(torch.equal(p, ref))
            else:
                scale = float(ref.abs().max()) + 1e-30
                report["off_fraction"] = float(((p - ref).abs() > 1e-6 * max(scale, 1.0)).double().mean())
                report["matches_eager"] = report["off_fraction"] < 1e-3
        ok = all(v for k, v in report.items() if isinstance(v, bool))
        chk = torch.stack([p.double().sum(), (p.double() ** 2).sum()]) if launched else torch.full((2,), float("nan"), dtype=torch.float64, device=p.device)
        if self.world > 1:
            import torch.distributed as dist
            got = [torch.empty_like(chk) for _ in range(self.world)]
            dist.all_gather(got, chk, group=self.pg)
            report["replicas_identical"] = all(bool(torch.isfinite(g).all()) and bool(torch.equal(g, got[0])) for g in got)
            ok = ok and report["replicas_identical"]
            flag = torch.tensor([1.0 if ok else 0.0], device=p.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)      # one verdict for all ranks (a rank alone on the other form would hang the rest)
            ok = bool(flag.item() > 0.5)
        self._replay_report = report
        p.copy_(before)                                  # (callers with a snapshot restore everything; this covers the others)
        return ok

    def exchange(self, bucket):
        """Sums `bucket` ([item table grads | small part]) over the ranks, in place."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        if self.sparse:
            spec = self.replica.sparse_spec()
            n_item, D = spec["n_item"], spec["D"]
            all_ids = exchange_table_rows(bucket[:n_item].view(-1, D), spec["ids"]().to(bucket.device), self.world, self.pg)
            if hasattr(self.replica, "set_lazy_ids"):
                self.replica.set_lazy_ids(torch.cat(all_ids))                     # row-sparse Adam: the rows ANY rank touched
            dist.all_reduce(bucket[n_item:], group=self.pg)                       # positional table, dense grads, loss statistics
        else:
            dist.all_reduce(bucket, group=self.pg)                                # sum of grads and of loss statistics


class EngineReplica:
    """Adapter: castrec_amd.engine.Engine as a DataParallel replica (optionally replaying a HIP graph)."""

    def __init__(self, engine, use_graph=True):
        self.e = engine
        self.has_phases = bool(use_graph)
        if use_graph:
            engine.capture_dp_phases()
            engine.graph = None

    def set_batch(self, shard):
        self.e.set_batch(*shard)

    def bucket(self):
        return self.e.Gflat

    def phase(self, i, eager=False):
        if eager:
            self.e._run(self.e.dp_progs[i], torch.cuda.current_stream().cuda_stream)
        else:
            self.e.dp_graphs[i].launch()

    def capture(self, fn):
        """HIP graph of fn() (kernel launches and stream-ordered collectives on the current stream)."""
        from . import ops as O
        s0 = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(s0)
        g = O.Graph()
        with torch.cuda.stream(side):
            g.begin()
            try:
                fn()
            finally:
                g.end()
        s0.wait_stream(side)
        self._keep_stream = side
        return g

    def param_vector(self):
        return self.e.P

    def snapshot(self):
        """Everything a step changes (parameters, Adam moments, the state block with the step counter and the loss sums, the
        gradient bucket, the static id buffers, row-sparse Adam's claim flags): capture_step's rehearsals are undone with restore()."""
        e = self.e
        snap = dict(P=e.P.clone(), Mom=e.Mom.clone(), Vel=e.Vel.clone(), state=e.state.clone(), Gflat=e.Gflat.clone())
        if getattr(e, "batch_buf", None) is not None:
            # (with the id ring a step ENDS by moving the next slot's ids into the static id buffers: a rehearsal step undone without
            #  them would replay on the next batch -- bench.py's forced-collectives run fell back to three graphs that way, round 5)
            snap["batch_buf"] = e.batch_buf.clone()       # (ids + the batch's occurrence index)
        elif getattr(e, "ids_all", None) is not None:
            snap["ids_all"] = e.ids_all.clone()
        if getattr(e, "lazy_flags", None) is not None:
            snap["lazy_flags"] = e.lazy_flags.clone()
        return snap

    def restore(self, snap):
        e = self.e
        for k, v in snap.items():
            getattr(e, k).copy_(v)

    @property
    def bitwise_reproducible(self):
        return bool(getattr(self.e, "bitwise_reproducible", False))

    def backward_to_flat(self, shard):
        self.e.set_batch(*shard)
        if self.has_phases:
            self.e.dp_graphs[0].launch()
            self.e.dp_graphs[1].launch()
        else:
            self.e.launch_backward_to_flat()
        return self.e.Gflat

    def adam_from_flat(self):
        if self.has_phases:
            self.e.dp_graphs[2].launch()
        else:
            self.e.launch_adam_from_flat()

    @property
    def lazy_adam(self):
        return self.e.lazy_adam

    def set_lazy_ids(self, ids):
        if self.e.lazy_adam:
            self.e.lazy_ids.copy_(ids.to(self.e.lazy_ids.device, torch.int32))

    def sparse_spec(self):
        """The item table leads the bucket; its gradient rows of a step are those of the seq / pos / neg ids
        (embedding backward: modules.py:157; head: sasrec.py:89-90)."""
        e = self.e
        return dict(n_item=(e.itemnum + 1) * e.D, D=e.D, n_slots=3 * e.M, ids=lambda: e.ids_all[:3].reshape(-1))


class HostBounce:
    """Replica adapter whose bucket and parameter vector travel through host memory, for a `gloo` process group: ranks that
    SHARE one card (RCCL refuses two ranks per device: the one-GPU test box) or a host without RCCL.  Everything else is the
    production path -- row shards, global-row dropout keys, un-normalised shard gradients + target counts in one bucket, Adam
    dividing by the reduced count (tests/test_dist_gpu.py, the 2-rank main.py test).  Selected by CASTREC_DIST_BACKEND=gloo."""

    def __init__(self, inner):
        self.inner = inner
        self.host_p = inner.param_vector().cpu()
        self.host_g = None

    def param_vector(self):
        return self.host_p

    def adopt_params(self):
        """after DataParallel's broadcast of the (host) parameter vector: rank 0's parameters onto this card"""
        self.inner.param_vector().copy_(self.host_p)

    def backward_to_flat(self, shard):
        self.host_g = self.inner.backward_to_flat(shard).cpu()
        return self.host_g

    def adam_from_flat(self):
        self.inner.e.Gflat.copy_(self.host_g)
        self.inner.adam_from_flat()

    def sparse_spec(self):
        return self.inner.sparse_spec()

    @property
    def lazy_adam(self):
        return getattr(self.inner, "lazy_adam", False)

    def set_lazy_ids(self, ids):
        if hasattr(self.inner, "set_lazy_ids"):
            self.inner.set_lazy_ids(ids)

