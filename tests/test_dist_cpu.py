"""world_size-2 / 4 / 8 gloo tests of the data-parallel wrapper (castrec_amd.dist) on CPU.

The replica is oracle-backed (the HIP engine needs a GPU): it produces UN-normalised gradients of its
row shard in the flat bucket layout the engine uses, so the test covers row sharding, the single flat
all-reduce carrying gradients + loss statistics, global-target-count normalisation and the Adam hand-off.
Result must equal a single-process step on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import castrec_amd  # noqa: F401
from castrec_amd.dist import DataParallel, shard_rows
from oracle import fpmodel as fm

MODEL = "cast_1"
HP = fm.Hyper(maxlen=10, hidden_units=8, num_blocks=1, num_heads=2, dropout_rate=0.0, max_bins=6, lr=1e-2)
ITEMS = 20


class OracleReplica:
    def __init__(self, seed=0, itemnum=ITEMS):
        self.itemnum = itemnum
        self.P = fm.init_params(MODEL, 7, itemnum, HP, seed=seed)
        self.names = ["item_emb"] + sorted(k for k in self.P if k != "item_emb")     # the item table leads the bucket, as in the engine
        self.sizes = [self.P[k].numel() for k in self.names]
        self.vec = torch.cat([self.P[k].reshape(-1) for k in self.names])
        self.bucket = torch.zeros(self.vec.numel() + 4, dtype=torch.float64)
        self.opt = fm.AdamTF(self.P, lr=HP.lr)
        self.shard_ids = torch.zeros(3 * 4 * HP.maxlen, dtype=torch.int64)

    def param_vector(self):
        return self.vec

    def _sync_from_vec(self):
        off = 0
        for k, n in zip(self.names, self.sizes):
            self.P[k] = self.vec[off:off + n].reshape(self.P[k].shape).clone(); off += n

    def sparse_spec(self):
        return dict(n_item=(self.itemnum + 1) * HP.hidden_units, D=HP.hidden_units, n_slots=3 * self.shard_ids.numel() // 3,
                    ids=lambda: self.shard_ids)

    def backward_to_flat(self, shard):
        self.shard_ids = torch.tensor(np.concatenate([np.asarray(a).reshape(-1) for a in shard[:3]]))
        self._sync_from_vec()
        out, G = fm.loss_and_grads(MODEL, self.P, HP, fm.to_batch(*shard))
        n = float(out["istarget"].sum())
        self.bucket[:-4] = torch.cat([G[k].reshape(-1) for k in self.names]) * n     # un-normalised
        self.bucket[-4] = float(out["loss"]) * n
        self.bucket[-3] = float(out["auc"]) * n
        self.bucket[-2] = n
        return self.bucket

    def adam_from_flat(self):
        n = float(self.bucket[-2])
        off, G = 0, {}
        for k, sz in zip(self.names, self.sizes):
            G[k] = (self.bucket[off:off + sz] / n).reshape(self.P[k].shape); off += sz
        self.P = self.opt.step(self.P, G)
        self.vec = torch.cat([self.P[k].reshape(-1) for k in self.names])
        self.loss = float(self.bucket[-4]) / n


def make_batch(B=8, T=10, seed=3, itemnum=ITEMS):
    rs = np.random.RandomState(seed)
    seq = rs.randint(1, itemnum + 1, (B, T)); pos = rs.randint(1, itemnum + 1, (B, T)); neg = rs.randint(1, itemnum + 1, (B, T))
    for b in range(B):
        n = rs.randint(0, T - 2)
        seq[b, :n] = 0; pos[b, :n] = 0; neg[b, :n] = 0          # ragged: ranks see different target counts
    time = rs.randint(0, 7, (B, T)) * (seq != 0)
    z = np.zeros_like(seq)
    return seq, pos, neg, time, z, z


def _worker(rank, world, port, q, itemnum=ITEMS, sparse=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rep = OracleReplica(seed=rank, itemnum=itemnum)   # different init per rank: the wrapper must broadcast rank 0's
    dp = DataParallel(rep, rank, world, sparse=sparse)
    assert dp.sparse == bool(sparse)
    batch = make_batch(itemnum=itemnum)
    for _ in range(2):
        dp.step(batch)
    q.put((rank, rep.param_vector().numpy().copy(), rep.loss))
    dist.destroy_process_group()


def test_shard_rows():
    assert [shard_rows(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(ValueError):
        shard_rows(10, 0, 4)


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,itemnum,sparse", [(2, ITEMS, False), (2, ITEMS, True), (2, 3000, True),
                                                  (4, ITEMS, False), (4, 3000, True), (8, ITEMS, False), (8, ITEMS, True), (8, 3000, True)])
def test_n_rank_step_equals_single_process(world, itemnum, sparse):
    """dense: one flat all-reduce.  sparse: all-gather of de-duplicated (row id, gradient row) pairs for the item table
    (hot rows shared by the ranks at 20 items; V = 3001 >> 240 touched rows: the C5 regime) + all-reduce of the rest.
    World 2, 4 and 8 (one batch row per rank at 8): the rank-ordered sums of the sparse exchange and the shard arithmetic
    at every rank count the node has."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, itemnum, sparse)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=200) for _ in procs])
    for p in procs:
        p.join(timeout=30)
    ref = OracleReplica(seed=0, itemnum=itemnum)
    one = DataParallel(ref, 0, 1)
    batch = make_batch(itemnum=itemnum)
    for _ in range(2):
        one.step(batch)
    # (more ranks = another association of the fp64 shard sums; the key bias' gradient is identically 0 and Adam turns its
    #  1e-17 rounding residue into moves of up to ~1e-11 -- DESIGN section 2, "known ill-conditioned quantity")
    for rank, vec, loss in res:
        np.testing.assert_allclose(vec, ref.param_vector().numpy(), rtol=0, atol=1e-12 if world == 2 else 1e-10)
        assert loss == pytest.approx(ref.loss, rel=1e-12)
    for r in res[1:]:
        assert np.array_equal(res[0][1], r[1])                    # replicas bit-identical (fixed summation order)


class _StubReplica:
    def __init__(self):
        self.vec = torch.zeros(16, dtype=torch.float64)

    def param_vector(self):
        return self.vec


class _StubGraph:
    """Stands for the captured whole-step graph: `reduces` = its collective really runs at replay."""
    def __init__(self, rep, rank, reduces):
        self.rep, self.rank, self.reduces = rep, rank, reduces

    def launch(self):
        g = torch.full((16,), float(self.rank + 1), dtype=torch.float64)
        if self.reduces:
            dist.all_reduce(g)
        self.rep.vec -= 0.1 * g


def _validate_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for reduces in (True, False):
        rep = _StubReplica()
        dp = DataParallel(rep, rank, world)
        dp._step_graph = _StubGraph(rep, rank, reduces)
        ok = dp._replayed_step_agrees()
        out.append((ok, float(rep.vec.abs().max())))          # the parameters are put back either way
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_whole_step_graph_is_validated_by_one_replayed_step():
    """More than one rank: a captured step whose collective does not run at replay leaves the replicas apart after one update --
    every rank gets the same verdict (no rank alone on the other form), the parameters are restored."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_validate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in procs])
    for p in procs:
        p.join(timeout=30)
    for rank, out in res:
        assert out[0] == (True, 0.0), (rank, out)
        assert out[1] == (False, 0.0), (rank, out)


class _PhasedReplica:
    """A replica with phases and a "graph" (a closure that re-runs what was captured): what DataParallel.capture_step drives."""
    has_phases = True

    def __init__(self):
        self.vec = torch.zeros(16, dtype=torch.float64)
        self.buck = torch.zeros(20, dtype=torch.float64)
        self.batch = None

    def param_vector(self):
        return self.vec

    def set_batch(self, shard):
        self.batch = shard

    def bucket(self):
        return self.buck

    def sparse_spec(self):
        return dict(n_item=8, D=4, n_slots=4, ids=lambda: torch.zeros(4, dtype=torch.int64))

    def phase(self, i, eager=False):
        if i == 0:
            self.buck[:] = float(np.sum(self.batch[0]))      # this shard's "gradient": zero for an all-padding shard
        elif i == 2:
            self.vec -= 0.1 * self.buck[:16]

    def capture(self, fn):
        class _G:
            def launch(self_):
                fn()
        return _G()

    def snapshot(self):
        return dict(vec=self.vec.clone(), buck=self.buck.clone())

    def restore(self, snap):
        self.vec.copy_(snap["vec"]); self.buck.copy_(snap["buck"])


def _capture_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CASTREC_DP_VERBOSE="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for first in ("padding", "real"):
        rep = _PhasedReplica()
        dp = DataParallel(rep, rank, world, sparse=False)
        dp.request_capture()
        assert dp._step_graph is None                      # nothing is captured before a batch exists
        B = 2 * world
        ids = np.zeros((B, 4), np.int64) if first == "padding" else np.arange(1, 4 * B + 1).reshape(B, 4)
        dp.step((ids,) * 6)
        out.append((dp.step_form, dp.step_form_why, dp._step_graph is not None, rep.vec.clone().numpy()))
        dp.step((np.arange(1, 4 * B + 1).reshape(B, 4),) * 6)
        out[-1] += (rep.vec.clone().numpy(),)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_capture_waits_for_a_real_batch_and_says_which_form_runs():
    """request_capture() defers the whole-step capture into the first step(): validated on THAT batch.  A first batch of padding
    only moves no parameter, the validation refuses it and the three-graph form runs -- visibly (step_form / step_form_why), on every
    rank alike; a real batch validates and the replayed graph gives the same parameters as the phases."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_capture_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in procs])
    for p in procs:
        p.join(timeout=30)
    total = float(np.arange(1, 17).sum())                      # both shards' "gradients", all-reduced
    for rank, out in res:
        form, why, has_graph, v1, v2 = out[0]
        assert form == "three graphs" and "'moved': False" in why and not has_graph, out[0][:3]
        assert np.all(v1 == 0.0) and np.allclose(v2, -0.1 * total)          # the rehearsals left no trace; the next step is a plain one
        form, why, has_graph, v1, v2 = out[1]
        assert form == "one graph" and has_graph, out[1][:3]
        assert np.allclose(v1, -0.1 * total) and np.allclose(v2, -0.2 * total)


def test_exchange_picks_the_cheaper_form():
    from castrec_amd.dist import dense_allreduce_bytes, sparse_exchange_bytes
    # C3 (Beauty, V = 57 290, D = 64, B 128 x T 50 per rank, 8 ranks): dense 25.7 MB < sparse 34.9 MB
    assert dense_allreduce_bytes(57290 * 64, 8) < sparse_exchange_bytes(3 * 128 * 50, 64, 8)
    # C5 (V = 10 M, D = 256, B 128 x T 512 per rank): sparse 1.4 GB << dense 17.9 GB
    assert sparse_exchange_bytes(3 * 128 * 512, 256, 8) < 0.1 * dense_allreduce_bytes(10_000_001 * 256, 8)
