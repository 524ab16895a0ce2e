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
    many small collectives would each pay the ring latency).

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
            dist.init_process_group(backend)
    return rank, local_rank, world


class DataParallel:
    """Drives one replica per rank: local backward -> flat all-reduce -> identical Adam on every rank."""

    def __init__(self, replica, rank, world, process_group=None):
        self.replica, self.rank, self.world, self.pg = replica, rank, world, process_group
        if world > 1:
            import torch.distributed as dist
            dist.broadcast(replica.param_vector(), 0, group=process_group)       # same start everywhere

    def step(self, batch_global):
        """batch_global: tuple of [B_global, T] int arrays (seq, pos, neg, time, hours, days)."""
        lo, hi = shard_rows(len(batch_global[0]), self.rank, self.world)
        bucket = self.replica.backward_to_flat(tuple(a[lo:hi] for a in batch_global))
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(bucket, group=self.pg)                                # sum of grads and of loss statistics
        self.replica.adam_from_flat()


class EngineReplica:
    """Adapter: castrec_amd.engine.Engine as a DataParallel replica (optionally replaying a HIP graph)."""

    def __init__(self, engine, use_graph=True):
        self.e = engine
        if use_graph:
            engine.capture(dp=True)

    def param_vector(self):
        return self.e.P

    def backward_to_flat(self, shard):
        self.e.set_batch(*shard)
        if self.e.graph is not None:
            self.e.graph.launch()
        else:
            self.e.launch_backward_to_flat()
        return self.e.Gflat

    def adam_from_flat(self):
        self.e.launch_adam_from_flat()
