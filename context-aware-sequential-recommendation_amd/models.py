"""Model classes with the reference's constructor / predict surface (models/sasrec.py:5,127-129,
models/cast_N.py) on top of the native engine.

    model = SASRec(usernum, itemnum, args)              # or CAST1(usernum, itemnum, ratingnum, args) ... CAST9
    auc, loss = model.train_step(u, seq, pos, neg, timeseq, hours_seq, days_seq)    # == sess.run([auc, loss, train_op], feed)
    logits, attn = model.predict(sess, u, seq, item_idx, timeseq=, hours_seq=, days_seq=)

``sess`` is accepted and ignored (there is no TensorFlow session).  ``predict`` takes a whole batch of
users; ``item_idx`` is either the reference's shared list of 101 candidates or a [B, n] array.
"""
import os

import numpy as np
import torch

from .engine import MODELS, Engine, Hyper


class _Model(object):
    NAME = None

    def __init__(self, usernum, itemnum, args, reuse=None, n_slabs=None, batch_global=None, row_offset=0):
        self.name = self.NAME
        self.usernum, self.itemnum, self.args = usernum, itemnum, args
        self.hp = Hyper(args)
        self._n_slabs, self._batch_global, self._row_offset = n_slabs, batch_global, row_offset
        self._train = None
        self._pending_opt = None              # optimiser state loaded before the training engine exists (load / load_tf_checkpoint)
        self._eval = {}
        self._fed = []                        # data parallel: batches handed over by feed() wait here
        self._graph = bool(int(os.environ.get("CASTREC_GRAPH", "1")))
        # parameters exist from construction on (tf.global_variables_initializer, main.py:150)
        self._owner = Engine(self.name, usernum, itemnum, self.hp, 1, training=False)
        self.attention_weights = None

    # -- training ---------------------------------------------------------------------------------
    def data_parallel(self, rank, world, process_group=None, sparse=None):
        """One replica per rank (the reference has no multi-GPU path): every rank feeds train_step the SAME
        global batch (one sampler stream, same seed everywhere), takes rows [rank * B / world, (rank + 1) * B / world) of it,
        and the gradients meet in castrec_amd.dist.DataParallel (RCCL all-reduce of one bucket, or the sparse item-row
        exchange for large tables) before an identical Adam step on every rank.  Call before the first train_step."""
        if self._train is not None:
            raise RuntimeError("data_parallel() must be called before the first train_step")
        self._dp_cfg = (int(rank), int(world), process_group, sparse)

    def _train_engine(self, B):
        if self._train is None:
            # the training engine shares the owner's parameter vector: both must live on the device this process runs on
            # (under torch.distributed.run the process group -- which selects cuda:LOCAL_RANK -- comes before build_model)
            cur = torch.device("cuda", torch.cuda.current_device())
            if self._owner.P.device != cur:
                raise RuntimeError("model parameters are on %s but the current device is %s: select the device "
                                   "(castrec_amd.dist.init_from_env / torch.cuda.set_device) before building the model"
                                   % (self._owner.P.device, cur))
            dp_cfg = getattr(self, "_dp_cfg", None)
            if dp_cfg is not None and dp_cfg[1] > 1:
                from . import dist as D_
                rank, world, pg, sparse = dp_cfg
                lo, hi = D_.shard_rows(B, rank, world)
                self._train = Engine(self.name, self.usernum, self.itemnum, self.hp, hi - lo, training=True, share=self._owner,
                                     n_slabs=self._n_slabs, batch_global=B, row_offset=lo * self.hp.maxlen)
                self._dp_rows = (lo, hi, B)
                rep = D_.EngineReplica(self._train, use_graph=self._graph)
                bounce = os.environ.get("CASTREC_DIST_BACKEND") == "gloo"     # gloo cannot reduce device tensors: through host memory
                if bounce:
                    rep = D_.HostBounce(rep)
                self._dp = D_.DataParallel(rep, rank, world, pg, sparse=sparse)
                if bounce:
                    rep.adopt_params()
                if self._graph and not bounce and os.environ.get("CASTREC_DP_ONE_GRAPH") != "0":
                    # the whole step incl. the collectives as one HIP graph: captured and validated inside the FIRST train_step,
                    # on that step's batch (an all-padding batch -- the static id buffers before any set_batch -- moves nothing
                    # and validates nothing); else the three-graph form.  dist.DataParallel.step_form says which one runs.
                    self._dp.request_capture()
                if self._graph:
                    self._train.set_step(1); self._train.Mom.zero_(); self._train.Vel.zero_(); self._train.Gflat.zero_()
                if self._pending_opt is not None:
                    self._apply_opt(*self._pending_opt)
                    self._pending_opt = None
                return self._train
            self._train = Engine(self.name, self.usernum, self.itemnum, self.hp, B, training=True, share=self._owner,
                                 n_slabs=self._n_slabs, batch_global=self._batch_global, row_offset=self._row_offset)
            if self._graph and self._batch_global is None:
                self._train.capture()
                self._train.set_step(1); self._train.Mom.zero_(); self._train.Vel.zero_(); self._train.Gflat.zero_()
            if self._pending_opt is not None:          # a checkpoint's Adam slots and step count (saver.restore, main.py:165-175)
                self._apply_opt(*self._pending_opt)
                self._pending_opt = None
        if getattr(self, "_dp", None) is not None:
            if self._dp_rows[2] != B:
                raise ValueError("global batch size changed from %d to %d (static graph)" % (self._dp_rows[2], B))
            return self._train
        if self._train.B != B:
            raise ValueError("batch size changed from %d to %d (static graph)" % (self._train.B, B))
        return self._train

    def train_step(self, u, seq, pos, neg, time_seq=None, hours=None, days=None, fetch=True):
        """One optimisation step (main.py:212-219).  Returns (auc, loss) of the batch like the reference's fetch."""
        seq = np.asarray(seq)
        eng = self._train_engine(seq.shape[0])
        z = np.zeros_like(seq) if (time_seq is None or hours is None or days is None) else None
        arrs = (seq, np.asarray(pos), np.asarray(neg), z if time_seq is None else np.asarray(time_seq),
                z if hours is None else np.asarray(hours), z if days is None else np.asarray(days))
        if getattr(self, "_dp", None) is not None:
            self._dp.step(arrs)                            # this rank's rows -> backward -> exchange -> Adam (loss / auc: global)
        else:
            eng.train_step(*arrs)
        if fetch:
            loss, auc = eng.loss_auc()
            return auc, loss
        return None

    # -- the same step with its batch handed over AHEAD of time ------------------------------------------------------
    def feed(self, u, seq, pos, neg, time_seq=None, hours=None, days=None):
        """Hands over the batch of a coming step (the sampler's output, as train_step takes it) while earlier steps run:
        one pinned copy over PCIe on a copy stream into a device ring (Engine.enable_feed).  train_fed() then runs the oldest
        waiting batch.  Feeding one batch ahead -- feed(b0); loop: feed(b[i + 1]); train_fed() -- leaves no copy between two
        steps on the device.  Same arithmetic, same batches as train_step: only the transport differs."""
        seq = np.asarray(seq)
        eng = self._train_engine(seq.shape[0])
        if getattr(self, "_dp", None) is not None:
            self._fed.append((u, seq, pos, neg, time_seq, hours, days))     # data parallel: the batch waits on the host
            return
        if getattr(eng, "_feed_ring", None) is None:
            g = self.steps_per_launch
            eng.enable_feed(n_slots=16 if g > 1 else 8, steps_per_graph=g)
        eng.feed(seq, pos, neg, time_seq, hours, days)

    @property
    def steps_per_launch(self):
        """Steps a graph launch of the fed training path runs when as many batches wait (Engine.capture(n_steps)): between two graph
        launches the device idles for the 5-9 us the next launch takes to start, 1-2 % of a 0.33 ms step.  CASTREC_STEPS_PER_GRAPH
        overrides the default of 4; 1 without HIP graphs, with row-sparse Adam, or data parallel."""
        dp_cfg = getattr(self, "_dp_cfg", None)
        data_parallel = getattr(self, "_dp", None) is not None or (dp_cfg is not None and dp_cfg[1] > 1)
        # (row-sparse Adam: the engine's own setting once it exists, before that what it will read -- Engine.__init__)
        lazy = self._train.lazy_adam if self._train is not None else bool(int(os.environ.get("CASTREC_LAZY_ADAM", "0")))
        if not self._graph or data_parallel or self._batch_global is not None or lazy:
            return 1
        return max(1, int(os.environ.get("CASTREC_STEPS_PER_GRAPH", "4")))

    @property
    def feed_ahead(self):
        """How many fed batches may wait for their steps (main.py keeps that many handed over): one more than steps_per_launch, so that
        a launch's last step finds its successor's batch in place."""
        g = self.steps_per_launch
        return g + 1 if g > 1 else 1

    def train_fed_many(self, max_steps=None):
        """Runs the oldest waiting batch's step -- or steps_per_launch steps in one graph launch when as many batches wait and
        max_steps allows it.  Returns the number of steps run; loss / auc of the last one: loss_auc()."""
        if getattr(self, "_dp", None) is not None:
            self.train_step(*self._fed.pop(0), fetch=False)
            return 1
        eng = self._train
        if eng is None or getattr(eng, "_feed_ring", None) is None:
            raise RuntimeError("train_fed_many(): feed() a batch first")
        return eng.train_fed(max_steps=max_steps)

    def loss_auc(self):
        """(auc, loss) of the last step run, like train_step's fetch."""
        loss, auc = self._train.loss_auc()
        return auc, loss

    def train_fed(self, fetch=True):
        if getattr(self, "_dp", None) is not None:
            return self.train_step(*self._fed.pop(0), fetch=fetch)
        eng = self._train
        if eng is None or getattr(eng, "_feed_ring", None) is None:
            raise RuntimeError("train_fed(): feed() a batch first")
        eng.train_fed(max_steps=1)
        if fetch:
            loss, auc = eng.loss_auc()
            return auc, loss
        return None

    # -- inference --------------------------------------------------------------------------------
    def predict(self, sess, u, seq, item_idx, timeseq=None, input_context_seq=None, hours_seq=None, days_seq=None,
                want_attention=True):
        seq = np.asarray(seq)
        if seq.ndim == 1:
            seq = seq[None]
        B = seq.shape[0]
        key = (B, bool(want_attention))
        if key not in self._eval:
            self._eval[key] = Engine(self.name, self.usernum, self.itemnum, self.hp, B, training=False, share=self._owner,
                                     want_attn=want_attention)
        eng = self._eval[key]
        z = np.zeros_like(seq)
        f = lambda a: z if a is None else np.asarray(a).reshape(B, -1)
        eng.forward_eval(seq, f(timeseq), f(hours_seq), f(days_seq))
        cand = np.asarray(item_idx, np.int32)
        if cand.ndim == 1:
            cand = np.tile(cand[None], (B, 1))
        lg = eng.test_logits(torch.from_numpy(np.ascontiguousarray(cand)).to(eng.dev))
        self.attention_weights = eng.attn_weights
        attn = eng.attn_weights.cpu().numpy() if (want_attention and eng.attn_weights is not None) else None
        return lg.cpu().numpy(), attn

    # -- checkpoints (tf.train.Saver, main.py:159,227) ----------------------------------------------
    def state_dict(self):
        d = {"model": self.name, "P": self._owner.P.detach().cpu(), "names": self._owner.layout.logical_names()}
        if self._train is not None:
            d.update(M=self._train.Mom.cpu(), V=self._train.Vel.cpu(), state=self._train.state.cpu())
        return d

    def save(self, path):
        torch.save(self.state_dict(), path)
        return path

    def _apply_opt(self, M, V, next_step):
        self._train.Mom.copy_(M); self._train.Vel.copy_(V)
        self._train.set_step(int(next_step))

    def load(self, path):
        """Parameters AND optimiser state (Adam m / v, step number), like saver.restore (main.py:165-166): a training
        step after load() continues the saved run.  The state is kept aside until the training engine exists."""
        d = torch.load(path, map_location="cpu")
        if d["model"] != self.name or d["P"].numel() != self._owner.P.numel():
            raise ValueError("checkpoint %s does not match model %s" % (path, self.name))
        self._owner.P.copy_(d["P"])
        if "M" in d:
            st = d["state"]
            cur = int(st[4:5].view(torch.int32)[0])
            nxt = cur if st.numel() == 16 else cur + 1   # 8-float state of earlier files: [4] counted COMPLETED steps
            if self._train is not None:
                self._apply_opt(d["M"], d["V"], nxt)
            else:
                self._pending_opt = (d["M"], d["V"], nxt)

    def load_tf_checkpoint(self, prefix):
        """Loads a checkpoint written by the reference (tf.train.Saver bundle `<prefix>.index` +
        `<prefix>.data-00000-of-00001`, main.py:231-233 there), Adam slots and step count included when the bundle
        holds them (they then seed the next train_step, as saver.restore does before main.py:167-175's one step)."""
        from . import tf_bundle
        params, slot_m, slot_v, steps = tf_bundle.load_logical_with_slots(prefix)
        want = {n: tuple(self._owner.layout.view(self._owner.P, n).shape) for n in self._owner.layout.logical_names()}
        missing, extra = sorted(set(want) - set(params)), sorted(set(params) - set(want))
        if missing or extra:
            raise ValueError("checkpoint %s does not match model %s: missing %s, unexpected %s" % (prefix, self.name, missing, extra))
        for n, shp in want.items():
            if tuple(params[n].shape) != shp:
                raise ValueError("checkpoint %s: %s has shape %s, model expects %s" % (prefix, n, params[n].shape, shp))
        self._owner.load_params(params)
        have_m, have_v = set(slot_m) == set(want), set(slot_v) == set(want)
        if (slot_m or slot_v) and not (have_m and have_v):
            raise ValueError("checkpoint %s holds Adam slots for only part of model %s's variables (m: %d, v: %d of %d)"
                             % (prefix, self.name, len(slot_m), len(slot_v), len(want)))
        if have_m and have_v:
            lay = self._owner.layout
            M, V = torch.zeros(lay.n_total), torch.zeros(lay.n_total)
            for n in want:
                lay.view(M, n).copy_(torch.tensor(np.asarray(slot_m[n], np.float32)))
                lay.view(V, n).copy_(torch.tensor(np.asarray(slot_v[n], np.float32)))
            if self._train is not None:
                self._apply_opt(M, V, steps + 1)
            else:
                self._pending_opt = (M, V, steps + 1)

    def get_params(self):
        return self._owner.get_params()

    def load_params(self, d):
        self._owner.load_params(d)


class SASRec(_Model):
    """models/sasrec.py:4-5: SASRec(usernum, itemnum, args, static=False, reuse=None)."""

    def __init__(self, usernum, itemnum, args, static=False, reuse=None, **kw):
        self.NAME = "sasrec_static" if static else "sasrec"
        super().__init__(usernum, itemnum, args, reuse, **kw)


def _cast(n):
    class _C(_Model):
        NAME = "cast_%d" % n
        __doc__ = "models/cast_%d.py: CAST%d(usernum, itemnum, ratingnum, args, reuse=None)." % (n, n)

        def __init__(self, usernum, itemnum, ratingnum, args, reuse=None, **kw):
            self.ratingnum = ratingnum
            super().__init__(usernum, itemnum, args, reuse, **kw)
    _C.__name__ = "CAST%d" % n
    return _C


CAST1, CAST2, CAST3, CAST4, CAST5, CAST6, CAST7, CAST8, CAST9 = (_cast(n) for n in range(1, 10))


def build_model(name, usernum, itemnum, ratingnum, args, **kw):
    """The registry of main.py:121-142."""
    name = name.lower()
    if name not in MODELS:
        raise ValueError("provide model from %s" % MODELS)
    if name == "sasrec":
        return SASRec(usernum, itemnum, args, **kw)
    if name == "sasrec_static":
        return SASRec(usernum, itemnum, args, static=True, **kw)
    return globals()["CAST%s" % name.split("_")[1]](usernum, itemnum, ratingnum, args, **kw)
