"""Static forward/backward programs of the SASRec / CAST graphs, composed of libcastrec kernels.

The reference builds a TensorFlow graph once (models/sasrec.py, models/cast_N.py) and runs it with
``sess.run``; here the graph of a model is compiled ONCE into two flat lists of pre-built C-ABI
launches (forward, backward) over persistent device buffers, which makes the whole step capturable
as a HIP graph (launch-bound problem sizes).  Gradients are hand-derived per op; there is no autograd
and no CPU fallback.

Memory layout (all fp32, resident in HBM for the life of the engine):
  P / M / V   flat parameter, Adam-m, Adam-v vectors: [table section | dense section]
  Gt          table-gradient accumulator (float atomics), zeroed by cr_adam_step after use
  Gs          dense-gradient slabs [n_slabs, n_dense]: every reducing kernel writes slab s from its
              workgroup s (no atomics, bitwise reproducible); cr_adam_step sums the slabs
  activations one [B*T, ld] buffer per tensor of the graph (row = b*T + t)
"""
import ctypes as C
import math
import os
import zlib

import numpy as np
import torch

from . import lib as L
from . import ops as O

# arithmetic of the attention products (castrec.h CR_PREC_*): "f32" = exact fp32 MFMA; "bf16x3" = bf16 MFMA on hi + lo
# split operands (three products, ~1e-5 relative: inside the 1e-3 fp32 bound); "bf16" = plain bf16 operands
ATTN_PRECISIONS = {"f32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16}
DEFAULT_ATTN_PRECISION = "bf16x3"

MODELS = ["cast_1", "cast_2", "cast_3", "cast_4", "cast_5", "cast_6", "cast_7", "cast_8", "cast_9",
          "sasrec", "sasrec_static"]          # main.py:28


def site_id(name):
    """Stable 24-bit id of a dropout call site (names follow the graph: 'emb', 'trunk.0.attn', ...)."""
    return zlib.crc32(name.encode()) & 0xFFFFFF


def positional_encoding(dim, length):
    """modules.py:27-37 (sin on even FLAT indices, cos on odd ones; exponent 2*i/dim)."""
    v = np.array([pos / np.power(10000, 2 * i / dim) for pos in range(length) for i in range(dim)])
    v[::2] = np.sin(v[::2])
    v[1::2] = np.cos(v[1::2])
    return v.reshape(length, dim).astype(np.float32)


class Hyper:
    """The fields of the reference's argparse namespace the graphs read (main.py:45-82)."""

    def __init__(self, args=None, **kw):
        d = dict(maxlen=50, hidden_units=50, num_blocks=2, num_heads=1, dropout_rate=0.5, l2_emb=0.0, lr=1e-3,
                 max_bins=200, num_context_blocks=2, seed=42)
        for k in d:
            if args is not None and hasattr(args, k):
                d[k] = getattr(args, k)
        d.update(kw)
        self.__dict__.update(d)
        if self.hidden_units % self.num_heads != 0:
            raise ValueError("hidden_units must be divisible by num_heads (tf.split, modules.py:208)")


def auto_slabs(M):
    """Number of gradient slabs = workgroups of every weight-gradient kernel.  The fused backward kernels work on two
    64-row tile groups per workgroup: ceil(M / 128) workgroups have both groups full (M = 25 600: 200 slabs, 0.8 %
    faster end to end than 256 slabs of 100 rows, and 22 % fewer slab bytes for Adam to sum); small problems get one
    64-row tile per workgroup so that enough workgroups exist; never more than one workgroup per CU."""
    n = -(-M // 128)
    if n < 128:
        n = -(-M // 64)
    return max(1, min(256, n))


# ---------------------------------------------------------------------------------------------------
# parameter layout
# ---------------------------------------------------------------------------------------------------
def _stack_entries(prefix, L_, D):
    out = []
    for i in range(L_):
        p = "%s.%d." % (prefix, i)
        out += [(p + "ln1.gamma", (D,), "ones"), (p + "ln1.beta", (D,), "zeros"),
                (p + "wqkv", (D, 3 * D), "glorot3"), (p + "bqkv", (3 * D,), "zeros"),
                (p + "ln2.gamma", (D,), "ones"), (p + "ln2.beta", (D,), "zeros"),
                (p + "w1", (D, D), "glorot"), (p + "b1", (D,), "zeros"),
                (p + "w2", (D, D), "glorot"), (p + "b2", (D,), "zeros")]
    out += [(prefix + ".lnf.gamma", (D,), "ones"), (prefix + ".lnf.beta", (D,), "zeros")]
    return out


def model_structure(model, hp):
    """Which pieces a graph has: context stacks, small tables, mlp width (SURVEY 8a variant table)."""
    L_, Lc = hp.num_blocks, hp.num_context_blocks
    s = dict(ctx={}, small=[], mlp=0, learned_pos=model in ("sasrec", "cast_9"))
    if model in ("cast_1", "cast_2", "cast_3", "cast_4", "cast_5", "cast_6"):
        s["ctx"]["ctx_time"] = L_
    if model == "cast_8":
        s["ctx"].update(ctx_hours=L_, ctx_days=L_)
    if model == "cast_9":
        s["ctx"].update(ctx_hours=Lc, ctx_days=Lc, ctx_time=Lc)
    if "ctx_time" in s["ctx"]:
        s["small"].append(("time_emb", hp.max_bins + 1))
    if model in ("cast_3", "cast_4", "cast_5", "cast_6", "cast_7", "cast_8", "cast_9"):
        s["small"] += [("hours_emb", 25), ("days_emb", 8)]
    s["mlp"] = {"cast_2": 2, "cast_3": 3, "cast_4": 4, "cast_5": 3, "cast_6": 4, "cast_7": 3, "cast_8": 3, "cast_9": 4}.get(model, 0)
    return s


class ParamLayout:
    def __init__(self, model, usernum, itemnum, hp):
        D, T = hp.hidden_units, hp.maxlen
        st = model_structure(model, hp)
        table = [("item_emb", (itemnum + 1, D), "glorot")]
        if st["learned_pos"]:
            table.append(("pos_emb", (T, D), "glorot"))
        dense = [(n, (v, D), "glorot") for n, v in st["small"]]
        for pfx, Lc in st["ctx"].items():
            dense += _stack_entries(pfx, Lc, D)
        if st["mlp"]:
            k = st["mlp"]
            dense += [("mlp.w1", (k * D, k * D), "glorot"), ("mlp.b1", (k * D,), "zeros"),
                      ("mlp.w2", (k * D, D), "glorot"), ("mlp.b2", (D,), "zeros")]
        dense += _stack_entries("trunk", hp.num_blocks, D)
        self.entries, off = {}, 0
        for (n, shape, init) in table:
            self.entries[n] = (off, shape, init, "table"); off += int(np.prod(shape))
        self.n_table = off
        for (n, shape, init) in dense:
            self.entries[n] = (off, shape, init, "dense"); off += int(np.prod(shape))
        self.n_total, self.n_dense, self.D = off, off - self.n_table, D
        # the lookup tables lead the flat vector (item_emb, pos_emb, then the small context tables): the range
        # tf.contrib.layers.l2_regularizer(l2_emb) covers (modules.py:149-153)
        self.n_l2 = self.n_table + sum(int(np.prod(self.entries[n][1])) for n, _ in st["small"])

    def logical_names(self):
        """Names as the oracle / reference variables see them (wq, wk, wv instead of the fused wqkv)."""
        out = []
        for n in self.entries:
            if n.endswith("wqkv"):
                out += [n[:-4] + x for x in ("wq", "wk", "wv")]
            elif n.endswith("bqkv"):
                out += [n[:-4] + x for x in ("bq", "bk", "bv")]
            else:
                out.append(n)
        return out

    def view(self, flat, name):
        """torch view of a (logical) parameter inside a flat [n_total] tensor."""
        D = self.D
        for j, x in enumerate(("q", "k", "v")):
            if name.endswith(".w" + x):
                off, shape, _, _ = self.entries[name[:-2] + "wqkv"]
                return flat[off:off + D * 3 * D].view(D, 3 * D)[:, j * D:(j + 1) * D]
            if name.endswith(".b" + x):
                off, shape, _, _ = self.entries[name[:-2] + "bqkv"]
                return flat[off + j * D:off + (j + 1) * D]
        off, shape, _, _ = self.entries[name]
        return flat[off:off + int(np.prod(shape))].view(*shape)

    def init_host(self, seed):
        """glorot-uniform kernels/tables (TF default of get_variable, tf.layers.dense/conv1d), zero biases,
        LayerNorm gamma=1 / beta=0 (modules.py:75-76)."""
        rs = np.random.RandomState(seed)
        flat = np.zeros(self.n_total, np.float32)
        for n, (off, shape, init, _) in self.entries.items():
            size = int(np.prod(shape))
            if init == "glorot":
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                flat[off:off + size] = rs.uniform(-lim, lim, size)
            elif init == "glorot3":       # three independent [D,D] dense kernels side by side
                D = shape[0]
                lim = math.sqrt(6.0 / (D + D))
                flat[off:off + size] = rs.uniform(-lim, lim, size)
            elif init == "ones":
                flat[off:off + size] = 1.0
        return flat


# ---------------------------------------------------------------------------------------------------
class Engine:
    def __init__(self, model, usernum, itemnum, hp, batch_size, training=True, seed=None, n_slabs=None,
                 share=None, batch_global=None, row_offset=0, want_attn=False, device="cuda", fused=None,
                 attn_precision=None, lazy_adam=None):
        if model not in MODELS:
            raise ValueError("model must be one of %s" % MODELS)
        if not torch.cuda.is_available():
            raise RuntimeError("castrec_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.model, self.hp, self.B, self.T, self.D = model, hp, batch_size, hp.maxlen, hp.hidden_units
        self.H = hp.num_heads
        if attn_precision is None:
            attn_precision = os.environ.get("CASTREC_ATTN_PRECISION") or getattr(hp, "attn_precision", None) or DEFAULT_ATTN_PRECISION
        if attn_precision not in ATTN_PRECISIONS:
            raise ValueError("attn_precision must be one of %s" % sorted(ATTN_PRECISIONS))
        self.attn_precision = attn_precision
        if attn_precision == "f32" and hp.hidden_units > 64:
            import warnings
            warnings.warn("attn_precision='f32' at hidden_units %d leaves the tuned path: the exact-fp32 row phases above 64 columns are "
                          "the unfused kernels (2.6x slower at the C4 shape, 11x at C5; profiles/r02d_other_shapes.log); the "
                          "fp32-grade default is 'bf16x3'" % hp.hidden_units, RuntimeWarning, stacklevel=2)
        # row-sparse Adam on the item table (DEVIATION from the reference's dense update, for tables like C5's 10 M rows;
        # castrec.h cr_adam_desc.lazy_ids): off unless asked for
        self.lazy_adam = bool(int(os.environ.get("CASTREC_LAZY_ADAM", "0"))) if lazy_adam is None else bool(lazy_adam)
        self.M = self.B * self.T
        self.usernum, self.itemnum = usernum, itemnum
        self.training = training
        # fused row-phase kernels (cr_block_*) need the hidden size to fit one 64-column tile
        self.single_pass_bwd = os.environ.get("CASTREC_TWO_PASS_ATTN_BWD") != "1"
        self.fuse_tails = os.environ.get("CASTREC_NO_TAILS") != "1"
        self.fuse_embed = os.environ.get("CASTREC_NO_EMBED_FUSION") != "1"
        self.fuse_stack = os.environ.get("CASTREC_NO_STACK_KERNEL") != "1"
        self.fuse_stack_bwd = os.environ.get("CASTREC_NO_STACK_BWD") != "1"
        self.fuse_wide = os.environ.get("CASTREC_NO_WIDE") != "1"
        self._pending_embed = {}
        self._scatter_recipe, self._scatter_claimed = {}, set()
        self._ln_recipe, self._ln_claimed = {}, set()
        self._lnf_intent = {}                         # y buffer of a stack's last block -> (LN output pointer, param prefix, out, ld, col)
        # one launch per block backward (cr_stack_block_bwd): the gradient of a block input then exists as TWO partials;
        # _grad2 maps an activation buffer to the second addend of its gradient until a consumer takes it (see _second_grad)
        self.fuse_block_bwd = os.environ.get("CASTREC_NO_BLOCK_BWD") != "1"
        self._grad2 = {}
        self._grad2_bufs = {}                         # data_ptr -> (activation buffer, second-addend buffer)
        self._block_bwd_ranges = []                   # slab-0 float ranges written by cr_stack_block_bwd (one slab per sequence)
        self.fuse_head_ln = os.environ.get("CASTREC_NO_HEAD_LN") != "1"
        self._check_ids = os.environ.get("CASTREC_NO_ID_CHECK") != "1"
        self.fused = (4 <= hp.hidden_units <= 64) if fused is None else bool(fused)
        if self.fused and not 4 <= hp.hidden_units <= 64:
            raise ValueError("fused block kernels need 4 <= hidden_units <= 64")
        self.dev = torch.device(device)
        self.seed = hp.seed if seed is None else seed
        self.layout = share.layout if share is not None else ParamLayout(model, usernum, itemnum, hp)
        lay = self.layout
        if not n_slabs:
            n_slabs = auto_slabs(batch_size * hp.maxlen)
            # the register-layout backward kernels (cr_stack_bwd.hip) deal (sequence, round) items to the slab workgroups:
            # with a slab per item every workgroup runs one round (B = 128 sequences of 13 tiles: 256 workgroups, 256 CUs)
            if (self.fused and self.attn_precision != "f32" and 8 <= self.D <= 64 and self.T <= 224
                    and os.environ.get("CASTREC_NO_STACK_BWD") != "1"):
                items = batch_size * (2 if self.T > 112 else 1)
                if n_slabs < items <= 512:
                    n_slabs = items
                # the prediction head as the tail of the trunk's last forward launch (cr_stack_fwd_head): its two workgroups per sequence
                # write a slab each of the final LayerNorm's gradient
                if (training and self.T > 16 and batch_size <= 160 and n_slabs < 2 * batch_size <= 512
                        and os.environ.get("CASTREC_NO_HEAD_FUSION") != "1"):
                    n_slabs = 2 * batch_size
        self.n_slabs = n_slabs
        # cr_gemm_wgrad reduces over rows in n_wslabs workgroups per output tile: at the large hidden sizes of the unfused / wide
        # path (D x D weights, 4 blocks) a slab per 128 rows is 264 MB written and read again per step (config C4); 32 suffice
        # to fill the chip (tiles x slabs workgroups).  cr_adam_step learns per 256-parameter block how many slabs are in use.
        self.n_wslabs = n_slabs if (self.fused or self.D <= 128) else min(n_slabs, int(os.environ.get("CASTREC_WSLABS", "32")))
        self._wgrad_ranges = []
        self.slab_counts = None
        f32 = dict(dtype=torch.float32, device=self.dev)
        if share is not None:
            self.P = share.P
        else:
            self.P = torch.from_numpy(lay.init_host(self.seed)).to(self.dev)
        self.state = torch.zeros(L.CR_STATE_FLOATS, **f32)
        self.state[4:5].view(torch.int32)[0] = 1                      # number of the first step (see set_step)
        if training:
            self.Mom = torch.zeros(lay.n_total, **f32)
            self.Vel = torch.zeros(lay.n_total, **f32)
            # flat gradient bucket [table | dense | loss stats(3) + pad]: the table part receives float atomics
            # directly; the dense part is written by cr_reduce_slabs (data-parallel path only)
            self.Gflat = torch.zeros(lay.n_total + 4, **f32)
            self.Gt = self.Gflat[:lay.n_table]
            self.Gs = torch.zeros(n_slabs, max(lay.n_dense, 1), **f32)
        self.drop = O.Drop(hp.dropout_rate if training else 0.0, self.seed, self.state, row_offset)
        self.batch_global = self.B if batch_global is None else batch_global
        self.want_attn = want_attn
        self.attn_weights = None
        self._bufs, self._keep = {}, []
        self.fwd, self.bwd = [], []
        self._bwd_factories = []
        self._grad_written = set()
        i32 = dict(dtype=torch.int32, device=self.dev)
        # the six id inputs of one batch live in ONE [6, M] buffer so a staged batch is a single D2D copy
        self.ID_KEYS = ("seq", "pos", "neg", "time", "hours", "days")
        # The occurrence index of the batch (castrec.h, "occurrence index"; round 5): item -> the rows that hold it as seq / pos /
        # neg id, built on the host beside the batch and travelling with the ids -- a SLOT is [6 M ids | pad | index] -- so that the
        # item (and learned positional) table's gradient is an ordered gather inside cr_adam_step instead of float atomics from the
        # head and the embedding backward.  Off (CASTREC_NO_INDEX=1, row-sparse Adam, hidden sizes the gather does not take): the
        # atomics of rounds 1-4.
        ng, ent = C.c_int(0), C.c_int(0)                        # the gather's geometry at this hidden size (0: a size it does not take)
        self.use_index = bool(training and not self.lazy_adam and os.environ.get("CASTREC_NO_INDEX") != "1"
                              and L.lib.cr_tgrad_geometry(self.D, C.byref(ng), C.byref(ent)))
        self.ids_words = 6 * self.M
        self.index_off = (self.ids_words + 3) // 4 * 4
        self.index_lay, self._ixb = None, None
        if self.use_index:
            self.index_lay = L.IndexLayout()
            t_pos = self.T if "pos_emb" in self.layout.entries else 0
            L.check(L.lib.cr_batch_index_layout(self.M, itemnum + 1, t_pos, ng.value, ent.value, C.byref(self.index_lay)), "cr_batch_index_layout")
            self._ixb = L.lib.cr_index_builder_create(self.M, itemnum + 1, t_pos, ng.value, ent.value)
            if not self._ixb:
                raise MemoryError("cr_index_builder_create(%d, %d, %d)" % (self.M, itemnum + 1, t_pos))
            # workspace of the rows a batch cuts into slices (a hot item with more occurrences than one workgroup sums)
            self._tg_part = torch.zeros(self.index_lay.cap_blocks, (self.D + 3) // 4 * 4, dtype=torch.float32, device=self.dev)
            self._tg_tickets = torch.zeros(self.index_lay.cap_blocks, dtype=torch.int32, device=self.dev)
        self.slot_words = self.index_off + (int(self.index_lay.total_words) if self.use_index else 0) if self.use_index else self.ids_words
        self.batch_buf = torch.zeros(self.slot_words, **i32)
        self.ids_all = self.batch_buf[:self.ids_words].view(6, self.M)
        self.ids = {k: self.ids_all[i] for i, k in enumerate(self.ID_KEYS)}
        self._host_slot = np.zeros(self.slot_words, np.int32)          # host mirror of the static batch buffer (set_batch builds the index in it)
        self._tg_rows = None                                            # (rows, rows2 or None, ld, scale): the seq lookup's gradient rows
        self.bitwise_reproducible = self.use_index                      # no float atomics left in a step's gradients (loss sums apart)
        self.static_pe = torch.from_numpy(positional_encoding(self.D, self.T)).to(self.dev)
        self._build()
        self._finalize()
        self.graph = None
        self._id_ring, self._ring_next = None, None

    # ---- small helpers -------------------------------------------------------------------------
    def buf(self, name, cols, rows=None):
        if name not in self._bufs:
            self._bufs[name] = torch.zeros(self.M if rows is None else rows, cols, dtype=torch.float32, device=self.dev)
        return self._bufs[name]

    def vec(self, name, n=None):
        if name not in self._bufs:
            self._bufs[name] = torch.zeros(self.M if n is None else n, dtype=torch.float32, device=self.dev)
        return self._bufs[name]

    def p(self, name):
        return self.layout.view(self.P, name)

    def _pptr(self, name):
        off = self.layout.entries[name][0]
        return self.P.data_ptr() + 4 * off

    def _gptr(self, name):
        """slab-0 pointer of a dense parameter's gradient / pointer into Gt for a table parameter."""
        off, _, _, kind = self.layout.entries[name]
        if kind == "table":
            return self.Gt.data_ptr() + 4 * off
        return self.Gs.data_ptr() + 4 * (off - self.layout.n_table)

    def _acc(self, key):
        """Backward-time accumulate flag of a gradient buffer: False for its first writer."""
        first = key not in self._grad_written
        self._grad_written.add(key)
        return 0 if first else 1

    def _call(self, lst, name, *args):
        if name in ("cr_gemm_rows", "cr_gemm_wgrad"):
            # the unfused dense layers (hidden sizes above 64, the CAST mlp) follow the engine's matmul arithmetic
            for i in range(args[1]):
                args[0][i].precision = ATTN_PRECISIONS[self.attn_precision]
        if name == "cr_gemm_wgrad":
            args = args[:3] + (self.n_wslabs,)
            base = self.Gs.data_ptr()
            for i in range(args[1]):                      # the slab-0 float ranges this launch writes
                w = args[0][i]
                o = (w.dW - base) // 4
                self._wgrad_ranges += [(o + k * w.ldw, o + k * w.ldw + w.N) for k in range(w.K)]
                if w.db:
                    self._wgrad_ranges.append(((w.db - base) // 4, (w.db - base) // 4 + w.N))
        fn = getattr(L.lib, name)
        self._keep.append(args)
        lst.append((name, fn, args))

    def rng(self, site_name, enabled=True):
        return self.drop.rng(site_id(site_name), enabled and self.training)

    # ---- graph pieces --------------------------------------------------------------------------
    def op_embed(self, ids_key, table, out, ld_out, col_off, scale, pos=None, addend=None, drop_site=None,
                 mask=False, small=False, into_stack=False):
        """modules.py:83-164 + the input composition of each graph. `addend` = (buffer, gradbuffer).
        into_stack: `out` is the dense input of a transformer stack that follows immediately; on the fused path the
        gather then runs inside the first block's LN1 + QKV kernel (cr_block_ln_qkv_fwd_gather) instead of here."""
        V, D = self.layout.entries[table][1]
        pos_ptr = None
        if pos == "static":
            pos_ptr = self.static_pe.data_ptr()
        elif pos == "learned":
            pos_ptr = self._pptr("pos_emb")
        add_buf = addend[0] if addend else None
        d = L.EmbedDesc(self.ids[ids_key].data_ptr(), self._pptr(table), self.M, self.T, D, V, 1, float(scale), pos_ptr,
                        add_buf.data_ptr() if addend else None, add_buf.shape[1] if addend else 0,
                        self.rng(drop_site) if drop_site else O.NO_DROP,
                        self.ids["seq"].data_ptr() if mask else None, out.data_ptr(), ld_out, col_off)
        if into_stack and self.fused and self.fuse_embed and ld_out == D and col_off == 0:
            self._pending_embed[out.data_ptr()] = d                  # emitted by the first block of op_stack
        else:
            self._call(self.fwd, "cr_embed_fwd", C.byref(d))
        if not self.training:
            return

        # the item table's seq lookup under the occurrence index: no table / positional gradient is scattered -- the masked,
        # dropped-out gradient rows are left in a row buffer for the gather (cr_adam_desc.tg / cr_table_grad)
        indexed = self.use_index and table == "item_emb" and ids_key == "seq"
        if indexed:
            assert not small and self._tg_rows is None, "one seq lookup of the item table per graph"

        def make_bwd_desc(in_block=False):
            f = L.EmbedDesc.from_buffer_copy(d)
            f.out = self._grad_of(out).data_ptr()
            dadd = None
            if addend:
                assert self._acc(id(addend[1])) == 0, "addend gradient must be first-written here"
                dadd = addend[1].data_ptr()
            g2 = self._second_grad(out) if (ld_out == D and col_off == 0) else None   # a gradient in two partials: the kernels add them
            if indexed:
                if not in_block:
                    # cr_embed_bwd writes the rows as `d_addend` (the addend's gradient where the graph has one -- it IS the row --
                    # else a buffer of their own)
                    if addend:
                        rows = addend[1]
                        assert rows.shape[1] % (4 if D % 4 == 0 else (2 if D % 2 == 0 else 1)) == 0
                    else:
                        rows = self.buf("item_rows", D)
                        f.ld_add = D
                        dadd = rows.data_ptr()
                    self._tg_rows = (rows, None, rows.shape[1], float(scale))
                return L.EmbedBwdDesc(f, None, None, dadd, 0, 0, g2.data_ptr() if g2 is not None else None)
            return L.EmbedBwdDesc(f, self._gptr(table), self._gptr("pos_emb") if pos == "learned" else None, dadd,
                                  self.Gs.shape[1] if small else 0, self.n_slabs if small else 0, g2.data_ptr() if g2 is not None else None)

        # the first block's QKV backward can apply this scatter itself (cr_block_ln_qkv_bwd_scatter): large-table
        # mode only (a learned positional table's gradient is then accumulated with atomics: its slot in the table
        # section is zero at that point); the block's factory (it runs before this one) claims it
        # (a small context table: only cr_stack_block_bwd takes it -- reduced in LDS, written as slabs -- and only with two slabs
        #  per sequence pair to write to)
        if (out.data_ptr() in self._pending_embed and (addend is None or addend[0].shape[1] == D)
                and (not small or (V <= 256 and D <= 64 and pos is None and self.n_slabs >= 2 * min(self.B, self.n_slabs)))):
            self._scatter_recipe[out.data_ptr()] = (make_bwd_desc, addend[0] if addend else None, small, table, indexed, float(scale))

        def factory():
            if out.data_ptr() in self._scatter_claimed:
                return []
            lst = []
            self._call(lst, "cr_embed_bwd", C.byref(make_bwd_desc()))
            return lst
        factory.pair_aware = ld_out == D and col_off == 0
        self._bwd_factories.append(factory)

    def _grad_of(self, t):
        """gradient buffer paired with activation buffer t (same shape), created on demand."""
        key = "d@%d" % t.data_ptr()
        if key not in self._bufs:
            self._bufs[key] = torch.zeros_like(t)
        return self._bufs[key]

    def _second_grad(self, t):
        """Second addend of the gradient of activation buffer t (or None), handed to a consumer that adds it itself."""
        return self._grad2.pop(t.data_ptr(), None)

    def _fold_pairs(self):
        """cr_eltwise launches that add every pending second addend into its gradient buffer: for consumers that read
        one gradient pointer."""
        lst = []
        for ptr, (t, g2) in list(self._grad2_bufs.items()):
            if ptr in self._grad2:
                g = self._grad_of(t)
                e = L.EltDesc(L.ELT_ADD, g.data_ptr(), g.shape[1], g2.data_ptr(), g2.shape[1], g.data_ptr(), g.shape[1], g.shape[0], g.shape[1],
                              O.NO_DROP, None, 0)
                self._call(lst, "cr_eltwise", C.byref(e))
                del self._grad2[ptr]
                if self._tg_rows is not None and self._tg_rows[0] is g and self._tg_rows[1] is g2:
                    # the occurrence index's row pair was this pair: g now holds the SUM (the add above is in place), so the gather
                    # reads one partial
                    self._tg_rows = (g, None) + self._tg_rows[2:]
        return lst

    def op_layernorm(self, x, y, y_ld, y_col, pname, flags=None, skip_fwd=False):
        """modules.py:53-80.  y may be a column block of a wider (concat) buffer.  skip_fwd: the forward ran as the
        tail of the producing FFN kernel; only the backward is registered here."""
        M, D = self.M, self.D
        yptr = y.data_ptr() + 4 * y_col
        d = L.LnDesc(x.data_ptr(), D, self._pptr(pname + ".gamma"), self._pptr(pname + ".beta"), yptr, y_ld, M, D, 1e-8,
                     flags[0].data_ptr() if flags else None, flags[1].data_ptr() if flags else None)
        if not skip_fwd:
            self._call(self.fwd, "cr_layernorm_fwd", C.byref(d))
        if not self.training:
            return
        self._ln_recipe[yptr] = (x, pname)            # op_head may take this backward over (cr_head_fwd_bwd_ln)

        def factory():
            if yptr in self._ln_claimed or self._lnf_fused(x):
                return []
            dy = self._grad_of(y)
            dx = self._grad_of(x)
            bd = L.LnBwdDesc(x.data_ptr(), D, self._pptr(pname + ".gamma"), dy.data_ptr() + 4 * y_col, y_ld,
                             dx.data_ptr(), D, self._acc(id(dx)), self._gptr(pname + ".gamma"), self._gptr(pname + ".beta"),
                             self.Gs.shape[1], self.n_slabs, M, D, 1e-8)
            lst = []
            self._call(lst, "cr_layernorm_bwd", C.byref(bd))
            return lst
        self._bwd_factories.append(factory)

    def _lnf_fused(self, x):
        """True when the final LayerNorm whose input is buffer x has its backward applied inside the last block's FFN
        backward (cr_stack_ffn_bwd_ln) -- decided the same way by the LayerNorm's factory and by the block's."""
        it = self._lnf_intent.get(x.data_ptr())
        return it is not None and it[0] not in self._ln_claimed

    def op_dropout_inplace(self, c, ncols, site):
        """tf.layers.dropout on the first `ncols` columns of concat buffer c (cast_2.py:90-92, cast_4.py:115-124)."""
        ld = c.shape[1]
        d = L.EltDesc(L.ELT_DROPOUT, c.data_ptr(), ld, None, 0, c.data_ptr(), ld, self.M, ncols, self.rng(site), None, 0)
        self._call(self.fwd, "cr_eltwise", C.byref(d))
        if not self.training:
            return

        def factory():
            g = self._grad_of(c)
            bd = L.EltDesc(L.ELT_DROPOUT, g.data_ptr(), ld, None, 0, g.data_ptr(), ld, self.M, ncols, self.rng(site), None, 0)
            lst = []
            self._call(lst, "cr_eltwise", C.byref(bd))
            return lst
        self._bwd_factories.append(factory)

    def op_mlp(self, c, out, k, mask_out=False):
        """modules.py:321-335: relu(dense(relu(dense(c)))), widths kD -> kD -> D."""
        M, D = self.M, self.D
        K = k * D
        h = self.buf("mlp.h", K)
        d1 = O.gemm_desc(c, K, None, K, h, K, M, K, K, relu=True)
        d1.B, d1.bias = self._pptr("mlp.w1"), self._pptr("mlp.b1")
        d2 = O.gemm_desc(h, K, None, D, out, out.shape[1], M, D, K, relu=True,
                         mask_ids=self.ids["seq"] if mask_out else None)
        d2.B, d2.bias = self._pptr("mlp.w2"), self._pptr("mlp.b2")
        for d in (d1, d2):
            arr = (L.GemmDesc * 1)(d)
            self._call(self.fwd, "cr_gemm_rows", arr, 1)
        if not self.training:
            return

        def factory():
            lst = []
            dout, dh, dc = self._grad_of(out), self._grad_of(h), self._grad_of(c)
            g2 = self.buf("mlp.g2", D)
            e = L.EltDesc(L.ELT_GRADPREP, dout.data_ptr(), out.shape[1], out.data_ptr(), out.shape[1], g2.data_ptr(), D, M, D,
                          O.NO_DROP, self.ids["seq"].data_ptr() if mask_out else None, 0)
            self._call(lst, "cr_eltwise", C.byref(e))
            b = O.gemm_desc(g2, D, None, D, dh, K, M, K, D, trans_b=True, accumulate=bool(self._acc(id(dh))))
            b.B = self._pptr("mlp.w2")
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 1)(b), 1)
            e2 = L.EltDesc(L.ELT_GRADPREP, dh.data_ptr(), K, h.data_ptr(), K, dh.data_ptr(), K, M, K, O.NO_DROP, None, 0)
            self._call(lst, "cr_eltwise", C.byref(e2))
            w = (L.WgradDesc * 2)(L.WgradDesc(h.data_ptr(), K, g2.data_ptr(), D, self._gptr("mlp.w2"), D, self._gptr("mlp.b2"), M, D, K),
                                  L.WgradDesc(c.data_ptr(), K, dh.data_ptr(), K, self._gptr("mlp.w1"), K, self._gptr("mlp.b1"), M, K, K))
            self._call(lst, "cr_gemm_wgrad", w, 2, self.Gs.shape[1], self.n_slabs)
            b1 = O.gemm_desc(dh, K, None, K, dc, K, M, K, K, trans_b=True, accumulate=bool(self._acc(id(dc))))
            b1.B = self._pptr("mlp.w1")
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 1)(b1), 1)
            return lst
        self._bwd_factories.append(factory)

    def op_block(self, x, y, pfx, attn_out=None, skip_qkv=False, tail=None, collect=None):
        """One transformer block (sasrec.py:65-83): y = mask * FFN(LN2(MHA(LN1(x), x))).
        Fused path only: `skip_qkv` = the LN1 + QKV phase was run as the tail of the previous block's FFN kernel;
        `tail` = ("next", next block's prefix, its output buffer) or ("lnf", param prefix, out, ld, col)."""
        M, D, H, B, T = self.M, self.D, self.H, self.B, self.T
        d_head = D // H
        q_in, o, f_in, hid = (self.buf(pfx + n, D) for n in ("q_in", "o", "f_in", "hid"))
        qkv = self.buf(pfx + "qkv", D, rows=3 * M)            # [3, M, D]: Q rows, K rows, V rows
        kvalid, qvalid = self.vec(pfx + "kvalid"), self.vec(pfx + "qvalid")
        ids = self.ids["seq"]
        wqkv, bqkv = self._pptr(pfx + "wqkv"), self._pptr(pfx + "bqkv")
        if self.fused:
            return self._op_block_fused(x, y, pfx, attn_out, q_in, o, f_in, hid, qkv, kvalid, qvalid, skip_qkv, tail, collect)
        # hidden sizes 128 / 192 / 256 on bf16 arithmetic: the row phases as ONE launch each (cr_wide.hip)
        prec = ATTN_PRECISIONS[self.attn_precision]
        wbd = self._block_desc(x, y, pfx)
        wide = bool(self.fuse_wide and prec != L.PREC_F32 and L.lib.cr_wide_supported(C.byref(wbd), prec))
        MD4 = 4 * M * D
        if wide and skip_qkv:
            pass                                          # ran as the tail of the previous block's FFN kernel
        elif wide:
            self._call(self.fwd, "cr_wide_ln_qkv_fwd", C.byref(wbd), prec)
        else:
            assert not skip_qkv and tail is None, "tails need the fused or the wide (D = 128) row kernels"
            # LN1 (+ data-dependent key / query masks, modules.py:222,248-249)
            ln1 = L.LnDesc(x.data_ptr(), D, self._pptr(pfx + "ln1.gamma"), self._pptr(pfx + "ln1.beta"), q_in.data_ptr(), D, M, D,
                           1e-8, kvalid.data_ptr(), qvalid.data_ptr())
            self._call(self.fwd, "cr_layernorm_fwd", C.byref(ln1))
            # Q = LN1(x) Wq + bq ; K = x Wk + bk ; V = x Wv + bv   (modules.py:203-205), one batched launch
            gs = []
            for part, src in enumerate((q_in, x, x)):
                gd = O.gemm_desc(src, D, None, 3 * D, None, D, M, D, D)
                gd.B, gd.bias, gd.C = wqkv + 4 * part * D, bqkv + 4 * part * D, qkv.data_ptr() + part * MD4
                gs.append(gd)
            self._call(self.fwd, "cr_gemm_rows", (L.GemmDesc * 3)(*gs), 3)
        # attention core (modules.py:208-269), residual = queries
        ad = O.attn_desc(qkv, None, None, D, kvalid, qvalid, q_in, D, o, D, B, T, H, d_head,
                         rng=self.rng(pfx + "attn"), batch_global=self.batch_global,
                         dead_ids=None if attn_out is not None else ids, attn_weights=attn_out)
        ad.K, ad.V = qkv.data_ptr() + MD4, qkv.data_ptr() + 2 * MD4
        ad.precision = ATTN_PRECISIONS[self.attn_precision]
        if self.training and ad.precision != L.PREC_F32:
            # the bf16-MFMA backward recomputes the probabilities from the forward's row statistics
            ad.row_stats = self.vec(pfx + "row_stats", H * B * T * 4).data_ptr()
        self._call(self.fwd, "cr_attn_fwd", C.byref(ad))
        # LN2 + FFN (modules.py:280-318), residual = LN2 output, then * mask (sasrec.py:83)
        if wide and tail is not None and tail[0] == "next":
            nbd = self._block_desc(y, tail[2], tail[1])
            td = L.BlockTailDesc(1, C.pointer(nbd), None, None, None, 0, 0)
            self._keep.append(nbd)
            self._call(self.fwd, "cr_wide_ln_ffn_fwd_tail", C.byref(wbd), C.byref(td), prec)
        elif wide and tail is not None:
            _, pname, lo, lo_ld, lo_col = tail
            td = L.BlockTailDesc(2, None, self._pptr(pname + ".gamma"), self._pptr(pname + ".beta"), lo.data_ptr(), lo_ld, lo_col)
            self._call(self.fwd, "cr_wide_ln_ffn_fwd_tail", C.byref(wbd), C.byref(td), prec)
        elif wide:
            self._call(self.fwd, "cr_wide_ln_ffn_fwd", C.byref(wbd), prec)
        else:
            ln2 = L.LnDesc(o.data_ptr(), D, self._pptr(pfx + "ln2.gamma"), self._pptr(pfx + "ln2.beta"), f_in.data_ptr(), D, M, D,
                           1e-8, None, None)
            self._call(self.fwd, "cr_layernorm_fwd", C.byref(ln2))
            f1 = O.gemm_desc(f_in, D, None, D, hid, D, M, D, D, relu=True, rng=self.rng(pfx[:-1] + ".ffn1"))
            f1.B, f1.bias = self._pptr(pfx + "w1"), self._pptr(pfx + "b1")
            self._call(self.fwd, "cr_gemm_rows", (L.GemmDesc * 1)(f1), 1)
            f2 = O.gemm_desc(hid, D, None, D, y, D, M, D, D, rng=self.rng(pfx[:-1] + ".ffn2"), residual=f_in, ldr=D, mask_ids=ids)
            f2.B, f2.bias = self._pptr(pfx + "w2"), self._pptr(pfx + "b2")
            self._call(self.fwd, "cr_gemm_rows", (L.GemmDesc * 1)(f2), 1)
        if not self.training:
            return

        def factory():
            lst = []
            dy, dx = self._grad_of(y), self._grad_of(x)
            g2, g1, df, do, dq_in = (self.buf(pfx + n, D) for n in ("g2", "g1", "df", "do", "dq_in"))
            dqkv = self.buf(pfx + "dqkv", D, rows=3 * M)          # [3, M, D]
            stats = self.vec(pfx + "stats", H * B * T * 4)
            S = self.Gs.shape[1]
            if wide:
                G = self._gptr
                bbd = L.BlockBwdDesc(L.BlockDesc.from_buffer_copy(wbd), dy.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dx.data_ptr(),
                                     self._acc(id(dx)), G(pfx + "ln1.gamma"), G(pfx + "ln1.beta"), G(pfx + "wqkv"), G(pfx + "bqkv"),
                                     G(pfx + "ln2.gamma"), G(pfx + "ln2.beta"), G(pfx + "w1"), G(pfx + "b1"), G(pfx + "w2"), G(pfx + "b2"),
                                     S, self.n_slabs)
                # dy -> g2, g1, d_o, slabs of dgamma2 dbeta2 (and dW2 db2 dW1 db1 unless left to cr_gemm_wgrad)
                # (the kernels form the weight gradients themselves at D = 128: a [D, D] slab per 128 rows is what an activation
                #  row block costs; at 192 / 256 the slab of a 64-row block is 4 x its rows, and cr_gemm_wgrad reduces more rows
                #  per slab)
                own_w = D == 128 and os.environ.get("CASTREC_WIDE_NO_WGRAD") != "1"
                if not own_w:
                    bbd.g_w1 = bbd.g_b1 = bbd.g_w2 = bbd.g_b2 = bbd.g_wqkv = bbd.g_bqkv = None
                # The one-launch attention backward (both passes side by side, tiles dealt to eight waves) needs the per-head row
                # term delta up front; the FFN backward emits it (it reads o and q_in once more for that).  C4 shape: 1.136 ->
                # 1.094 ms per step since the fused kernel deals its tiles by iteration count (before that: equal).
                dh = D // H
                fuse_attn = (T <= 256 and 8 <= dh <= 64 and dh % 16 == 0 and os.environ.get("CASTREC_WIDE_NO_DELTA") != "1"
                             and os.environ.get("CASTREC_BF_TWO_KERNELS") is None)
                if fuse_attn:
                    delta = self.vec("attn_delta_heads", H * M)
                    bbd.attn_delta = delta.data_ptr()
                self._call(lst, "cr_wide_ln_ffn_bwd", C.byref(bbd), g2.data_ptr(), g1.data_ptr(), H if fuse_attn else 0, prec)
                if not own_w:
                    w = (L.WgradDesc * 2)(L.WgradDesc(hid.data_ptr(), D, g2.data_ptr(), D, G(pfx + "w2"), D, G(pfx + "b2"), M, D, D),
                                          L.WgradDesc(f_in.data_ptr(), D, g1.data_ptr(), D, G(pfx + "w1"), D, G(pfx + "b1"), M, D, D))
                    self._call(lst, "cr_gemm_wgrad", w, 2, S, self.n_slabs)
                dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + MD4, dqkv.data_ptr() + 2 * MD4
                abd = L.AttnBwdDesc(L.AttnDesc.from_buffer_copy(ad), do.data_ptr(), D, dq, dk, dv, D, stats.data_ptr())
                if fuse_attn:
                    abd.delta = delta.data_ptr()
                self._call(lst, "cr_attn_bwd", C.byref(abd))
                if not own_w:
                    gw, gb = G(pfx + "wqkv"), G(pfx + "bqkv")
                    w2 = (L.WgradDesc * 3)(L.WgradDesc(q_in.data_ptr(), D, dq, D, gw, 3 * D, gb, M, D, D),
                                           L.WgradDesc(x.data_ptr(), D, dk, D, gw + 4 * D, 3 * D, gb + 4 * D, M, D, D),
                                           L.WgradDesc(x.data_ptr(), D, dv, D, gw + 8 * D, 3 * D, gb + 8 * D, M, D, D))
                    self._call(lst, "cr_gemm_wgrad", w2, 3, S, self.n_slabs)
                # (dQ | dK | dV, d_o) -> dx (+ dgamma1 / dbeta1 slabs)
                self._call(lst, "cr_wide_ln_qkv_bwd", C.byref(bbd), prec)
                return lst
            # (a) gradient wrt FFN2 pre-dropout output: dy * mask * keep/(1-rate)
            e = L.EltDesc(L.ELT_GRADPREP, dy.data_ptr(), D, None, 0, g2.data_ptr(), D, M, D, self.rng(pfx[:-1] + ".ffn2"), ids.data_ptr(), 0)
            self._call(lst, "cr_eltwise", C.byref(e))
            # (b) dhid = g2 W2^T ; (c) gate by the stored post-dropout ReLU output
            b = O.gemm_desc(g2, D, None, D, g1, D, M, D, D, trans_b=True)
            b.B = self._pptr(pfx + "w2")
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 1)(b), 1)
            e1 = L.EltDesc(L.ELT_GRADPREP, g1.data_ptr(), D, hid.data_ptr(), D, g1.data_ptr(), D, M, D, self.rng(pfx[:-1] + ".ffn1"), None, 0)
            self._call(lst, "cr_eltwise", C.byref(e1))
            # (d) dW2, db2, dW1, db1
            w = (L.WgradDesc * 2)(L.WgradDesc(hid.data_ptr(), D, g2.data_ptr(), D, self._gptr(pfx + "w2"), D, self._gptr(pfx + "b2"), M, D, D),
                                  L.WgradDesc(f_in.data_ptr(), D, g1.data_ptr(), D, self._gptr(pfx + "w1"), D, self._gptr(pfx + "b1"), M, D, D))
            self._call(lst, "cr_gemm_wgrad", w, 2, S, self.n_slabs)
            # (e) df_in = (g1 W1^T + dy) * mask     (residual branch of modules.py:313)
            b1 = O.gemm_desc(g1, D, None, D, df, D, M, D, D, trans_b=True, residual=dy, ldr=D, mask_ids=ids)
            b1.B = self._pptr(pfx + "w1")
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 1)(b1), 1)
            # (f) LN2 backward -> do
            l2 = L.LnBwdDesc(o.data_ptr(), D, self._pptr(pfx + "ln2.gamma"), df.data_ptr(), D, do.data_ptr(), D, 0,
                             self._gptr(pfx + "ln2.gamma"), self._gptr(pfx + "ln2.beta"), S, self.n_slabs, M, D, 1e-8)
            self._call(lst, "cr_layernorm_bwd", C.byref(l2))
            # (g) attention backward -> dQ | dK | dV
            dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + MD4, dqkv.data_ptr() + 2 * MD4
            abd = L.AttnBwdDesc(L.AttnDesc.from_buffer_copy(ad), do.data_ptr(), D, dq, dk, dv, D, stats.data_ptr())
            self._call(lst, "cr_attn_bwd", C.byref(abd))
            # (h) dWq, dbq ; dWk, dbk ; dWv, dbv  (column blocks of the fused [D,3D] weight gradient)
            gw = self._gptr(pfx + "wqkv"); gb = self._gptr(pfx + "bqkv")
            w2 = (L.WgradDesc * 3)(L.WgradDesc(q_in.data_ptr(), D, dq, D, gw, 3 * D, gb, M, D, D),
                                   L.WgradDesc(x.data_ptr(), D, dk, D, gw + 4 * D, 3 * D, gb + 4 * D, M, D, D),
                                   L.WgradDesc(x.data_ptr(), D, dv, D, gw + 8 * D, 3 * D, gb + 8 * D, M, D, D))
            self._call(lst, "cr_gemm_wgrad", w2, 3, S, self.n_slabs)
            # (i) dq_in = dQ Wq^T + do (residual, modules.py:269) ; dx (+)= dK Wk^T, then dx += dV Wv^T
            bq = O.gemm_desc(None, D, None, 3 * D, dq_in, D, M, D, D, trans_b=True, residual=do, ldr=D)
            bq.A, bq.B = dq, wqkv
            bk = O.gemm_desc(None, D, None, 3 * D, dx, D, M, D, D, trans_b=True, accumulate=bool(self._acc(id(dx))))
            bk.A, bk.B = dk, wqkv + 4 * D
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 2)(bq, bk), 2)
            bv = O.gemm_desc(None, D, None, 3 * D, dx, D, M, D, D, trans_b=True, accumulate=True)
            bv.A, bv.B = dv, wqkv + 8 * D
            self._call(lst, "cr_gemm_rows", (L.GemmDesc * 1)(bv), 1)
            # (j) LN1 backward accumulates into dx
            l1 = L.LnBwdDesc(x.data_ptr(), D, self._pptr(pfx + "ln1.gamma"), dq_in.data_ptr(), D, dx.data_ptr(), D, 1,
                             self._gptr(pfx + "ln1.gamma"), self._gptr(pfx + "ln1.beta"), S, self.n_slabs, M, D, 1e-8)
            self._call(lst, "cr_layernorm_bwd", C.byref(l1))
            return lst
        self._bwd_factories.append(factory)

    def _block_desc(self, x, y, pfx):
        """cr_block_desc of the block with parameter prefix `pfx` (buffers are created on first use, by name)."""
        M, D = self.M, self.D
        q_in, o, f_in, hid = (self.buf(pfx + n, D) for n in ("q_in", "o", "f_in", "hid"))
        qkv = self.buf(pfx + "qkv", D, rows=3 * M)
        kvalid, qvalid = self.vec(pfx + "kvalid"), self.vec(pfx + "qvalid")
        P = self._pptr
        return L.BlockDesc(M, D, P(pfx + "ln1.gamma"), P(pfx + "ln1.beta"), P(pfx + "wqkv"), P(pfx + "bqkv"),
                           P(pfx + "ln2.gamma"), P(pfx + "ln2.beta"), P(pfx + "w1"), P(pfx + "b1"), P(pfx + "w2"), P(pfx + "b2"),
                           x.data_ptr(), q_in.data_ptr(), qkv.data_ptr(), kvalid.data_ptr(), qvalid.data_ptr(), o.data_ptr(),
                           f_in.data_ptr(), hid.data_ptr(), y.data_ptr(), self.ids["seq"].data_ptr(),
                           self.rng(pfx[:-1] + ".ffn1"), self.rng(pfx[:-1] + ".ffn2"))

    def _op_block_fused(self, x, y, pfx, attn_out, q_in, o, f_in, hid, qkv, kvalid, qvalid, skip_qkv=False, tail=None,
                        collect=None):
        """Same block through the fused row-phase kernels (cr_block_*): 3 launches forward, 3 backward; with tails
        (the next block's LN1 + QKV, or the stack's final LayerNorm, inside the FFN kernel) 2 forward.
        collect: a list -- the forward is not emitted here; (block desc, attention desc) is appended for op_stack's
        one cr_stack_fwd launch.  The backward is the same in both cases."""
        M, D, H, B, T = self.M, self.D, self.H, self.B, self.T
        ids = self.ids["seq"]
        bd = self._block_desc(x, y, pfx)
        if not skip_qkv and collect is None:
            e = self._pending_embed.pop(x.data_ptr(), None)
            if e is not None:
                self._keep.append(e)
                self._call(self.fwd, "cr_block_ln_qkv_fwd_gather", C.byref(bd), C.byref(e))
            else:
                self._call(self.fwd, "cr_block_ln_qkv_fwd", C.byref(bd))
        MD4 = 4 * M * D
        ad = O.attn_desc(qkv, None, None, D, kvalid, qvalid, q_in, D, o, D, B, T, H, D // H,
                         rng=self.rng(pfx + "attn"), batch_global=self.batch_global,
                         dead_ids=None if attn_out is not None else ids, attn_weights=attn_out)
        ad.K, ad.V = qkv.data_ptr() + MD4, qkv.data_ptr() + 2 * MD4
        # single-pass attention backward (H = 1): the forward saves its row statistics, the FFN backward emits
        # delta, the attention backward returns dQ as two partials that the QKV backward adds
        ad.precision = ATTN_PRECISIONS[self.attn_precision]
        bf = ad.precision != L.PREC_F32 and 8 <= D // H <= 64 and T <= 1024     # shapes the bf16-MFMA kernels take
        one_pass = self.training and H == 1 and self.single_pass_bwd and not bf
        if one_pass or (self.training and bf):
            row_stats = self.vec(pfx + "row_stats", H * B * T * 4)
            ad.row_stats = row_stats.data_ptr()
        if collect is not None:
            collect.append((bd, ad))
        else:
            self._call(self.fwd, "cr_attn_fwd", C.byref(ad))
        if collect is not None:
            pass
        elif tail is None:
            self._call(self.fwd, "cr_block_ln_ffn_fwd", C.byref(bd))
        elif tail[0] == "next":
            nbd = self._block_desc(y, tail[2], tail[1])
            td = L.BlockTailDesc(1, C.pointer(nbd), None, None, None, 0, 0)
            self._keep.append(nbd)
            self._call(self.fwd, "cr_block_ln_ffn_fwd_tail", C.byref(bd), C.byref(td))
        else:
            _, pname, out, out_ld, out_col = tail
            td = L.BlockTailDesc(2, None, self._pptr(pname + ".gamma"), self._pptr(pname + ".beta"), out.data_ptr(), out_ld, out_col)
            self._call(self.fwd, "cr_block_ln_ffn_fwd_tail", C.byref(bd), C.byref(td))
        if not self.training:
            return

        def factory():
            lst = []
            dy, dx = self._grad_of(y), self._grad_of(x)
            do = self.buf(pfx + "do", D)
            dqkv = self.buf(pfx + "dqkv", D, rows=3 * M)          # [3, M, D]
            stats = self.vec(pfx + "stats", H * B * T * 4)
            G = self._gptr
            bbd = L.BlockBwdDesc(L.BlockDesc.from_buffer_copy(bd), dy.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dx.data_ptr(),
                                 self._acc(id(dx)), G(pfx + "ln1.gamma"), G(pfx + "ln1.beta"), G(pfx + "wqkv"), G(pfx + "bqkv"),
                                 G(pfx + "ln2.gamma"), G(pfx + "ln2.beta"), G(pfx + "w1"), G(pfx + "b1"), G(pfx + "w2"), G(pfx + "b2"),
                                 self.Gs.shape[1], self.n_slabs)
            abd = L.AttnBwdDesc(L.AttnDesc.from_buffer_copy(ad), do.data_ptr(), D, dqkv.data_ptr(), dqkv.data_ptr() + MD4,
                                dqkv.data_ptr() + 2 * MD4, D, stats.data_ptr())
            if one_pass:
                delta = self.vec("attn_delta", M)                   # shared: consumed within the same block
                dq_part = self.buf("attn_dq_part", D)
                bbd.attn_delta, bbd.dq_part = delta.data_ptr(), dq_part.data_ptr()
                abd.delta, abd.dQ_part = delta.data_ptr(), dq_part.data_ptr()
            elif bf and H == 1:
                # bf16-MFMA backward: the FFN backward emits delta (one head: its row sum IS the head's), which lets
                # both passes run side by side in one launch; dQ comes back whole
                delta = self.vec("attn_delta", M)
                bbd.attn_delta = delta.data_ptr()
                abd.delta = delta.data_ptr()
            # bf16 arithmetic: the row phases run on the register-layout kernels (cr_stack_bwd.hip) where they take the shape
            prec = ATTN_PRECISIONS[self.attn_precision]
            rows_bf = bool(self.fuse_stack_bwd and bf and not one_pass and L.lib.cr_stack_bwd_supported(C.byref(bbd), B, T, prec))
            # ... and where the whole block fits one launch (cr_stack_bwd1.hip: one head, D < 64): ffn_bwd -> attn_bwd -> qkv_bwd per
            # sequence on a pair of workgroups; the input gradient leaves as two partials (dx, dx2)
            if (rows_bf and self.fuse_block_bwd and (H == 1 or (H == 2 and D == 64))
                    and L.lib.cr_stack_block_bwd_supported(C.byref(bbd), C.byref(abd.f), B, T, prec)):
                dx2 = self.buf(pfx + "dx2", D)
                dy2 = self._second_grad(y)
                ext = L.BlockBwd1Ext(dy2.data_ptr() if dy2 is not None else None, dx2.data_ptr(), None, None)
                nd = None
                if self._lnf_fused(y):
                    _, pname, lo, lo_ld, lo_col = self._lnf_intent[y.data_ptr()]
                    dlo = self._grad_of(lo)
                    nd = L.LnBwdDesc(y.data_ptr(), D, self._pptr(pname + ".gamma"), dlo.data_ptr() + 4 * lo_col, lo_ld, None, D, 0,
                                     self._gptr(pname + ".gamma"), self._gptr(pname + ".beta"), self.Gs.shape[1], self.n_slabs, M, D, 1e-8)
                    assert id(dy) not in self._grad_written and dy2 is None, "the final LayerNorm must be the only consumer of the stack's output"
                    dlo2 = self._second_grad(lo)
                    if dlo2 is not None:
                        assert dlo2.shape == dlo.shape
                        ext.lnf_dy2 = dlo2.data_ptr() + 4 * lo_col
                    self._keep.append(nd)
                    self._block_slab_range(pname + ".gamma", pname + ".beta")
                sc = None
                rec = self._scatter_recipe.get(x.data_ptr())
                if rec is not None and not bbd.dx_accumulate:
                    self._scatter_claimed.add(x.data_ptr())
                    sc = rec[0](True) if rec[4] else rec[0]()
                    if rec[2]:                                             # small table: two slabs per sequence pair
                        self._block_slab_range(rec[3], rec[3], 2 * min(self.B, self.n_slabs))
                    if sc.d_addend:
                        dadd2 = self.buf(pfx + "dadd2", D)
                        ext.d_addend2 = dadd2.data_ptr()
                        self._grad2[rec[1].data_ptr()] = dadd2            # the addend's gradient is d_addend + d_addend2
                        self._grad2_bufs[rec[1].data_ptr()] = (rec[1], dadd2)
                    if rec[4]:
                        # occurrence index: the kernel leaves the two masked partials of the input gradient where they wait (the
                        # addend's gradient pair, else dx / dx2) and scatters nothing; cr_adam_step gathers scale * (rows + rows2)
                        if sc.d_addend:
                            self._tg_rows = (self._grad_of(rec[1]), dadd2, D, rec[5])
                        else:
                            self._tg_rows = (dx, dx2, D, rec[5])
                    self._keep.append(sc)
                else:
                    self._grad2[x.data_ptr()] = dx2                       # the block input's gradient is dx + dx2
                    self._grad2_bufs[x.data_ptr()] = (x, dx2)
                self._keep.append(ext)
                self._block_slab_range(pfx + "ln1.gamma", pfx + "b2")
                self._call(lst, "cr_stack_block_bwd", C.byref(bbd), C.byref(abd.f), C.byref(ext), C.byref(nd) if nd is not None else None,
                           C.byref(sc) if sc is not None else None, B, T, prec)
                return lst
            lst += self._fold_pairs()                              # the kernels below read one gradient pointer
            two_heads = (rows_bf and H == 2 and D == 64 and T <= 256 and os.environ.get("CASTREC_BF_TWO_KERNELS") is None
                         and os.environ.get("CASTREC_NO_HEAD_DELTA") != "1")
            if two_heads:
                # config C3's shape: the register-layout FFN backward emits delta per head ([2, M]), so the attention backward is
                # ONE launch here too
                delta = self.vec("attn_delta_heads", H * M)
                bbd.attn_delta = delta.data_ptr()
                abd.delta = delta.data_ptr()
                nd = None
                if self._lnf_fused(y):
                    _, pname, lo, lo_ld, lo_col = self._lnf_intent[y.data_ptr()]
                    dlo = self._grad_of(lo)
                    nd = L.LnBwdDesc(y.data_ptr(), D, self._pptr(pname + ".gamma"), dlo.data_ptr() + 4 * lo_col, lo_ld, None, D, 0,
                                     self._gptr(pname + ".gamma"), self._gptr(pname + ".beta"), self.Gs.shape[1], self.n_slabs, M, D, 1e-8)
                    assert id(dy) not in self._grad_written, "the final LayerNorm must be the only consumer of the stack's output"
                    self._keep.append(nd)
                self._call(lst, "cr_stack_ffn_bwd_heads", C.byref(bbd), C.byref(nd) if nd is not None else None, B, T, H, prec)
            elif rows_bf and self._lnf_fused(y):
                _, pname, lo, lo_ld, lo_col = self._lnf_intent[y.data_ptr()]
                dlo = self._grad_of(lo)
                nd = L.LnBwdDesc(y.data_ptr(), D, self._pptr(pname + ".gamma"), dlo.data_ptr() + 4 * lo_col, lo_ld, None, D, 0,
                                 self._gptr(pname + ".gamma"), self._gptr(pname + ".beta"), self.Gs.shape[1], self.n_slabs, M, D, 1e-8)
                assert id(dy) not in self._grad_written, "the final LayerNorm must be the only consumer of the stack's output"
                self._call(lst, "cr_stack_ffn_bwd_ln", C.byref(bbd), C.byref(nd), B, T, prec)
            elif rows_bf:
                self._call(lst, "cr_stack_ffn_bwd", C.byref(bbd), B, T, prec)
            else:
                assert not self._lnf_fused(y), "final LayerNorm backward was left to a kernel that does not take it"
                self._call(lst, "cr_block_ln_ffn_bwd", C.byref(bbd))
            self._call(lst, "cr_attn_bwd", C.byref(abd))
            recipe = self._scatter_recipe.get(x.data_ptr())
            if recipe is not None and recipe[4]:
                recipe = None                                              # (indexed: these kernels only know the atomic scatter; cr_embed_bwd leaves the rows)
            if recipe is not None and not recipe[2] and not bbd.dx_accumulate:
                self._scatter_claimed.add(x.data_ptr())
                if rows_bf:
                    self._call(lst, "cr_stack_qkv_bwd_scatter", C.byref(bbd), C.byref(recipe[0]()), B, T, prec)
                else:
                    self._call(lst, "cr_block_ln_qkv_bwd_scatter", C.byref(bbd), C.byref(recipe[0]()))
            elif rows_bf:
                self._call(lst, "cr_stack_qkv_bwd", C.byref(bbd), B, T, prec)
            else:
                self._call(lst, "cr_block_ln_qkv_bwd", C.byref(bbd))
            return lst
        factory.pair_aware = True                      # (takes dy2 where it can; any other pending pair is not its input)
        self._bwd_factories.append(factory)

    def _block_slab_range(self, first, last, count=None):
        """Parameters first..last (contiguous in the layout) get their gradient from cr_stack_block_bwd: one slab per sequence
        pair, min(B, n_slabs) slabs in use (cr_adam_desc.slab_counts); `count`: another number of slabs."""
        a = self.layout.entries[first][0] - self.layout.n_table
        off, shape, _, _ = self.layout.entries[last]
        self._block_bwd_ranges.append((a, off + int(np.prod(shape)) - self.layout.n_table, min(self.B, self.n_slabs) if count is None else count))

    def _stack_kernel_fits(self, nblocks, want_attn):
        """Shapes cr_stack_fwd takes (castrec.h): one head (or two of 32 columns), D 8..64, bf16 arithmetic, K / V images +
        weights within the LDS."""
        heads_ok = self.H == 1 or (self.H == 2 and self.D == 64)          # two heads: head dim 32 = one k-step of the score product
        if not (self.fused and self.fuse_stack and nblocks >= 1 and not want_attn and heads_ok and 8 <= self.D <= 64):
            return False
        if self.attn_precision == "f32" or self.T > 256:
            return False
        T16, split = (self.T + 15) // 16 * 16, self.attn_precision == "bf16x3"
        lds = T16 * 64 * 2 * (4 if split else 2) + 3 * 4096 * 2 * (2 if split else 1) + 11 * 64 * 4 + T16 * 4
        if T16 > 208 and (split or self.H != 1):                 # more than 13 tiles: plain bf16, one head (castrec.h)
            return False
        return lds <= 160 * 1024

    def op_stack(self, x, prefix, nblocks, out, out_ld, out_col, want_attn=False):
        """block loop + final LayerNorm (sasrec.py:65-85); returns nothing, writes `out` columns.
        Fused path: block i+1's LN1 + QKV phase and the final LayerNorm run as tails of the FFN kernels; where the
        whole-stack kernel fits (cr_stack_fwd) the forward of up to four blocks + the final LayerNorm is ONE launch."""
        # (the wide row kernels take the same tails at D = 128: cr_wide_ln_ffn_fwd_tail)
        wide_tails = (not self.fused and self.fuse_wide and self.attn_precision != "f32" and self.D == 128
                      and os.environ.get("CASTREC_WIDE_NO_TAILS") != "1")
        tails = (self.fused or wide_tails) and self.fuse_tails and nblocks > 0
        stack = self._stack_kernel_fits(nblocks, want_attn)
        emb = None
        if stack and x.data_ptr() in self._pending_embed:                # the stack kernel composes its input itself
            emb = self._pending_embed.pop(x.data_ptr())
        if nblocks == 0 and x.data_ptr() in self._pending_embed:         # no block kernel to carry the gather
            self._call(self.fwd, "cr_embed_fwd", C.byref(self._pending_embed.pop(x.data_ptr())))
        ys = [self.buf("%s.%d.y" % (prefix, i), self.D) for i in range(nblocks)]
        cur = x
        collect = [] if stack else None
        for i in range(nblocks):
            nxt = ys[i]
            aw = None
            if want_attn and i == nblocks - 1:
                self.attn_weights = torch.zeros(self.H * self.B, self.T, self.T, dtype=torch.float32, device=self.dev)
                aw = self.attn_weights
            tail = None
            if tails:
                tail = ("next", "%s.%d." % (prefix, i + 1), ys[i + 1]) if i + 1 < nblocks else ("lnf", prefix + ".lnf", out, out_ld, out_col)
            self.op_block(cur, nxt, "%s.%d." % (prefix, i), attn_out=aw, skip_qkv=tails and i > 0, tail=tail, collect=collect)
            cur = nxt
        if stack:
            for i0 in range(0, nblocks, 4):
                part = collect[i0:i0 + 4]
                bds = (L.BlockDesc * len(part))(*[c[0] for c in part])
                ads = (L.AttnDesc * len(part))(*[c[1] for c in part])
                fin = i0 + 4 >= nblocks
                sd = L.StackDesc(len(part), C.cast(bds, C.POINTER(L.BlockDesc)), C.cast(ads, C.POINTER(L.AttnDesc)),
                                 self._pptr(prefix + ".lnf.gamma") if fin else None, self._pptr(prefix + ".lnf.beta") if fin else None,
                                 out.data_ptr() if fin else None, out_ld, out_col,
                                 C.pointer(emb) if (emb is not None and i0 == 0) else None)
                self._keep.append(emb)
                if not L.lib.cr_stack_fwd_supported(C.byref(sd)):
                    raise RuntimeError("cr_stack_fwd does not take the stack %s it was sized for" % prefix)
                self._keep.append((bds, ads))
                self._call(self.fwd, "cr_stack_fwd", C.byref(sd))
        if (stack and self.training and self.fuse_stack_bwd and self.D <= 64 and self.T <= 224
                and os.environ.get("CASTREC_NO_LNF_FUSION") != "1"):
            # the register-layout FFN backward of the last block can apply this LayerNorm's backward on its way in
            self._lnf_intent[cur.data_ptr()] = (out.data_ptr() + 4 * out_col, prefix + ".lnf", out, out_ld, out_col)
        self.op_layernorm(cur, out, out_ld, out_col, prefix + ".lnf", skip_fwd=tails or stack)

    def op_head(self, seq_emb):
        """sasrec.py:87-115 (+ unnormalised gradients) / test_logits sasrec.py:93-97."""
        M, D = self.M, self.D
        self.seq_emb = seq_emb
        if not self.training:
            return
        rec = self._ln_recipe.get(seq_emb.data_ptr()) if self.fuse_head_ln else None
        if rec is not None:
            # seq_emb is the output of a LayerNorm (the stack's final one): its backward runs inside the head kernel,
            # on the gradient rows while they are in registers; d(seq_emb) is never stored
            x, pname = rec
            dx = self._grad_of(x)
            assert self._acc(id(dx)) == 0
            self._ln_claimed.add(seq_emb.data_ptr())
            d = L.HeadDesc(seq_emb.data_ptr(), seq_emb.shape[1], self._pptr("item_emb"), self.ids["pos"].data_ptr(),
                           self.ids["neg"].data_ptr(), M, D, self.itemnum + 1, self.state.data_ptr(), None, 0,
                           None if self.use_index else self._gptr("item_emb"), None, None, self._head_coef())
            n = L.LnBwdDesc(x.data_ptr(), D, self._pptr(pname + ".gamma"), None, 0, dx.data_ptr(), D, 0,
                            self._gptr(pname + ".gamma"), self._gptr(pname + ".beta"), self.Gs.shape[1], self.n_slabs, M, D, 1e-8)
            # the head as the tail of the stack's last forward launch (cr_stack_fwd_head): no launch of its own.  Needs the occurrence
            # index (nothing is scattered from there) and the launch form with two workgroups per sequence.
            last = self.fwd[-1] if self.fwd else None
            if (self.use_index and last is not None and last[0] == "cr_stack_fwd" and os.environ.get("CASTREC_NO_HEAD_FUSION") != "1"
                    and last[2][0]._obj.out == seq_emb.data_ptr()
                    and L.lib.cr_stack_fwd_head_supported(last[2][0], C.byref(d), C.byref(n))):
                self._keep += [d, n]
                self.fwd[-1] = ("cr_stack_fwd_head", L.lib.cr_stack_fwd_head, (last[2][0], C.byref(d), C.byref(n)))
                self._block_slab_range(pname + ".gamma", pname + ".beta", 2 * self.B)      # workgroup (sequence, half) writes slab half * B + sequence
                return
            self._call(self.fwd, "cr_head_fwd_bwd_ln", C.byref(d), C.byref(n))
            return
        ds = self._grad_of(seq_emb)
        self._grad_written.add(id(ds))
        d = L.HeadDesc(seq_emb.data_ptr(), seq_emb.shape[1], self._pptr("item_emb"), self.ids["pos"].data_ptr(),
                       self.ids["neg"].data_ptr(), M, D, self.itemnum + 1, self.state.data_ptr(), ds.data_ptr(), ds.shape[1],
                       None if self.use_index else self._gptr("item_emb"), None, None, self._head_coef())
        self._call(self.fwd, "cr_head_fwd_bwd", C.byref(d))

    def _head_coef(self):
        """[2, M] d loss_sum / d logit of the pos / neg item per row: what the occurrence index's gather multiplies seq_emb rows with."""
        if not self.use_index:
            return None
        return self.vec("head.coef", 2 * self.M).data_ptr()

    # ---- the eleven graphs -----------------------------------------------------------------------
    def _build(self):
        m, D, T, L_ = self.model, self.D, self.T, self.hp.num_blocks
        sq = math.sqrt(D)    # modules.py:159-160 (num_units ** 0.5)
        st = model_structure(m, self.hp)
        wa_ctx = self.want_attn and m in ("cast_1", "cast_2", "cast_3", "cast_4", "cast_5", "cast_6")
        wa_trunk = self.want_attn and not wa_ctx

        def ctx_stack(table, ids_key, prefix, out, out_ld, out_col, want=False):
            e = self.buf(prefix + ".emb", D)
            self.op_embed(ids_key, table, e, D, 0, sq, small=True, into_stack=True)        # cast_1.py:30-38
            self.op_stack(e, prefix, st["ctx"][prefix], out, out_ld, out_col, want_attn=want)   # cast_1.py:42-60

        if m in ("sasrec", "sasrec_static"):
            x = self.buf("x0", D)
            self.op_embed("seq", "item_emb", x, D, 0, sq, pos="learned" if m == "sasrec" else "static",
                          drop_site="emb", mask=True, into_stack=True)                      # sasrec.py:27-62
            s = self.buf("seq_emb", D)
            self.op_stack(x, "trunk", L_, s, D, 0, want_attn=wa_trunk)
        elif m == "cast_1":
            tseq = self.buf("tseq", D)
            ctx_stack("time_emb", "time", "ctx_time", tseq, D, 0, want=wa_ctx)
            x = self.buf("x0", D)
            self.op_embed("seq", "item_emb", x, D, 0, sq, pos="static", addend=(tseq, self._grad_of(tseq)),
                          drop_site="emb", mask=True, into_stack=True)                      # cast_1.py:86-91
            s = self.buf("seq_emb", D)
            self.op_stack(x, "trunk", L_, s, D, 0)
        elif m in ("cast_2", "cast_3", "cast_4", "cast_7", "cast_8", "cast_9"):
            k = st["mlp"]
            c = self.buf("concat", k * D)
            if m == "cast_2":
                ctx_stack("time_emb", "time", "ctx_time", c, k * D, D, want=wa_ctx)
                self.op_embed("seq", "item_emb", c, k * D, 0, sq, pos="static", mask=True)   # cast_2.py:85-86
                self.op_dropout_inplace(c, 2 * D, "concat1")                                # cast_2.py:89-92
            elif m == "cast_3":
                tseq = self.buf("tseq", D)
                ctx_stack("time_emb", "time", "ctx_time", tseq, D, 0, want=wa_ctx)
                self.op_embed("hours", "hours_emb", c, k * D, D, sq, small=True)            # cast_3.py:32-41
                self.op_embed("days", "days_emb", c, k * D, 2 * D, sq, small=True)          # cast_3.py:43-52
                self.op_embed("seq", "item_emb", c, k * D, 0, sq, pos="static", addend=(tseq, self._grad_of(tseq)),
                              mask=True)                                                    # cast_3.py:112-114
                self.op_dropout_inplace(c, 3 * D, "concat1")                                # cast_3.py:117-120
            elif m == "cast_4":
                ctx_stack("time_emb", "time", "ctx_time", c, k * D, D, want=wa_ctx)
                self.op_embed("seq", "item_emb", c, k * D, 0, sq, pos="static", mask=True)   # cast_4.py:111-112
                self.op_dropout_inplace(c, 2 * D, "concat1")                                # cast_4.py:115-118
                self.op_embed("hours", "hours_emb", c, k * D, 2 * D, sq, small=True)
                self.op_embed("days", "days_emb", c, k * D, 3 * D, sq, small=True)
                self.op_dropout_inplace(c, 4 * D, "concat2")                                # cast_4.py:121-124
            elif m in ("cast_7", "cast_8"):
                if m == "cast_7":
                    self.op_embed("hours", "hours_emb", c, k * D, D, sq, small=True)        # cast_7.py:30-51
                    self.op_embed("days", "days_emb", c, k * D, 2 * D, sq, small=True)
                else:
                    ctx_stack("hours_emb", "hours", "ctx_hours", c, k * D, D)              # cast_8.py:56-74
                    ctx_stack("days_emb", "days", "ctx_days", c, k * D, 2 * D)             # cast_8.py:78-95
                self.op_embed("seq", "item_emb", c, k * D, 0, sq, pos="static", mask=True)   # cast_7.py:77-78
                self.op_dropout_inplace(c, 3 * D, "concat1")                                # cast_7.py:81-84
            else:  # cast_9
                ctx_stack("hours_emb", "hours", "ctx_hours", c, k * D, 2 * D)              # cast_9.py:56-74
                ctx_stack("days_emb", "days", "ctx_days", c, k * D, 3 * D)                 # cast_9.py:78-95
                ctx_stack("time_emb", "time", "ctx_time", c, k * D, D)                     # cast_9.py:101-129
                self.op_embed("seq", "item_emb", c, k * D, 0, sq, pos="learned")            # cast_9.py:149-160
                self.op_dropout_inplace(c, 4 * D, "concat1")                                # cast_9.py:163-168
            x = self.buf("x0", D)
            self.op_mlp(c, x, k, mask_out=(m == "cast_9"))                                  # cast_2.py:95 / cast_9.py:171-174
            s = self.buf("seq_emb", D)
            self.op_stack(x, "trunk", L_, s, D, 0, want_attn=wa_trunk)
        elif m in ("cast_5", "cast_6"):
            k = st["mlp"]
            c = self.buf("concat", k * D)
            x = self.buf("x0", D)
            if m == "cast_5":
                tseq = self.buf("tseq", D)
                ctx_stack("time_emb", "time", "ctx_time", tseq, D, 0, want=wa_ctx)
                self.op_embed("seq", "item_emb", x, D, 0, sq, pos="static", addend=(tseq, self._grad_of(tseq)),
                              into_stack=True)                                              # cast_5.py:113-114
                self.op_stack(x, "trunk", L_, c, k * D, 0)                                  # cast_5.py:118-139
                self.op_embed("hours", "hours_emb", c, k * D, D, sq, small=True)
                self.op_embed("days", "days_emb", c, k * D, 2 * D, sq, small=True)
                self.op_dropout_inplace(c, 3 * D, "concat1")                                # cast_5.py:143-146
            else:
                self.op_embed("seq", "item_emb", x, D, 0, sq, pos="static", into_stack=True)   # cast_6.py:113
                self.op_stack(x, "trunk", L_, c, k * D, 0)                                  # cast_6.py:117-138
                ctx_stack("time_emb", "time", "ctx_time", c, k * D, D, want=wa_ctx)
                self.op_dropout_inplace(c, 2 * D, "concat1")                                # cast_6.py:142-145
                self.op_embed("hours", "hours_emb", c, k * D, 2 * D, sq, small=True)
                self.op_embed("days", "days_emb", c, k * D, 3 * D, sq, small=True)
                self.op_dropout_inplace(c, 4 * D, "concat2")                                # cast_6.py:148-151
            s = self.buf("seq_emb", D)
            self.op_mlp(c, s, k)                                                            # cast_5.py:149
        self.op_head(s)

    def _finalize(self):
        if self.training:
            for fac in reversed(self._bwd_factories):
                lst = fac()
                if lst and self._grad2 and not getattr(fac, "pair_aware", False):
                    self.bwd += self._fold_pairs()            # a consumer with one gradient pointer: fold the pending pairs first
                self.bwd += lst
            assert not self._grad2, "a gradient partial was left without a consumer"
            # data parallelism overlaps the item / positional table's exchange with what is left of the backward once the last
            # launch that adds to the table gradient has been issued: the embedding backward of a looked-up table (its own launch,
            # or inside a block's backward) -- the head's rows are in from the start
            self.bwd_table_done = 0
            for i, (name, _, a) in enumerate(self.bwd):
                sc = a[0] if name == "cr_embed_bwd" else (a[4] if name == "cr_stack_block_bwd" else (a[1] if name.endswith("_scatter") else None))
                if sc is not None and sc._obj.n_slabs == 0:              # (n_slabs > 0: a small context table, written as slabs)
                    self.bwd_table_done = i + 1
            lay = self.layout
            # the step ends inside Adam (castrec.h, state block): sums and step number are read from the snapshot the
            # head kernel took, and the kernel zeroes the sums and advances the counter -- no cr_step_begin launch
            snap, tsnap = self.state.data_ptr() + 4 * 8, self.state.data_ptr() + 4 * 11
            l2 = float(self.hp.l2_emb)
            n_l2 = lay.n_l2 if l2 != 0.0 else 0
            lazy = (None, 0, 0, 0, None)
            if self.lazy_adam:
                self.lazy_flags = torch.zeros(self.itemnum + 1, dtype=torch.int32, device=self.dev)
                world = max(1, self.batch_global // self.B)
                # this step's seq | pos | neg ids; under data parallelism the ids of ALL ranks (the sparse exchange hands
                # them over, dist.DataParallel.exchange): every replica must update the same rows
                self.lazy_ids = self.ids_all[:3].reshape(-1) if world == 1 else torch.zeros(world * 3 * self.M, dtype=torch.int32, device=self.dev)
                lazy = (self.lazy_ids.data_ptr(), self.lazy_ids.numel(), self.itemnum + 1, self.D, self.lazy_flags.data_ptr())
            # slabs in use per 256-parameter block: n_wslabs where cr_gemm_wgrad is the only writer, n_slabs elsewhere
            counts = None
            per = np.full(lay.n_dense, self.n_slabs, np.int32)       # slabs that hold gradient, per dense parameter
            if self.n_wslabs < self.n_slabs:
                for a, b in self._wgrad_ranges:
                    per[a:b] = self.n_wslabs
            for a, b, c in self._block_bwd_ranges:                    # cr_stack_block_bwd: one slab per sequence pair
                per[a:b] = c
            if lay.n_dense and int(per.min()) < self.n_slabs:
                nb = (lay.n_dense + 255) // 256
                pad = np.zeros(nb * 256, np.int32)
                pad[:lay.n_dense] = per
                cnt = pad.reshape(nb, 256).max(1).astype(np.int32)   # a block shared with a parameter of more slabs reads them all (zeros)
                self.slab_counts = torch.from_numpy(cnt).to(self.dev)
                counts = self.slab_counts.data_ptr()
            ad = L.AdamDesc(self.P.data_ptr(), self.Mom.data_ptr(), self.Vel.data_ptr(), self.Gt.data_ptr(), self.Gs.data_ptr(),
                            lay.n_table, lay.n_dense, self.n_slabs, float(self.hp.lr), 0.9, 0.98, 1e-8, self.state.data_ptr(),
                            snap, tsnap, l2, n_l2, *lazy, counts)
            self._tgd, self._tgrad = None, None
            if self.use_index:
                # the table section's gradient by gather (castrec.h cr_tgrad_desc): inside the plain step's Adam launch; as a launch of
                # its own (cr_table_grad -> Gt) in front of the data-parallel exchange, and for grads()
                assert self._tg_rows is not None, "occurrence index: no gradient-row buffer for the item table's seq lookup"
                rows, rows2, ld_rows, scale = self._tg_rows
                se = self.seq_emb
                self._tgd = L.TgradDesc(self.batch_buf.data_ptr() + 4 * self.index_off, None, 0, 0, 0, self.state.data_ptr() + 4 * 4,
                                        self.index_lay, rows.data_ptr(), rows2.data_ptr() if rows2 is not None else None, ld_rows, scale,
                                        se.data_ptr(), se.shape[1], self._head_coef(), self.D, self._tg_part.data_ptr(), self._tg_tickets.data_ptr())
                ad.tg = C.pointer(self._tgd)
                self._tgrad = ("cr_table_grad", L.lib.cr_table_grad, (C.byref(self._tgd), self.Gt.data_ptr()))
            self._adam = ("cr_adam_step", L.lib.cr_adam_step, (C.byref(ad),))
            ad1 = L.AdamDesc(self.P.data_ptr(), self.Mom.data_ptr(), self.Vel.data_ptr(), self.Gt.data_ptr(),
                             self.Gflat.data_ptr() + 4 * lay.n_table, lay.n_table, lay.n_dense, 1, float(self.hp.lr), 0.9, 0.98,
                             1e-8, self.state.data_ptr(), self.Gflat.data_ptr() + 4 * lay.n_total, tsnap, l2, n_l2, *lazy, None)
            self._adam_flat = ("cr_adam_step", L.lib.cr_adam_step, (C.byref(ad1),))
            self._reduce = ("cr_reduce_slabs", L.lib.cr_reduce_slabs,
                            (self.Gs.data_ptr(), self.n_slabs, lay.n_dense, self.Gflat.data_ptr() + 4 * lay.n_table,
                             self.state.data_ptr(), self.Gflat.data_ptr() + 4 * lay.n_total, counts))
            # l2_emb != 0: the penalty of the CURRENT parameters, just before Adam moves them (one more launch; every
            # reference run uses 0.0)
            self._l2 = ("cr_l2_penalty", L.lib.cr_l2_penalty, (self.P.data_ptr(), n_l2, 0.5 * l2, self.state.data_ptr())) if n_l2 else None
            self._keep += [ad, ad1]

    # ---- running ---------------------------------------------------------------------------------
    def set_batch(self, seq, pos=None, neg=None, time=None, hours=None, days=None):
        """Copies one batch ([B,T] int arrays, host or device) into the static input buffers.  Host arrays are range
        checked first (the kernels index the tables with these ids unchecked; tf.nn.embedding_lookup raises
        InvalidArgument for an id outside its table -- e.g. unsorted timestamps give negative time bins, and
        sampler.py:66 produces bins up to 200 under --log_scale whatever --max_bins says).  CASTREC_NO_ID_CHECK=1 skips it."""
        limits = dict(seq=self.itemnum, pos=self.itemnum, neg=self.itemnum, time=self.hp.max_bins, hours=24, days=7)
        used = {"seq", "pos", "neg"} | ({"time"} if "time_emb" in self.layout.entries else set()) \
            | ({"hours", "days"} if "hours_emb" in self.layout.entries else set())
        for k, a in (("seq", seq), ("pos", pos), ("neg", neg), ("time", time), ("hours", hours), ("days", days)):
            if a is None:
                continue
            if self._check_ids and k in used and not isinstance(a, torch.Tensor):
                a = np.asarray(a)
                if a.size and (int(a.min()) < 0 or int(a.max()) > limits[k]):
                    raise ValueError("%s ids outside [0, %d] (min %d, max %d): the lookup table has %d rows"
                                     % (k, limits[k], int(a.min()), int(a.max()), limits[k] + 1))
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.int32))
            if self.use_index:
                # the host mirror of the static batch buffer: the index is built from it below, and ids + index go over in ONE copy
                i = self.ID_KEYS.index(k)
                self._host_slot[i * self.M:(i + 1) * self.M] = t.reshape(-1).to(torch.int32).cpu().numpy()
            else:
                self.ids[k].copy_(t.reshape(-1).to(torch.int32), non_blocking=True)
        if self.use_index:
            used = self._build_index(self._host_slot)
            self.batch_buf[:used].copy_(torch.from_numpy(self._host_slot[:used]))

    def _build_index(self, host_slot):
        """host_slot: int32 [slot_words] holding the six id rows; the occurrence index of its seq / pos / neg ids is written behind them."""
        M = self.M
        base = host_slot.ctypes.data
        L.check(L.lib.cr_index_build(self._ixb, base, base + 4 * M, base + 8 * M, base + 4 * self.index_off), "cr_index_build")
        return self.index_off + int(host_slot[self.index_off + 4])     # words of the slot in use: a copy may stop there

    def pack_slot(self, seq, pos, neg, time=None, hours=None, days=None, out=None):
        """One batch as a SLOT of the id ring / the staging area: int32 [slot_words] = the six id rows (+ the batch's occurrence index
        behind them where the engine uses one).  `out`: a host array to fill (e.g. a pinned ring slot's numpy view)."""
        host = np.zeros(self.slot_words, np.int32) if out is None else out
        for i, a in enumerate((seq, pos, neg, time, hours, days)):
            if a is None:
                host[i * self.M:(i + 1) * self.M] = 0
            else:
                host[i * self.M:(i + 1) * self.M] = np.asarray(a).reshape(-1)
        if self.use_index:
            self._build_index(host)
        return host

    def load_slot(self, slot):
        """A packed slot (device tensor [slot_words], pack_slot's layout) into the static batch buffer: ids AND index."""
        self.batch_buf.copy_(slot.reshape(-1))

    def set_step(self, k=1):
        """The next launch runs as step number k (dropout keys, Adam bias correction); loss sums cleared."""
        self.state[:4].zero_()
        self.state[8:].zero_()
        self.state[4:5].view(torch.int32)[0] = k
        if getattr(self, "_feed_ring", None) is not None:
            if self._feed_have:
                raise RuntimeError("set_step(): %d fed batches are waiting (their ring slots follow the step number)" % self._feed_have)
            self._feed_next, self._feed_started, self._feed_first = int(k), False, int(k)
            torch.cuda.synchronize()
            self._feed_done.clear()
        if getattr(self, "_tg_tickets", None) is not None:
            self._tg_tickets.zero_()                     # (every launch leaves them zero; a failed one might not)
        if getattr(self, "lazy_flags", None) is not None:
            # row-sparse Adam claims a row with atomicExch(flag, step) != step: a flag left by an earlier run of the same
            # step number (counter moved back, checkpoint loaded) would read as "already claimed" and skip the row
            self.lazy_flags.zero_()

    def step_number(self):
        """Number of the step the next launch will run (1 + completed optimiser steps)."""
        return int(self.state[4:5].view(torch.int32)[0])

    def _run(self, lst, stream):
        for name, fn, args in lst:
            rc = fn(*args, stream)
            if rc != 0:
                raise RuntimeError("castrec %s failed (%d): %s" % (name, rc, L.lib.cr_last_error().decode()))

    def launch_step(self, apply=True, between=None):
        """forward -> backward -> [between()] -> Adam, on the current stream.  state[4] holds the number of the step
        being run; Adam ends the step (zeroes the loss sums, advances the counter).  Without Adam (apply=False) the
        state is left as the kernels wrote it: call set_step() before running again."""
        s = torch.cuda.current_stream().cuda_stream
        self._run(self.fwd, s)
        if self.training:
            self._run(self.bwd, s)
            if between is not None:
                between()
            if apply:
                self._run(([self._l2] if self._l2 else []) + [self._adam], s)
                if self._id_ring is not None and self.lazy_adam:     # (row-sparse Adam reads this step's ids: the batch moves behind it)
                    self._run([self._ring_next], s)

    # ---- the id batches of the coming steps, resident in HBM -------------------------------------
    def use_id_ring(self, ring):
        """`ring`: int32 device tensor [n_slots, 6, M] (rows in ID_KEYS order), or None to go back to set_batch() per step.
        Slot (k mod n_slots) holds the batch of step number k.  From the next launch_step() / capture() on, a step ends by moving
        the batch of the following step into the static id buffers (inside the cr_adam_step launch): the caller sets the first
        batch with set_batch() and keeps the slots of the coming steps filled -- no copy between two steps."""
        ads = [self._adam[2][0]._obj, self._adam_flat[2][0]._obj]      # the plain step's Adam and the data-parallel one (on the reduced bucket)
        if ring is None:
            self._id_ring = None
            for ad in ads:
                ad.ids_ring, ad.ids_ring_slots, ad.ids_slot_elems, ad.ids_dst, ad.ids_copy_elems = None, 0, 0, None, 0
            if self._tgd is not None:
                self._tgd.ring, self._tgd.ring_slots, self._tgd.slot_words, self._tgd.index_off = None, 0, 0, 0
            return
        assert ring.dtype == torch.int32 and ring.is_cuda and ring.is_contiguous(), ring.shape
        assert int(np.prod(ring.shape[1:])) == self.slot_words, \
            "ring slots of %s words, the engine's slots have %d (pack_slot: 6 x %d ids%s)" % (tuple(ring.shape[1:]), self.slot_words, self.M, " + the occurrence index" if self.use_index else "")
        self._id_ring = ring
        if self.lazy_adam:
            self._ring_next = ("cr_ids_ring_next", L.lib.cr_ids_ring_next,
                               (ring.data_ptr(), int(ring.shape[0]), 6 * self.M, self.ids_all.data_ptr(), self.state.data_ptr() + 4 * 11))
        else:
            # extra workgroups of the Adam launch move the batch (castrec.h cr_adam_desc.ids_ring): no launch of its own.  (A forked
            # graph branch beside Adam was measured first: the fork and join cost 17 us per step, four times the copy they hid.)
            # Only the ids move: the occurrence index of a step is read where it lies, in the ring slot of that step.
            for ad in ads:
                ad.ids_ring, ad.ids_ring_slots, ad.ids_slot_elems, ad.ids_dst = ring.data_ptr(), int(ring.shape[0]), self.slot_words, self.ids_all.data_ptr()
                ad.ids_copy_elems = self.ids_words
            if self._tgd is not None:
                self._tgd.ring, self._tgd.ring_slots, self._tgd.slot_words, self._tgd.index_off = ring.data_ptr(), int(ring.shape[0]), self.slot_words, self.index_off

    # ---- host batches fed AHEAD of the steps that use them --------------------------------------
    def enable_feed(self, n_slots=8, steps_per_graph=1):
        """Turns on the pipelined input path of a training engine: feed() packs a host batch into a pinned buffer and sends
        it -- ONE copy over PCIe, on a copy stream, while earlier steps run -- into a slot of a device ring; train_fed() runs the
        step of the oldest waiting batch, whose tail moves the following batch into the static id buffers (use_id_ring).  With
        one batch fed ahead nothing stands between two steps on the device.  (set_batch() per step: six pageable copies on the
        compute stream, 0.45 ms per step at the headline shape against 0.37 for the device step, tools/e2e_rate.py.)"""
        assert self.training and n_slots >= 4
        # an event behind every `_feed_every`-th step tells feed() that a slot's last reader has finished (behind EVERY step it
        # cost 5 us per step, tools/probes/ev_cost.py); a slot is rewritten n_slots steps after its batch ran
        self._feed_every = 4 if n_slots >= 8 else 1
        self._feed_ring = torch.zeros(n_slots, self.slot_words, dtype=torch.int32, device=self.dev)
        self._feed_host = torch.zeros(n_slots, self.slot_words, dtype=torch.int32).pin_memory()
        self._feed_np = self._feed_host.numpy()
        self._feed_stream = torch.cuda.Stream()
        self._feed_h2d = [None] * n_slots            # per slot: its host -> device copy has finished
        self._feed_done = {}                         # step number -> event behind that step's launch
        recapture = self.graph is not None
        self.use_id_ring(self._feed_ring)
        if recapture:
            # (the ring's address travels in the Adam launch's arguments).  steps_per_graph > 1: train_fed() runs that many steps per
            # graph launch whenever as many batches wait (capture()); the ring must hold them beside the slots being refilled
            assert steps_per_graph == 1 or n_slots >= 2 * steps_per_graph + 4, "ring too small for %d steps per launch" % steps_per_graph
            self.capture(n_steps=steps_per_graph if not self.lazy_adam else 1)
        self._feed_next, self._feed_have, self._feed_started = self.step_number(), 0, False
        self._feed_first = self._feed_next           # steps before it did not read the ring
        used = {"seq", "pos", "neg"} | ({"time"} if "time_emb" in self.layout.entries else set()) \
            | ({"hours", "days"} if "hours_emb" in self.layout.entries else set())
        limits = dict(seq=self.itemnum, pos=self.itemnum, neg=self.itemnum, time=self.hp.max_bins, hours=24, days=7)
        self._feed_limits = [(i, k, limits[k]) for i, k in enumerate(self.ID_KEYS) if k in used]

    def feed(self, seq, pos, neg, time=None, hours=None, days=None):
        """Queues the batch of the next step that has none yet (at most 5 wait at a time in the ring of 8, n_slots - 1 in a ring
        of fewer than 8 slots).  Ids are range checked as set_batch() does."""
        n = self._feed_ring.shape[0]
        # a free slot beside the one the running step's tail reads, and a recorded step in [k - n, last launched]
        if self._feed_have > min(n - self._feed_every, n - 2):
            raise RuntimeError("feed(): %d batches are waiting already (ring of %d slots)" % (self._feed_have, n))
        k = self._feed_next
        slot = k % n
        if self._feed_h2d[slot] is not None:
            self._feed_h2d[slot].synchronize()       # the pinned buffer's previous copy (n batches ago) has left it
        host = self._feed_np[slot]
        hid = host[:self.ids_words].reshape(6, self.M)
        for i, a in enumerate((seq, pos, neg, time, hours, days)):
            if a is None:
                hid[i] = 0
            else:
                hid[i] = np.asarray(a).reshape(-1)
        if self._check_ids:
            lo, hi = hid.min(axis=1), hid.max(axis=1)
            for i, key, lim in self._feed_limits:
                if lo[i] < 0 or hi[i] > lim:
                    raise ValueError("%s ids outside [0, %d] (min %d, max %d): the lookup table has %d rows" % (key, lim, lo[i], hi[i], lim + 1))
        used = self.slot_words
        if self.use_index:
            used = self._build_index(host)               # (after the range check: the builder indexes its work arrays with these ids)
        # the slot held batch k - n: read by the tail of step k - n - 1 or, where that step ran with nothing fed ahead, by step
        # k - n's own copy into the static buffers -- the new copy waits for the launch of step k - n
        # (a HOST wait, like the two in train_fed: step k - n ended long ago, while a device-side wait between two streams
        # costs ~10 us of the waiting stream's time on this stack -- measured as 17 us per step for a forked graph branch)
        if k - n >= self._feed_first:
            behind = [j for j in self._feed_done if j >= k - n]
            if behind:
                self._feed_done[min(behind)].synchronize()
            else:
                # no recorded step at or behind the slot's last reader (cannot happen while train_fed records what it says it
                # records; never skip the wait silently): the compute stream itself
                torch.cuda.current_stream().synchronize()
        with torch.cuda.stream(self._feed_stream):
            self._feed_ring[slot][:used].copy_(self._feed_host[slot][:used], non_blocking=True)      # (the index's capacity beyond its use stays behind)
            ev = torch.cuda.Event()
            ev.record()
        self._feed_h2d[slot] = ev
        self._feed_next += 1
        self._feed_have += 1

    def train_fed(self, max_steps=None):
        """Runs the step of the oldest fed batch -- or, where capture(n_steps=G) has made a graph of G steps and at least G batches
        wait (and max_steps allows it), those G steps in one launch.  Returns the number of steps run."""
        if self._feed_have < 1:
            raise RuntimeError("train_fed(): no batch has been fed")
        n = self._feed_ring.shape[0]
        k = self._feed_next - self._feed_have
        G = self.graph_steps if (getattr(self, "graph_multi", None) is not None and self.graph is not None) else 1
        if self._feed_have < G or (max_steps is not None and max_steps < G):
            G = 1
        cur = torch.cuda.current_stream()
        if not self._feed_started:
            # nobody moved this batch into the static buffers (first step, or the step before ran with no batch fed ahead)
            self._feed_h2d[k % n].synchronize()
            self.ids_all.copy_(self._feed_ring[k % n][:self.ids_words].view(6, self.M))
        # the tail of step k + j - 1 moves batch k + j: those that are fed (a step or more ago: long there) must have arrived
        for j in range(1, min(G, self._feed_have - 1) + 1):
            self._feed_h2d[(k + j) % n].synchronize()
        self._feed_started = self._feed_have >= G + 1
        if G > 1:
            self.graph_multi.launch()
        elif self.graph is not None:
            self.graph.launch()
        else:
            self.launch_step()
        last = k + G - 1
        # an event behind every step number that is a multiple of _feed_every -- feed()'s admission bound counts on one in every
        # window of that many steps.  A launch of G steps records ONE event at its end and files it under every such step it
        # covers (an upper bound for each of them) and under its last step.
        marks = [j for j in range(k, last + 1) if j % self._feed_every == 0]
        if G > 1 and last not in marks:
            marks.append(last)
        if marks:
            ev = torch.cuda.Event()
            ev.record(cur)
            for j in marks:
                self._feed_done[j] = ev
            for old in [j for j in self._feed_done if j < last - n]:
                del self._feed_done[old]
        self._feed_have -= G
        return G

    # ---- data-parallel pieces (castrec_amd.dist drives them around an RCCL all-reduce) ----------
    def launch_backward_to_flat(self):
        """forward -> backward -> slabs collapsed into Gflat (+ loss stats in its tail)."""
        s = torch.cuda.current_stream().cuda_stream
        self._run(self.fwd, s)
        self._run(self.bwd, s)
        self._run(([self._tgrad] if self._tgrad else []) + [self._reduce], s)

    def launch_adam_from_flat(self):
        """Adam on the (all-reduced) flat bucket; the global loss statistics are read from its tail."""
        self._run(([self._l2] if self._l2 else []) + [self._adam_flat], torch.cuda.current_stream().cuda_stream)

    def capture_dp_phases(self):
        """Three HIP graphs for a data-parallel step: A1 = forward + the backward up to the last launch that adds to the table
        gradient, A2 = the rest of the backward + the slab collapse into the flat bucket, B = Adam on the (reduced) bucket.
        The table exchange runs between A1 and B beside A2 (castrec_amd.dist.DataParallel)."""
        s0 = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(s0)
        k = self.bwd_table_done
        # (occurrence index: the table gradient is WRITTEN into the bucket by cr_table_grad behind the last launch that leaves rows for it)
        progs = [self.fwd + self.bwd[:k] + ([self._tgrad] if self._tgrad else []), self.bwd[k:] + [self._reduce],
                 ([self._l2] if self._l2 else []) + [self._adam_flat]]
        graphs = []
        with torch.cuda.stream(side):
            for prog in progs:
                g = O.Graph()
                g.begin()
                self._run(prog, torch.cuda.current_stream().cuda_stream)
                g.end()
                graphs.append(g)
        s0.wait_stream(side)
        self.dp_graphs, self.dp_progs, self._graph_stream = graphs, progs, side
        return graphs

    def capture(self, dp=False, n_steps=1):
        """Captures launch_step() into a HIP graph (inputs are read from the static id buffers).
        n_steps > 1 (with use_id_ring(): every step's tail moves the next step's batch into the static buffers on the device) captures
        that many consecutive steps into a second graph, `graph_multi`: between two launches of a graph the device idles for the
        7-9 us the command processor needs to start the next one (rocprofv3: gaps of 0.2 us inside a step's graph, 8.7 us between two
        steps); n steps per launch pay that once."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = O.Graph()
        with torch.cuda.stream(side):
            g.begin()
            if dp:
                self.launch_backward_to_flat()
            else:
                self.launch_step()
            g.end()
        self.graph_multi, self.graph_steps = None, 1
        if n_steps > 1:
            assert not dp and self.training and self._id_ring is not None and not self.lazy_adam, "several steps per graph need the id ring (use_id_ring / enable_feed)"
            gm = O.Graph()
            with torch.cuda.stream(side):
                gm.begin()
                for _ in range(n_steps):
                    self.launch_step()
                gm.end()
            self.graph_multi, self.graph_steps = gm, int(n_steps)
        torch.cuda.current_stream().wait_stream(side)
        self.graph, self._graph_stream = g, side
        return g

    def train_step(self, seq, pos, neg, time=None, hours=None, days=None):
        self.set_batch(seq, pos, neg, time, hours, days)
        if self.graph is not None:
            self.graph.launch()
        else:
            self.launch_step()

    def loss_auc(self):
        s = self.state.cpu()
        return float(s[5]), float(s[6])

    def forward_eval(self, seq, time=None, hours=None, days=None):
        assert not self.training
        self.set_batch(seq, None, None, time, hours, days)
        self.launch_step()

    def test_logits(self, cand):
        """[B, n_cand] logits of the last position (sasrec.py:93-97); cand int32 device tensor [B, n_cand]."""
        out = torch.empty(self.B, cand.shape[1], dtype=torch.float32, device=self.dev)
        O.test_logits(self.seq_emb, self.seq_emb.shape[1], self.p("item_emb"), cand, self.B, self.T, self.D, out)
        return out

    # ---- parameter / gradient access (tests, checkpoints) ----------------------------------------
    def get_params(self):
        return {n: self.layout.view(self.P, n).detach().clone() for n in self.layout.logical_names()}

    def load_params(self, d):
        for n in self.layout.logical_names():
            self.layout.view(self.P, n).copy_(torch.tensor(np.asarray(d[n]), dtype=torch.float32).to(self.dev))

    def grads(self):
        """Normalised gradients by logical name, valid after launch_step(apply=False)."""
        n = float(self.state[2])
        if self._tgrad is not None:
            # the plain step never materialises the table gradient (cr_adam_step gathers it row by row): form it now, by the same gather
            self.Gt.zero_()
            self._run([self._tgrad], torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        flat = torch.cat([self.Gt, self.Gs.sum(0)[:self.layout.n_dense]]) / max(n, 1.0)
        return {k: self.layout.view(flat, k).clone() for k in self.layout.logical_names()}

    def n_launches(self):
        """C-ABI calls per step (an entry may launch more than one kernel: n_kernel_launches)."""
        return len(self.fwd) + len(self.bwd) + (1 if self.training else 0) + (1 if self.training and self._l2 else 0)

    def n_kernel_launches(self):
        """Kernel launches per step, as rocprofv3 counts them: cr_stack_fwd launches once per block when a sequence gets two
        workgroups (B <= 160, more than one tile: cr_stack.hip, g_stack_pair_max_b), cr_stack_block_bwd once per min(B, n_slabs)
        sequences; every other entry of the fused path is one kernel."""
        n = 0
        pair_max = int(os.environ.get("CASTREC_STACK_PAIR_MAX_B", 160))
        for name, _, args in self.fwd + self.bwd:
            if name in ("cr_stack_fwd", "cr_stack_fwd_head"):
                sd = args[0]._obj
                n += sd.n_blocks if (self.B <= pair_max and self.T > 16) else 1
            elif name == "cr_stack_block_bwd":
                per = min(self.B, self.n_slabs)
                n += -(-self.B // per)
            else:
                n += 1
        return n + (1 if self.training else 0) + (1 if self.training and self._l2 else 0)
