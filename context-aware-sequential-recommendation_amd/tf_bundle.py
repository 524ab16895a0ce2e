"""Reader for the reference's saved checkpoints (TensorFlow-1.x `tf.train.Saver` bundles:
`model.ckpt.index` + `model.ckpt.data-00000-of-00001`, written by main.py:231-233 of the reference) and the
mapping of its variable names onto this package's logical parameter names.

No TensorFlow needed: the index is a LevelDB-format table (uncompressed blocks, prefix-compressed keys) whose
values are `BundleEntryProto` messages {dtype, shape, shard_id, offset, size, crc32c}; the data file holds the raw
little-endian tensors.  Only what these checkpoints use is implemented (one shard, no slices, no compression).

Name mapping (reference scopes -> logical names; verified against the index files of the eight model classes the
reference ships under saved_models/ml-1m.txt, tests/golden/tf_index/):
  SASRec/input_embeddings/lookup_table -> item_emb          SASRec/dec_pos/lookup_table -> pos_emb
  CONTEXT/time_embeddings/lookup_table -> time_emb          INPUT-CONTEXT/{hours,days}_embeddings/... -> hours_emb, days_emb
  SASRec/num_blocks_i/{ln, ln_1}        -> trunk.i.{ln1, ln2}          (Variable = beta, Variable_1 = gamma, modules.py:74-77)
  CONTEXT/timeseq_num_blocks_i/{ln_1, ln_2} -> ctx_time.i.{ln1, ln2}   (`ln` there is created but never used: cast_1.py:45)
  .../self_attention/dense{,_1,_2}/{kernel,bias} -> wq/bq, wk/bk, wv/bv;  .../multihead_attention/conv1d{,_1} -> w1/b1, w2/b2
  SASRec/ln -> trunk.lnf,  CONTEXT/ln -> ctx_time.lnf,  SASRec/MLP/dense{,_1} -> mlp.{w1,b1,w2,b2}
Adam slots (`.../Adam`, `.../Adam_1`), beta*_power and global_step are skipped."""
import os
import re
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


def _varint(b, i):
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _block(data, off, size):
    if data[off + size] != 0:
        raise ValueError("compressed table blocks are not supported (type %d)" % data[off + size])
    blk = data[off:off + size]
    nrest = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrest
    i, key, out = 0, b"", []
    while i < end:
        shared, i = _varint(blk, i)
        non_shared, i = _varint(blk, i)
        vlen, i = _varint(blk, i)
        key = key[:shared] + blk[i:i + non_shared]
        i += non_shared
        out.append((key, blk[i:i + vlen]))
        i += vlen
    return out


def _fields(msg):
    """(field number, wire type, value) triples of one protobuf message."""
    i = 0
    while i < len(msg):
        tag, i = _varint(msg, i)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, i = _varint(msg, i)
        elif wt == 1:
            v = msg[i:i + 8]; i += 8
        elif wt == 5:
            v = msg[i:i + 4]; i += 4
        elif wt == 2:
            n, i = _varint(msg, i)
            v = msg[i:i + n]; i += n
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _entry(value):
    e = dict(dtype=None, shape=[], shard=0, offset=0, size=0)
    for f, wt, v in _fields(value):
        if f == 1:
            e["dtype"] = v
        elif f == 2:                                   # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}
            for f2, _, dim in _fields(v):
                if f2 == 2:
                    e["shape"].append(next((x for g, w, x in _fields(dim) if g == 1 and w == 0), 0))
        elif f == 3:
            e["shard"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 7:
            raise ValueError("sliced tensors are not supported")
    return e


def read_index(index_path):
    """{variable name: dict(dtype, shape, shard, offset, size)} of a `*.index` file."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow bundle index (bad table magic)" % index_path)
    footer = data[-48:]
    _, i = _varint(footer, 0)
    _, i = _varint(footer, i)                           # metaindex handle (unused)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    out = {}
    for _, handle in _block(data, ioff, isize):
        boff, j = _varint(handle, 0)
        bsize, j = _varint(handle, j)
        for key, value in _block(data, boff, bsize):
            if key:                                     # the empty key holds the BundleHeaderProto
                out[key.decode()] = _entry(value)
    return out


def load(prefix):
    """{variable name: ndarray} of the checkpoint `prefix` (e.g. .../model.ckpt)."""
    idx = read_index(prefix + ".index")
    shards = {e["shard"] for e in idx.values()}
    if shards - {0}:
        raise ValueError("multi-shard bundles are not supported")
    data_path = prefix + ".data-00000-of-00001"
    out = {}
    with open(data_path, "rb") as f:
        for name, e in idx.items():
            dt = _DTYPES.get(e["dtype"])
            if dt is None:
                raise ValueError("%s: unsupported dtype enum %s" % (name, e["dtype"]))
            f.seek(e["offset"])
            raw = f.read(e["size"])
            a = np.frombuffer(raw, dtype=np.dtype(dt).newbyteorder("<")).reshape(e["shape"])
            if a.nbytes != e["size"]:
                raise ValueError("%s: %d bytes on disk, shape %s" % (name, e["size"], e["shape"]))
            out[name] = a
    return out


_SKIP = re.compile(r"(/Adam(_1)?$)|(^beta[12]_power$)|(^global_step$)")
_FIXED = {"SASRec/input_embeddings/lookup_table": "item_emb", "SASRec/dec_pos/lookup_table": "pos_emb",
          "CONTEXT/time_embeddings/lookup_table": "time_emb",
          "INPUT-CONTEXT/hours_embeddings/lookup_table": "hours_emb", "INPUT-CONTEXT/days_embeddings/lookup_table": "days_emb",
          "SASRec/ln/Variable": "trunk.lnf.beta", "SASRec/ln/Variable_1": "trunk.lnf.gamma",
          "CONTEXT/ln/Variable": "ctx_time.lnf.beta", "CONTEXT/ln/Variable_1": "ctx_time.lnf.gamma",
          # cast_8 / cast_9: the hours and days stacks share the INPUT-CONTEXT scope, their final LayerNorms are its
          # first and second `normalize` (cast_8.py:75,95); cast_9's time stack lives under TEMPORAL-CONTEXT (cast_9.py:100-129)
          "INPUT-CONTEXT/ln/Variable": "ctx_hours.lnf.beta", "INPUT-CONTEXT/ln/Variable_1": "ctx_hours.lnf.gamma",
          "INPUT-CONTEXT/ln_1/Variable": "ctx_days.lnf.beta", "INPUT-CONTEXT/ln_1/Variable_1": "ctx_days.lnf.gamma",
          "TEMPORAL-CONTEXT/ln/Variable": "ctx_time.lnf.beta", "TEMPORAL-CONTEXT/ln/Variable_1": "ctx_time.lnf.gamma",
          "TEMPORAL-CONTEXT/time_embeddings/lookup_table": "time_emb",
          "SASRec/MLP/dense/kernel": "mlp.w1", "SASRec/MLP/dense/bias": "mlp.b1",
          "SASRec/MLP/dense_1/kernel": "mlp.w2", "SASRec/MLP/dense_1/bias": "mlp.b2"}
_LEAF = {"self_attention/dense/kernel": "wq", "self_attention/dense/bias": "bq",
         "self_attention/dense_1/kernel": "wk", "self_attention/dense_1/bias": "bk",
         "self_attention/dense_2/kernel": "wv", "self_attention/dense_2/bias": "bv",
         "multihead_attention/conv1d/kernel": "w1", "multihead_attention/conv1d/bias": "b1",
         "multihead_attention/conv1d_1/kernel": "w2", "multihead_attention/conv1d_1/bias": "b2"}
_LN = {"Variable": "beta", "Variable_1": "gamma"}


def logical_name(tf_name):
    """Logical parameter name of a reference variable; None for optimiser state and the unused LayerNorm pair of
    the context blocks (cast_1.py:45).  Raises KeyError for a name this mapping does not know."""
    if _SKIP.search(tf_name):
        return None
    if tf_name in _FIXED:
        return _FIXED[tf_name]
    m = re.match(r"^(SASRec/num_blocks_|CONTEXT/timeseq_num_blocks_|INPUT-CONTEXT/hours_seq_num_blocks_|"
                 r"INPUT-CONTEXT/days_seq_num_blocks_|TEMPORAL-CONTEXT/timeseq_num_blocks_)(\d+)/(.+)$", tf_name)
    if not m:
        raise KeyError("unknown checkpoint variable %r" % tf_name)
    # (stack prefix, does the block create an unused LayerNorm first -- cast_1.py:45 does, cast_8 / cast_9 do not)
    stack, ctx = {"SASRec/num_blocks_": ("trunk", False), "CONTEXT/timeseq_num_blocks_": ("ctx_time", True),
                  "INPUT-CONTEXT/hours_seq_num_blocks_": ("ctx_hours", False), "INPUT-CONTEXT/days_seq_num_blocks_": ("ctx_days", False),
                  "TEMPORAL-CONTEXT/timeseq_num_blocks_": ("ctx_time", False)}[m.group(1)]
    prefix = "%s.%s." % (stack, m.group(2))
    leaf = m.group(3)
    if leaf in _LEAF:
        return prefix + _LEAF[leaf]
    m2 = re.match(r"^(ln|ln_1|ln_2)/(Variable|Variable_1)$", leaf)
    if not m2:
        raise KeyError("unknown checkpoint variable %r" % tf_name)
    which = {"ln": 0, "ln_1": 1, "ln_2": 2}[m2.group(1)]
    if ctx:
        if which == 0:
            return None                                 # created by `normalize(self.tseq)` at cast_1.py:45, never used
        which -= 1
    elif which == 2:
        raise KeyError("unknown checkpoint variable %r" % tf_name)
    return prefix + ("ln1." if which == 0 else "ln2.") + _LN[m2.group(2)]


def to_logical(tensors):
    """{logical name: float32 ndarray} from {reference variable name: ndarray} (conv1d kernels [1,in,out] -> [in,out])."""
    out = {}
    for name, a in tensors.items():
        ln = logical_name(name)
        if ln is None:
            continue
        a = np.asarray(a, np.float32)
        if a.ndim == 3 and a.shape[0] == 1:
            a = a[0]
        if ln in out:
            raise ValueError("two checkpoint variables map to %s" % ln)
        out[ln] = np.ascontiguousarray(a)
    return out


def load_logical(prefix):
    return to_logical(load(prefix))


def load_logical_with_slots(prefix):
    """(params, adam_m, adam_v, completed_steps): the variables plus tf.train.AdamOptimizer's slots (`<var>/Adam` = m,
    `<var>/Adam_1` = v) under the same logical names, and the number of optimiser steps taken.

    The step count is the bundle's `global_step` (sasrec.py:119-121 hands it to minimize(), which increments it once per
    step).  beta1_power is only the fallback: TF1's AdamOptimizer initialises it to beta1 and multiplies it AFTER each
    step, so after t steps it holds 0.9 ** (t + 1) -- and in fp32 it underflows to exactly 0 after ~980 steps (every run
    the reference ships ended at global_step 9400 with beta1_power == 0.0).  Slots missing from the bundle come back as
    empty dicts; a bundle with neither counter gives steps = 0."""
    import math
    raw = load(prefix)
    params = to_logical({k: v for k, v in raw.items() if not _SKIP.search(k)})
    m = to_logical({k[:-len("/Adam")]: v for k, v in raw.items() if k.endswith("/Adam")})
    v = to_logical({k[:-len("/Adam_1")]: a for k, a in raw.items() if k.endswith("/Adam_1")})
    steps = 0
    if "global_step" in raw:
        steps = int(np.asarray(raw["global_step"]).reshape(-1)[0])
    elif "beta1_power" in raw:
        b1p = float(np.asarray(raw["beta1_power"]).reshape(-1)[0])
        if 0.0 < b1p < 1.0:
            steps = max(0, int(round(math.log(b1p) / math.log(0.9))) - 1)
    return params, m, v, steps
