"""Data model, split and evaluation -- the reference's util.py surface on top of the native path.

  data_partition(fpath, log_scale=False) -> [train, valid, test, usernum, itemnum, ratingnum]   (util.py:204-227)
  evaluate(model, dataset, args, sess=None) / evaluate_valid(...) -> (NDCG@10, HR@10)          (util.py:230-430)

Events are ``(item, rating, unix_ts)`` tuples (the reference wraps them in UserItems objects,
util.py:32-43; objects with .item/.rating/.timestamp_raw are accepted everywhere as well).
Evaluation is batched (the reference runs one B=1 session call per user) but draws the 100 negatives
of every user from the SAME global numpy stream in the SAME order, so candidate sets -- and therefore
the metrics for given weights -- are identical to the reference's.
"""
import base64
import math
import os
import random
import struct
import zlib

import numpy as np
import torch

from .synth import Corpus, from_dict


def _ev(e):
    if isinstance(e, (tuple, list)):
        return e[0], e[1], e[2]
    return e.item, e.rating, e.timestamp_raw


def plot_attention_weights(attention_weights, path):
    """util.py:46-54: heat map of a [T, T] attention matrix saved as ``<path>/attention_weights.svg`` ('hot' colour
    map over the matrix's own min..max, nearest-neighbour cells, title "Attention weights").  Written directly (an
    RGB PNG embedded in an SVG) so the evaluation path needs no plotting library."""
    a = np.asarray(attention_weights, np.float64)
    if a.ndim != 2:
        raise ValueError("attention_weights must be a [T, T] matrix, got shape %s" % (a.shape,))
    lo, hi = float(a.min()), float(a.max())
    x = (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)
    k1, k2 = 0.365079, 0.746032                             # breakpoints of the 'hot' map: black-red-yellow-white
    rgb = np.stack([np.clip(x / k1, 0, 1), np.clip((x - k1) / (k2 - k1), 0, 1), np.clip((x - k2) / (1 - k2), 0, 1)], -1)
    px = np.rint(rgb * 255).astype(np.uint8)
    h, w = a.shape
    raw = b"".join(b"\x00" + px[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))
    side, top = 600, 40
    svg = ('<svg xmlns="http://www.w3.org/2000/svg" xmlns:xlink="http://www.w3.org/1999/xlink" width="%d" height="%d">'
           '<rect width="100%%" height="100%%" fill="white"/>'
           '<text x="%d" y="26" font-family="sans-serif" font-size="18" text-anchor="middle">Attention weights</text>'
           '<image x="20" y="%d" width="%d" height="%d" preserveAspectRatio="none" style="image-rendering:pixelated" '
           'xlink:href="data:image/png;base64,%s"/></svg>\n'
           % (side + 40, side + top + 20, (side + 40) // 2, top, side, side, base64.b64encode(png).decode()))
    out = os.path.join(path, "attention_weights.svg")
    with open(out, "w") as f:
        f.write(svg)
    return out


def hour_of(ts):
    """UTC hour + 1 in 1..24 (util.py:28)."""
    return (int(ts) // 3600) % 24 + 1


def day_of(ts):
    """ISO weekday 1..7, Monday = 1 (util.py:14-22,27); 1970-01-01 was a Thursday."""
    return ((int(ts) // 86400) + 3) % 7 + 1


def get_timedelta_bin(ts, bin_in_hours=48, max_bins=200, log_scale=False, min_ts=None, max_ts=None):
    """util.py:73-120 (ts = time delta in seconds)."""
    if log_scale:
        lo, hi, t = float(min_ts) + 1, float(max_ts) + 1, float(ts) + 1
        bin_size = (np.log(hi) - np.log(lo)) / max_bins
        time_bin = math.floor(np.log(t) / bin_size)
    else:
        time_bin = math.floor(float(ts) // 3600 / bin_in_hours)
    return int(max_bins if time_bin > max_bins else time_bin)


def get_delta_range(User):
    """util.py:123-160 -> (min, 90th percentile) of last_ts - ts."""
    deltas = []
    for _, ev in User.items():
        if not ev:
            continue
        last = _ev(ev[-1])[2]
        deltas += [float(last - _ev(e)[2]) for e in ev]
    deltas = np.array(deltas)
    return np.amin(deltas), np.percentile(deltas, 90)


def get_users(fpath):
    """util.py:163-182: 4-column ``user item rating ts`` text, time-sorted per user.

    Also reads the original SASRec format the reference ships (`data/Video.txt`, parsed by
    `baselines/SASRec/util.py:16-27`): 2 columns ``user item`` (or 3: ``user item ts``), events in order.  Those files carry
    no ratings and (2 columns) no time: rating 0 and one synthetic day per event from a fixed origin are filled in, so
    the context-free models run on them unchanged and the context models see a regular cadence."""
    usernum = itemnum = 0
    ratingnum = 0
    User = {}
    origin = 956_700_000                                  # first ml-1m timestamp, any fixed origin will do
    with open(fpath, "r") as f:
        for line in f:
            parts = line.rstrip().split(" ")
            if len(parts) == 4:
                u, i, r, t = int(parts[0]), int(parts[1]), float(parts[2]), int(parts[3])
            elif len(parts) == 3:
                u, i, r, t = int(parts[0]), int(parts[1]), 0.0, int(float(parts[2]))
            elif len(parts) == 2:
                u, i, r = int(parts[0]), int(parts[1]), 0.0
                t = origin + 86400 * len(User.get(u, ()))
            else:
                raise ValueError("%s: expected 2, 3 or 4 space-separated columns, got %r" % (fpath, line))
            usernum, itemnum, ratingnum = max(u, usernum), max(i, itemnum), max(r, ratingnum)
            User.setdefault(u, []).append((i, r, t))
    return User, usernum, itemnum, ratingnum


def data_partition(fpath, log_scale=False):
    """util.py:204-227: leave-last-two-out split."""
    User, usernum, itemnum, ratingnum = get_users(fpath)
    return partition(User, usernum, itemnum, ratingnum)


def partition(User, usernum, itemnum, ratingnum=5.0):
    train, valid, test = {}, {}, {}
    for u, ev in User.items():
        if len(ev) < 3:
            train[u], valid[u], test[u] = list(ev), [], []
        else:
            train[u], valid[u], test[u] = list(ev[:-2]), [ev[-2]], [ev[-1]]
    return [train, valid, test, usernum, itemnum, ratingnum]


def train_corpus(train, usernum, itemnum):
    """CSR view of the train split for the native sampler."""
    return from_dict({u: [_ev(e) for e in ev] for u, ev in train.items()}, usernum, itemnum)


# ---------------------------------------------------------------------------------------------------
def _eval_inputs(train, valid, test, u, mode, args, itemnum, min_td, max_td):
    """One user's predict() inputs (util.py:245-315 / 355-415); candidates come from np.random (global)."""
    target = test if mode == "test" else valid
    if len(train[u]) < 1 or len(target[u]) < 1:
        return None
    T = args.maxlen
    seq = np.zeros(T, np.int32); timeseq = np.zeros(T, np.int32)
    hours = np.zeros(T, np.int32); days = np.zeros(T, np.int32)
    orig = [None] * T
    idx = T - 1
    if mode == "test":                                    # util.py:255-264: the valid item ends the sequence
        it, _, ts = _ev(valid[u][0])
        seq[idx], orig[idx], hours[idx], days[idx] = it, ts, hour_of(ts), day_of(ts)
        idx -= 1
    for e in reversed(train[u]):                          # util.py:265-272
        it, _, ts = _ev(e)
        seq[idx], orig[idx], hours[idx], days[idx] = it, ts, hour_of(ts), day_of(ts)
        idx -= 1
        if idx == -1:
            break
    most_recent = orig[-1]                                # util.py:275-289
    for i, ts in enumerate(orig):
        if ts is not None:
            timeseq[i] = get_timedelta_bin(float(most_recent - ts), args.bin_in_hours, args.max_bins, bool(args.log_scale),
                                           min_td, max_td)
    rated = set(_ev(e)[0] for e in train[u]); rated.add(0)   # util.py:291-298
    item_idx = [_ev(target[u][0])[0]]
    for _ in range(100):
        t = np.random.randint(1, itemnum + 1)
        while t in rated:
            t = np.random.randint(1, itemnum + 1)
        item_idx.append(t)
    if getattr(args, "test_model", None):                 # util.py:300-315
        if not getattr(args, "test_seq_len", None):
            raise Exception("test_seq_len is not provided")
        n = min(args.test_seq_len, args.maxlen)
        seq[:-n] = 0; timeseq[:-n] = 0; hours[:-n] = 0; days[:-n] = 0
    return seq, timeseq, hours, days, np.asarray(item_idx, np.int32)


def _evaluate(model, dataset, args, mode, eval_batch=256):
    train, valid, test, usernum, itemnum = dataset[0], dataset[1], dataset[2], dataset[3], dataset[4]
    min_td, max_td = get_delta_range(train)               # util.py:234 / 345
    if usernum > 10000:                                   # util.py:241-244
        users = random.sample(range(1, usernum + 1), 10000)
    else:
        users = range(1, usernum + 1)
    rows = []
    for u in users:
        r = _eval_inputs(train, valid, test, u, mode, args, itemnum, min_td, max_td)
        if r is not None:
            rows.append((u,) + r)
    NDCG = HT = 0.0
    attn_sum, n_attn = None, 0
    for i in range(0, len(rows), eval_batch):
        chunk = rows[i:i + eval_batch]
        us = [c[0] for c in chunk]
        seq, ts, hrs, dys, cand = (np.stack([c[j] for c in chunk]) for j in range(1, 6))
        logits, attn = model.predict(None, us, seq, cand, timeseq=ts, hours_seq=hrs, days_seq=dys)
        pred = -np.asarray(logits, np.float64 if logits.dtype == np.float64 else np.float32)
        rank = pred.argsort(axis=1).argsort(axis=1)[:, 0]   # util.py:318-322
        for rk in rank:
            if rk < 10:                                   # util.py:326-328
                NDCG += 1 / np.log2(rk + 2)
                HT += 1
        if getattr(args, "test_model", None) and attn is not None:
            a = np.asarray(attn)
            attn_sum = a.sum(0) if attn_sum is None else attn_sum + a.sum(0)
            n_attn += a.shape[0]
    n = float(len(rows))
    if attn_sum is not None and mode == "test":
        avg = attn_sum / max(n_attn, 1)                                                 # util.py:334-336
        np.save(os.path.join(args.test_model, "attention_weights.npy"), avg)
        plot_attention_weights(avg, args.test_model)
    return NDCG / n, HT / n


def evaluate(model, dataset, args, sess=None):
    """util.py:230-339: test split (sequence = train + valid item, target = test item)."""
    return _evaluate(model, dataset, args, "test")


def evaluate_valid(model, dataset, args, sess=None):
    """util.py:342-430: validation split."""
    return _evaluate(model, dataset, args, "valid")
