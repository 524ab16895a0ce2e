"""Integer-path oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

numpy / pure-Python restatement of the reference's ``sampler.py`` and the
integer half of ``util.py``.  Every function cites the reference lines it
follows (paths relative to the reference repo root).  Pinned by the golden
vectors under ``tests/golden/`` which were generated from the reference's own
code (``tests/golden/make_golden.py``).

Corpus representation: ``{user_id: [(item, rating, unix_ts), ...]}`` -- the same
information the reference keeps in ``UserItems`` objects (``util.py:32-43``).
The random stream is the legacy numpy MT19937 stream (``np.random.seed`` /
``np.random.randint``, ``sampler.py:10,19,76``); a private
``np.random.RandomState`` carries the identical algorithm without touching the
global generator.
"""
from __future__ import annotations

import math
import random as _pyrandom

import numpy as np


# --------------------------------------------------------------------------
# util.py:14-43  TimeStamp / UserItems
# --------------------------------------------------------------------------
def hour_of(ts: int) -> int:
    """``int(strftime('%H')) + 1`` of the UTC datetime (util.py:28,39-40)."""
    return (int(ts) // 3600) % 24 + 1


def day_of(ts: int) -> int:
    """ISO weekday 1..7 (Monday=1) of the UTC datetime (util.py:14-22,27).

    1970-01-01 was a Thursday (=4): ((days + 3) % 7) + 1.
    """
    return ((int(ts) // 86400) + 3) % 7 + 1


# --------------------------------------------------------------------------
# util.py:57-120  get_bin_size / get_timedelta_bin
# --------------------------------------------------------------------------
def timedelta_bin(delta_s, bin_in_hours=48, max_bins=200, log_scale=False,
                  min_ts=None, max_ts=None) -> int:
    """util.py:73-120.  ``delta_s`` is a time delta in seconds."""
    if log_scale:
        lo = float(min_ts) + 1          # util.py:97-99
        hi = float(max_ts) + 1
        ts = float(delta_s) + 1
        bin_size = (np.log(hi) - np.log(lo)) / max_bins   # util.py:65-70
        time_bin = math.floor(np.log(ts) / bin_size)      # util.py:108-109
    else:
        time_bin = math.floor(float(delta_s) // 3600 / bin_in_hours)  # util.py:114
    if time_bin > max_bins:             # util.py:117-118
        time_bin = max_bins
    return int(time_bin)


# --------------------------------------------------------------------------
# util.py:123-160  get_delta_range
# --------------------------------------------------------------------------
def delta_range(corpus):
    """(min, 90th percentile) of (last_ts - ts) over every event (util.py:149-160).

    The ``max_percentile`` argument of the reference is ignored there too
    (hard-coded 90, util.py:158)."""
    deltas = []
    for _, events in corpus.items():
        last = events[-1][2]
        for (_, _, t) in events:
            deltas.append(float(last - t))
    deltas = np.array(deltas)
    return np.amin(deltas), np.percentile(deltas, 90)


# --------------------------------------------------------------------------
# util.py:163-227  get_users / data_partition
# --------------------------------------------------------------------------
def read_events(fpath):
    """util.py:163-182: 4-column ``user item rating ts`` text."""
    corpus, usernum, itemnum, ratingnum = {}, 0, 0, 0
    with open(fpath, "r") as f:
        for line in f:
            u, i, r, t = line.rstrip().split(" ")
            u, i, r, t = int(u), int(i), float(r), int(t)
            usernum, itemnum, ratingnum = max(u, usernum), max(i, itemnum), max(r, ratingnum)
            corpus.setdefault(u, []).append((i, r, t))
    return corpus, usernum, itemnum, ratingnum


def partition(corpus):
    """Leave-last-two-out split (util.py:204-227)."""
    train, valid, test = {}, {}, {}
    for u, ev in corpus.items():
        if len(ev) < 3:
            train[u], valid[u], test[u] = list(ev), [], []
        else:
            train[u], valid[u], test[u] = list(ev[:-2]), [ev[-2]], [ev[-1]]
    return train, valid, test


# --------------------------------------------------------------------------
# sampler.py:9-81  random_neq / sample_function
# --------------------------------------------------------------------------
class SamplerOracle:
    """One worker's batch stream (sampler.py:16-81), single RandomState."""

    def __init__(self, train, usernum, itemnum, batch_size, maxlen,
                 bin_in_hours, max_bins, log_scale, seed,
                 min_timedelta=None, max_timedelta=None):
        self.train, self.usernum, self.itemnum = train, usernum, itemnum
        self.B, self.T = batch_size, maxlen
        self.bin_in_hours, self.max_bins, self.log_scale = bin_in_hours, max_bins, log_scale
        if min_timedelta is None:
            min_timedelta, max_timedelta = delta_range(train)   # sampler.py:106
        self.min_td, self.max_td = min_timedelta, max_timedelta
        self.rs = np.random.RandomState(seed)                   # sampler.py:76

    def _random_neq(self, l, r, s):                              # sampler.py:9-14
        t = self.rs.randint(l, r)
        while t in s:
            t = self.rs.randint(l, r)
        return t

    def sample(self):                                            # sampler.py:17-74
        T = self.T
        user = self.rs.randint(1, self.usernum + 1)
        while len(self.train[user]) <= 1:
            user = self.rs.randint(1, self.usernum + 1)
        ev = self.train[user]
        seq = np.zeros(T, np.int32); pos = np.zeros(T, np.int32); neg = np.zeros(T, np.int32)
        timeseq = np.zeros(T, np.int32); ratings = np.zeros(T, np.int32)
        hours = np.zeros(T, np.int32); days = np.zeros(T, np.int32)
        orig_ts = [None] * T
        nxt = ev[-1][0]
        idx = T - 1
        items = set(e[0] for e in ev)                            # sampler.py:42
        for (item, rating, ts) in reversed(ev[:-1]):             # sampler.py:44-58
            seq[idx] = item
            ratings[idx] = rating
            hours[idx] = hour_of(ts)
            days[idx] = day_of(ts)
            orig_ts[idx] = ts
            pos[idx] = nxt
            if nxt != 0:
                neg[idx] = self._random_neq(1, self.itemnum + 1, items)
            nxt = item
            idx -= 1
            if idx == -1:
                break
        most_recent = orig_ts[-1]                                # sampler.py:61
        for i, ts in enumerate(orig_ts):                         # sampler.py:62-72
            if ts is not None:
                d = float(most_recent - ts)
                if self.log_scale:
                    # NB: reference hard-codes max_bins=200 here (sampler.py:66)
                    timeseq[i] = timedelta_bin(d, 48, 200, True, self.min_td, self.max_td)
                else:
                    timeseq[i] = timedelta_bin(d, self.bin_in_hours, self.max_bins, False)
        return user, seq, pos, neg, timeseq, ratings, hours, days

    def next_batch(self):
        """Returns (user[B], seq, pos, neg, timeseq, ratings, hours, days) int32 [B,T]."""
        rows = [self.sample() for _ in range(self.B)]            # sampler.py:78-81
        cols = list(zip(*rows))
        out = [np.asarray(cols[0], dtype=np.int32)]
        out += [np.stack(c).astype(np.int32) for c in cols[1:]]
        return tuple(out)


# --------------------------------------------------------------------------
# util.py:230-430  evaluate / evaluate_valid  (sequence + candidate building)
# --------------------------------------------------------------------------
def eval_users(usernum, py_random=None):
    """util.py:241-244 / 351-354: all users, or 10000 drawn with ``random.sample``."""
    if usernum > 10000:
        rnd = py_random if py_random is not None else _pyrandom
        return rnd.sample(range(1, usernum + 1), 10000)
    return list(range(1, usernum + 1))


def eval_inputs(train, valid, test, u, mode, maxlen, itemnum, rs,
                bin_in_hours, max_bins, log_scale, min_td, max_td,
                test_seq_len=None):
    """Inputs of one ``model.predict`` call.

    ``mode='test'`` follows util.py:245-315 (sequence = train + the valid item,
    target = test item); ``mode='valid'`` follows util.py:355-415 (sequence =
    train, target = valid item).  ``rs`` is the main-process RandomState that
    replaces the global numpy stream seeded at main.py:105.  Returns ``None``
    for skipped users (util.py:246,356) else
    ``(seq, timeseq, hours, days, item_idx[101])``.
    """
    target = test if mode == "test" else valid
    if len(train[u]) < 1 or len(target[u]) < 1:
        return None
    T = maxlen
    seq = np.zeros(T, np.int32); timeseq = np.zeros(T, np.int32)
    hours = np.zeros(T, np.int32); days = np.zeros(T, np.int32)
    orig_ts = [None] * T
    idx = T - 1
    if mode == "test":                                           # util.py:255-264
        it, _, ts = valid[u][0]
        seq[idx] = it; orig_ts[idx] = ts
        hours[idx] = hour_of(ts); days[idx] = day_of(ts)
        idx -= 1
    for (it, _, ts) in reversed(train[u]):                       # util.py:265-272
        seq[idx] = it; orig_ts[idx] = ts
        hours[idx] = hour_of(ts); days[idx] = day_of(ts)
        idx -= 1
        if idx == -1:
            break
    most_recent = orig_ts[-1]                                    # util.py:275-289
    for i, ts in enumerate(orig_ts):
        if ts is not None:
            d = float(most_recent - ts)
            if log_scale:
                timeseq[i] = timedelta_bin(d, bin_in_hours, max_bins, True, min_td, max_td)
            else:
                timeseq[i] = timedelta_bin(d, bin_in_hours, max_bins, False)
    rated = set(e[0] for e in train[u]); rated.add(0)            # util.py:291-298
    item_idx = [target[u][0][0]]
    for _ in range(100):
        t = rs.randint(1, itemnum + 1)
        while t in rated:
            t = rs.randint(1, itemnum + 1)
        item_idx.append(t)
    if test_seq_len is not None:                                 # util.py:300-315
        n = min(test_seq_len, maxlen)
        seq[:-n] = 0; timeseq[:-n] = 0; hours[:-n] = 0; days[:-n] = 0
    return seq, timeseq, hours, days, np.asarray(item_idx, np.int32)


def rank_metrics(logits_101):
    """util.py:318-327: rank of candidate 0 among 101; returns (ndcg, hit)."""
    pred = -np.asarray(logits_101)
    rank = pred.argsort().argsort()[0]
    if rank < 10:
        return 1.0 / np.log2(rank + 2), 1.0
    return 0.0, 0.0
