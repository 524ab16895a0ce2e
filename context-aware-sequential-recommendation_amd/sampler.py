"""WarpSampler -- same constructor / methods as the reference's sampler.py:83-136, backed by the
native bit-exact sampler (csrc/cr_sampler.cpp) instead of a spawned Python process.

    sampler = WarpSampler(args, train, usernum, itemnum, batch_size=128, maxlen=200, n_workers=1)
    u, seq, pos, neg, timeseq, ratings_seq, hours_seq, days_seq, orig_seq = sampler.next_batch()
    sampler.close()

Differences (documented in DESIGN.md): batches come back as int32 numpy arrays ``[B]`` / ``[B, maxlen]``
(the reference returns tuples of B arrays -- ``np.asarray`` of those is the same thing); ``orig_seq``
(B*maxlen Python objects, unused by every model) is ``None``; ``n_workers`` is accepted and the
stream is the single-worker stream (the reference seeds all workers identically, sampler.py:108-126).
"""
import ctypes as C

import numpy as np

from . import lib as L
from .synth import Corpus


def _as_corpus(User, usernum, itemnum):
    """Accepts a synth.Corpus, {user: [(item, rating, ts), ...]} or the reference's
    {user: [UserItems, ...]} (objects with .item / .rating / .timestamp_raw, util.py:32-43)."""
    if isinstance(User, Corpus):
        return User
    offsets = np.zeros(usernum + 2, np.int64)
    items, ratings, ts = [], [], []
    for u in range(1, usernum + 1):
        ev = User.get(u, []) if hasattr(User, "get") else User[u]
        offsets[u + 1] = offsets[u] + len(ev)
        for e in ev:
            if isinstance(e, (tuple, list)):
                i, r, t = e[0], e[1], e[2]
            else:
                i, r, t = e.item, e.rating, e.timestamp_raw
            items.append(i); ratings.append(r); ts.append(t)
    return Corpus(usernum, itemnum, offsets, np.asarray(items, np.int32), np.asarray(ratings, np.float32),
                  np.asarray(ts, np.int64))


def get_delta_range(corpus: Corpus):
    """util.py:123-160: (min, 90th percentile) of last_ts - ts over all events (vectorised)."""
    lens = np.diff(corpus.offsets[1:])
    has = lens > 0
    last_idx = corpus.offsets[2:][has] - 1
    last_ts = np.repeat(corpus.ts[last_idx], lens[has])
    deltas = (last_ts - corpus.ts).astype(np.float64)
    return np.amin(deltas), np.percentile(deltas, 90)


class WarpSampler(object):
    def __init__(self, args, User, usernum, itemnum, sample_func=None, batch_size=64, maxlen=10, n_workers=1):
        if args.log_scale and int(args.max_bins) < 200:
            # sampler.py:66 calls get_timedelta_bin(..., max_bins=200, log_scale=True) whatever --max_bins says, so bins
            # reach 200 while the models' time table has max_bins + 1 rows (cast_1.py:31): TensorFlow's lookup raises
            # InvalidArgument on the first such batch; refuse up front instead of reading past the table
            raise ValueError("--log_scale produces time bins up to 200 (sampler.py:66): --max_bins must be >= 200, got %d" % int(args.max_bins))
        corpus = _as_corpus(User, usernum, itemnum)
        self.corpus = corpus
        self.batch_size, self.maxlen = batch_size, maxlen
        self.min_timedelta, self.max_timedelta = get_delta_range(corpus)          # sampler.py:106
        seed = args.seed if getattr(args, "seed", None) else int(np.random.randint(2e9))   # sampler.py:108-111
        self.seed = int(seed) & 0xFFFFFFFF
        self._keep = (np.ascontiguousarray(corpus.offsets, np.int64), np.ascontiguousarray(corpus.items, np.int32),
                      np.ascontiguousarray(corpus.ratings, np.float32), np.ascontiguousarray(corpus.ts, np.int64))
        o, i, r, t = self._keep
        self._h = L.lib.cr_sampler_create(o.ctypes.data, i.ctypes.data, r.ctypes.data, t.ctypes.data,
                                          usernum, itemnum, batch_size, maxlen, int(args.bin_in_hours),
                                          int(args.max_bins), 1 if args.log_scale else 0,
                                          float(self.min_timedelta), float(self.max_timedelta),
                                          self.seed, 10 * max(1, n_workers))
        if not self._h:
            raise RuntimeError("cr_sampler_create failed (empty corpus, item id out of range, or no user with > 1 events)")

    def next_batch(self):
        B, T = self.batch_size, self.maxlen
        user = np.empty(B, np.int32)
        arrs = [np.empty((B, T), np.int32) for _ in range(7)]
        L.check(L.lib.cr_sampler_next(self._h, user.ctypes.data, *[a.ctypes.data for a in arrs]), "cr_sampler_next")
        seq, pos, neg, timeseq, ratings, hours, days = arrs
        return user, seq, pos, neg, timeseq, ratings, hours, days, None

    def close(self):
        if getattr(self, "_h", None):
            L.lib.cr_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
