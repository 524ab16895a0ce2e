"""Seeded synthetic (user, item-sequence) corpora of the shapes SURVEY.md section 8d
names for the BASELINE configs (no datasets are available offline).

A corpus is held in CSR form -- the layout the native sampler consumes:
``offsets[int64, n_users+2]`` (row u = user id u; row 0 is empty), ``items[int32]``,
``ratings[float32]``, ``ts[int64]`` -- events of one user are time-sorted, as the
reference's pre-processing writes them (data_reader.py:116-121).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Corpus:
    usernum: int
    itemnum: int
    offsets: np.ndarray   # int64 [usernum + 2]
    items: np.ndarray     # int32 [nnz]
    ratings: np.ndarray   # float32 [nnz]
    ts: np.ndarray        # int64 [nnz]

    def user(self, u):
        a, b = int(self.offsets[u]), int(self.offsets[u + 1])
        return self.items[a:b], self.ratings[a:b], self.ts[a:b]

    def to_dict(self):
        """{user: [(item, rating, ts), ...]} -- the oracle's / reference's view."""
        out = {}
        for u in range(1, self.usernum + 1):
            it, r, t = self.user(u)
            out[u] = [(int(a), float(b), int(c)) for a, b, c in zip(it, r, t)]
        return out

    def write_text(self, path):
        """4-column ``user item rating ts`` text (util.py:170 reads this)."""
        with open(path, "w") as f:
            for u in range(1, self.usernum + 1):
                it, r, t = self.user(u)
                for a, b, c in zip(it, r, t):
                    f.write("%d %d %.1f %d\n" % (u, a, b, c))


PRESETS = {
    # name: (n_users, n_items, mu, sigma, Lmax, zipf_alpha, seed)
    "tiny":   (5, 40, 2.0, 0.5, 30, 0.8, 7),
    "ml-1m":  (6040, 3416, 4.6, 0.9, 2300, 0.8, 42),      # C1 / C2
    "beauty": (52000, 57289, 2.0, 0.6, 200, 1.1, 43),     # C3 (long tail)
    "books":  (600000, 368000, 2.3, 0.9, 2000, 1.0, 44),  # C4
    # C5 item side at full size (10 M items, Zipf 1.05, half of the users longer than 512 events); the user count is cut
    # from 10^6 to 4 000 -- the sampler draws users uniformly, so a step's work does not depend on it.  A user's items are
    # distinct (round 3: first occurrences of a vectorised draw, see make_corpus(unique_items="fast"))
    "c5":     (4000, 10_000_000, 6.25, 0.6, 1500, 1.05, 45),
}


def make_corpus(n_users, n_items, mu, sigma, lmax, alpha, seed, lmin=3,
                t0=956_700_000, span_days=400, unique_items=True):
    """Lengths ~ clip(lognormal(mu, sigma), lmin, lmax); items ~ Zipf(alpha) over a
    random permutation of 1..n_items (distinct per user, like the 5-core datasets);
    timestamps sorted uniform over ``span_days`` from ``t0``; rating 4.0."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.floor(rs.lognormal(mu, sigma, n_users)), lmin, min(lmax, n_items // 2)).astype(np.int64)
    w = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** alpha
    cdf = np.cumsum(w / w.sum())
    perm = rs.permutation(n_items).astype(np.int32) + 1
    offsets = np.zeros(n_users + 2, np.int64)
    offsets[2:] = np.cumsum(lens)
    nnz = int(offsets[-1])
    items = np.empty(nnz, np.int32)
    ts = np.empty(nnz, np.int64)
    for i in range(n_users):
        a, n = int(offsets[i + 1]), int(lens[i])
        if unique_items == "fast":
            # distinct items as the first occurrences of a vectorised draw (same law as the element-wise loop below; a
            # different random stream, so only presets that never had the loop use it)
            got = np.empty(0, np.int32)
            while len(got) < n:
                cand = perm[np.searchsorted(cdf, rs.random_sample(2 * (n - len(got)) + 8)).clip(0, n_items - 1)]
                allc = np.concatenate([got, cand])
                _, first = np.unique(allc, return_index=True)
                got = allc[np.sort(first)][:n]
            items[a:a + n] = got
        elif unique_items:
            got = []
            seen = set()
            while len(got) < n:
                cand = perm[np.searchsorted(cdf, rs.random_sample(2 * (n - len(got)) + 8)).clip(0, n_items - 1)]
                for c in cand:
                    c = int(c)
                    if c not in seen:
                        seen.add(c); got.append(c)
                        if len(got) == n:
                            break
            items[a:a + n] = got
        else:
            items[a:a + n] = perm[np.searchsorted(cdf, rs.random_sample(n)).clip(0, n_items - 1)]
        ts[a:a + n] = np.sort(t0 + rs.randint(0, span_days * 86400, n))
    ratings = np.full(nnz, 4.0, np.float32)
    return Corpus(n_users, n_items, offsets, items, ratings, ts)


def preset(name):
    n_users, n_items, mu, sigma, lmax, alpha, seed = PRESETS[name]
    return make_corpus(n_users, n_items, mu, sigma, lmax, alpha, seed, unique_items=("fast" if name == "c5" else True))


def from_dict(corpus_dict, usernum, itemnum):
    """Inverse of ``Corpus.to_dict`` (users absent from the dict get empty rows)."""
    offsets = np.zeros(usernum + 2, np.int64)
    items, ratings, ts = [], [], []
    for u in range(1, usernum + 1):
        ev = corpus_dict.get(u, [])
        offsets[u + 1] = offsets[u] + len(ev)
        for (i, r, t) in ev:
            items.append(i); ratings.append(r); ts.append(t)
    return Corpus(usernum, itemnum, offsets, np.asarray(items, np.int32),
                  np.asarray(ratings, np.float32), np.asarray(ts, np.int64))
