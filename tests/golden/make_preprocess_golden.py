#!/usr/bin/env python3
"""Golden vectors for the preprocessing step: synthetic raw dumps run through the REFERENCE's own
``data_reader.DataReader`` (imported read-only from /root/reference; build container only).  Committed: the raw inputs
and the files the reference wrote for them, under tests/golden/preprocess/.

    python tests/golden/make_preprocess_golden.py

amazon_ratings: the reference reads ``self.input_context`` (data_reader.py:196) without ever setting it; the attribute
is set on the instance here (True, then False) so both of its output layouts are captured."""
import gzip
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "preprocess")
sys.path.insert(0, "/root/reference")
import data_reader as ref            # noqa: E402  (reference)


def synth_events(rs, n_users, n_items, n_events, t0, span, half_stars):
    """Zipf-ish users and items so that some of each fall under the 5-review floor; duplicated timestamps included."""
    pu = 1.0 / np.arange(1, n_users + 1) ** 0.9; pu /= pu.sum()
    pi = 1.0 / np.arange(1, n_items + 1) ** 0.8; pi /= pi.sum()
    users = rs.choice(n_users, n_events, p=pu); items = rs.choice(n_items, n_events, p=pi)
    ts = t0 + rs.randint(0, span, n_events)
    ts[rs.rand(n_events) < 0.15] = t0 + 1000                    # ties: the sort must be stable
    stars = rs.randint(1, 11, n_events) / 2.0 if half_stars else rs.randint(1, 6, n_events).astype(float)
    return users, items, stars, ts


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    rs = np.random.RandomState(7)
    # ---- movielens -----------------------------------------------------------------------------------
    d = os.path.join(OUT, "movielens"); os.makedirs(d)
    ext_u = rs.permutation(np.arange(1, 61)); ext_i = rs.permutation(np.arange(100, 180))
    u, i, r, t = synth_events(rs, 60, 80, 420, 978_300_000, 3_000_000, True)
    with open(os.path.join(d, "ratings.csv"), "w") as f:
        for k in range(len(u)):
            f.write("%d,%d,%s,%d\n" % (ext_u[u[k]], ext_i[i[k]], r[k], t[k]))
    with open(os.path.join(d, "movies.csv"), "w", encoding="ISO-8859-1") as f:
        for key in sorted(ext_i):
            f.write("%d,Title %d (19%02d),Genre%d|Genre%d\n" % (key, key, key % 100, key % 7, key % 3))
    for limit, name in ((None, "out.txt"), (200, "out_limit200.txt")):
        ref.DataReader(os.path.join(d, "ratings.csv"), os.path.join(d, name), "movielens", limit=limit).preprocess()
    # ---- amazon (gzip of python-literal dicts) ----------------------------------------------------
    d = os.path.join(OUT, "amazon"); os.makedirs(d)
    u, i, r, t = synth_events(rs, 50, 70, 380, 1_252_800_000, 40_000_000, False)
    with gzip.open(os.path.join(d, "reviews.json.gz"), "wb") as g:
        for k in range(len(u)):
            rec = {"reviewerID": "A%05dX" % (u[k] * 37 % 1000), "asin": "%010d" % (i[k] * 911), "reviewerName": "N. %d" % k,
                   "helpful": [int(k % 3), 3], "reviewText": "text's \"quoted\" %d" % k, "overall": float(r[k]),
                   "summary": "s", "unixReviewTime": int(t[k]), "reviewTime": "09 13, 2009"}
            g.write((repr(rec) + "\n").encode())
    ref.DataReader(os.path.join(d, "reviews.json.gz"), os.path.join(d, "out.txt"), "amazon").preprocess()
    # ---- amazon_ratings (csv, string ids, string time => lexicographic sort) -----------------------
    d = os.path.join(OUT, "amazon_ratings"); os.makedirs(d)
    u, i, r, t = synth_events(rs, 50, 70, 380, 999_000_000, 3_000_000, False)     # spans 9- and 10-digit timestamps
    with open(os.path.join(d, "ratings.csv"), "w") as f:
        for k in range(len(u)):
            f.write("U%03d,B%04d,%s,%d\n" % (u[k] * 13 % 977, i[k] * 7 % 997, r[k], t[k]))
    for flag, name in ((True, "out4.txt"), (False, "out3.txt")):
        dr = ref.DataReader(os.path.join(d, "ratings.csv"), os.path.join(d, name), "amazon_ratings")
        dr.input_context = flag
        dr.preprocess()
    for root, _, files in os.walk(OUT):
        for fn in sorted(files):
            p = os.path.join(root, fn)
            print("%7d  %s" % (os.path.getsize(p), os.path.relpath(p, OUT)))


if __name__ == "__main__":
    main()
