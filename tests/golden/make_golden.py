#!/usr/bin/env python3
"""Generate the integer-path golden vectors by RUNNING THE REFERENCE's own
``sampler.py`` / ``util.py`` (imported read-only from /root/reference).

Run in the build container only -- the reference never travels to the GPU box;
only the outputs below (inputs + expected outputs, plain int/float arrays) are
committed:

    tests/golden/sampler_golden.npz    sample_function batches
    tests/golden/util_golden.json      time bins, delta ranges, hour/day, partition
    tests/golden/eval_golden.npz       evaluate / evaluate_valid predict() inputs

    python tests/golden/make_golden.py
"""
import io
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))   # util.py:11, unused on this path
sys.path.insert(0, "/root/reference")

import util as ref_util          # noqa: E402  (reference)
import sampler as ref_sampler    # noqa: E402  (reference)

import castrec_amd               # noqa: E402
from castrec_amd import synth    # noqa: E402


class _Stop(Exception):
    pass


class CaptureQueue:
    """Stands in for multiprocessing.Queue: sample_function only calls .put (sampler.py:81)."""

    def __init__(self, n):
        self.n, self.out = n, []

    def put(self, zipped):
        cols = [list(c) for c in zipped]
        self.out.append(cols)
        if len(self.out) >= self.n:
            raise _Stop


def ref_corpus(corpus_dict):
    return {u: [ref_util.UserItems(i, r, t) for (i, r, t) in ev] for u, ev in corpus_dict.items()}


def run_ref_sampler(U, usernum, itemnum, B, T, bin_in_hours, max_bins, log_scale, seed, nb):
    mn, mx = ref_util.get_delta_range(U)
    q = CaptureQueue(nb)
    try:
        ref_sampler.sample_function(U, usernum, itemnum, B, T, q, bin_in_hours, max_bins,
                                    log_scale, mn, mx, seed)
    except _Stop:
        pass
    batches = []
    for cols in q.out:
        user, seq, pos, neg, timeseq, ratings, hours, days, _orig = cols
        batches.append([np.asarray(user, np.int32)] +
                       [np.stack(x).astype(np.int32) for x in (seq, pos, neg, timeseq, ratings, hours, days)])
    return (float(mn), float(mx)), batches


CORPORA = {
    # name: make_corpus args
    "tiny": dict(n_users=5, n_items=40, mu=2.0, sigma=0.5, lmax=30, alpha=0.8, seed=7),
    "ml": dict(n_users=300, n_items=500, mu=4.6, sigma=0.9, lmax=400, alpha=0.8, seed=42),
    "tail": dict(n_users=400, n_items=2000, mu=2.0, sigma=0.6, lmax=60, alpha=1.1, seed=43, span_days=30),
}


def main():
    arrays, meta = {}, {"cases": []}
    corp = {}
    for name, kw in CORPORA.items():
        kw = dict(kw)
        span = kw.pop("span_days", 400)
        c = synth.make_corpus(kw["n_users"], kw["n_items"], kw["mu"], kw["sigma"], kw["lmax"],
                              kw["alpha"], kw["seed"], span_days=span)
        # a few users with <= 1 train events (exercise the re-draw loop, sampler.py:20-21)
        corp[name] = c
        for k in ("offsets", "items", "ratings", "ts"):
            arrays["corpus/%s/%s" % (name, k)] = getattr(c, k)
        meta["corpus/" + name] = {"usernum": c.usernum, "itemnum": c.itemnum}

    # ---- sampler goldens ---------------------------------------------------
    # cases exercise: T shorter / longer than sequences, both bin modes, 2 seeds,
    # non-default bin_in_hours / max_bins (clamp), small itemnum (rejections).
    cases = []
    for name, B, nb in (("tiny", 4, 3), ("ml", 16, 2), ("tail", 16, 2)):
        for T in (6, 50, 200):
            for seed in (42, 1):
                for log_scale in (False, True):
                    cases.append((name, B, T, 24, 200, log_scale, seed, nb))
    cases.append(("ml", 8, 50, 48, 200, False, 42, 2))
    cases.append(("ml", 8, 50, 1, 20, False, 7, 2))      # clamp at max_bins
    cases.append(("tail", 8, 20, 2, 5, False, 3, 2))
    for ci, (name, B, T, bih, mb, ls, seed, nb) in enumerate(cases):
        c = corp[name]
        d = c.to_dict()
        # train split as main.py uses it (util.py:204-227)
        U = ref_corpus(d)
        train = {}
        for u in U:
            train[u] = U[u] if len(U[u]) < 3 else U[u][:-2]
        (mn, mx), batches = run_ref_sampler(train, c.usernum, c.itemnum, B, T, bih, mb, ls, seed, nb)
        key = "samp/%03d" % ci
        meta["cases"].append(dict(key=key, corpus=name, B=B, T=T, bin_in_hours=bih, max_bins=mb,
                                  log_scale=ls, seed=seed, nb=nb, min_td=mn, max_td=mx))
        for bi, cols in enumerate(batches):
            for nm, arr in zip(("user", "seq", "pos", "neg", "timeseq", "ratings", "hours", "days"), cols):
                arrays["%s/b%d/%s" % (key, bi, nm)] = arr
    np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **arrays)
    with open(os.path.join(HERE, "sampler_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)

    # ---- util goldens --------------------------------------------------------
    ug = {}
    # hour / day at boundary timestamps (midnight UTC, week wrap, DST-irrelevant)
    tss = [0, 1, 3599, 3600, 86399, 86400, 345600 - 1, 345600, 604800 - 1, 604800,
           956700000, 978300019, 978307199, 978307200, 1000000000, 1356998399, 1356998400,
           1571513047, 1700000000, 2000000000]
    ug["hour_day"] = [[t, ref_util.UserItems(1, 1.0, t).ts.hour, ref_util.UserItems(1, 1.0, t).ts.day] for t in tss]
    # linear bins
    rs = np.random.RandomState(5)
    deltas = [0, 1, 3599, 3600, 86399, 86400, 172799, 172800, 172801, 48 * 3600 * 200 - 1, 48 * 3600 * 200,
              48 * 3600 * 201, 10 ** 9] + [int(x) for x in rs.randint(0, 40000000, 200)]
    lin = []
    for bih, mb in ((48, 200), (24, 200), (1, 20), (2, 5), (168, 50)):
        lin.append(dict(bin_in_hours=bih, max_bins=mb,
                        bins=[ref_util.get_timedelta_bin(float(d), bin_in_hours=bih, max_bins=mb, log_scale=False)
                              for d in deltas]))
    ug["deltas"] = deltas
    ug["linear"] = lin
    logb = []
    for (mn, mx, mb) in ((0.0, 58896613.0, 200), (0.0, 31449600.0, 200), (0.0, 1234567.25, 50), (5.0, 10 ** 6, 10)):
        logb.append(dict(min_ts=mn, max_ts=mx, max_bins=mb,
                         bins=[ref_util.get_timedelta_bin(float(d), max_bins=mb, log_scale=True, min_ts=mn, max_ts=mx)
                               for d in deltas]))
    ug["log"] = logb
    # delta range per corpus (on the full corpus and on the train split)
    dr = {}
    for name, c in corp.items():
        U = ref_corpus(c.to_dict())
        mn, mx = ref_util.get_delta_range(U)
        dr[name] = [float(mn), float(mx)]
    ug["delta_range"] = dr
    # data_partition on a 4-column file incl. users with < 3 events and non-contiguous ids
    lines = []
    small = {1: [(3, 5.0, 100), (4, 4.0, 200), (5, 3.0, 300), (6, 1.0, 400)],
             2: [(7, 2.0, 50)],
             4: [(3, 5.0, 10), (9, 4.5, 20)],
             5: [(1, 1.0, 1), (2, 2.0, 2), (3, 3.0, 3)],
             9: [(11, 1.0, 5), (12, 1.0, 6), (13, 1.0, 7), (14, 1.0, 8), (15, 1.0, 9)]}
    for u, ev in small.items():
        for (i, r, t) in ev:
            lines.append("%d %d %.1f %d" % (u, i, r, t))
    txt = "\n".join(lines) + "\n"
    tmp = "/tmp/_golden_part.txt"
    with open(tmp, "w") as f:
        f.write(txt)
    train, valid, test, usernum, itemnum, ratingnum = ref_util.data_partition(tmp)
    dump = lambda d: {str(u): [[x.item, x.rating, x.timestamp_raw] for x in v] for u, v in d.items()}
    ug["partition"] = dict(text=txt, train=dump(train), valid=dump(valid), test=dump(test),
                           usernum=usernum, itemnum=itemnum, ratingnum=ratingnum)
    with open(os.path.join(HERE, "util_golden.json"), "w") as f:
        json.dump(ug, f, indent=1)

    # ---- evaluate goldens: capture the inputs of model.predict -----------------
    class Args:
        pass

    class FakeModel:
        def __init__(self):
            self.calls = []

        def predict(self, sess, u, seq, item_idx, timeseq=None, hours_seq=None, days_seq=None):
            self.calls.append((int(u[0]), np.array(seq[0]), np.array(timeseq[0]), np.array(hours_seq[0]),
                               np.array(days_seq[0]), np.array(item_idx, np.int32)))
            # deterministic pseudo-logits so the metric arithmetic is exercised too
            h = (np.array(item_idx, np.int64) * 2654435761 + int(u[0]) * 40503) % 1000003
            return (h.astype(np.float64) / 1000003.0)[None, :], np.zeros((1, 1, 1))

    ev_arrays, ev_meta = {}, []
    for ci, (name, T, bih, mb, ls, tsl) in enumerate((("tiny", 6, 24, 200, False, None),
                                                      ("ml", 50, 48, 200, False, None),
                                                      ("ml", 200, 24, 200, True, None),
                                                      ("tail", 20, 2, 5, False, None),
                                                      ("ml", 50, 48, 200, False, 10))):
        c = corp[name]
        U = ref_corpus(c.to_dict())
        train, valid, test = {}, {}, {}
        for u in U:
            if len(U[u]) < 3:
                train[u], valid[u], test[u] = U[u], [], []
            else:
                train[u], valid[u], test[u] = U[u][:-2], [U[u][-2]], [U[u][-1]]
        dataset = [train, valid, test, c.usernum, c.itemnum, 5.0]
        args = Args()
        args.maxlen, args.bin_in_hours, args.max_bins, args.log_scale = T, bih, mb, ls
        args.test_model = "/tmp" if tsl else None
        args.test_seq_len = tsl
        ref_util.plot_attention_weights = lambda *a, **k: None
        random.seed(42); np.random.seed(42)                       # main.py:104-105
        fm_t, fm_v = FakeModel(), FakeModel()
        t_test = ref_util.evaluate(fm_t, dataset, args, None)      # main.py:232
        t_valid = ref_util.evaluate_valid(fm_v, dataset, args, None)  # main.py:233
        key = "eval/%02d" % ci
        ev_meta.append(dict(key=key, corpus=name, T=T, bin_in_hours=bih, max_bins=mb, log_scale=ls,
                            test_seq_len=tsl, test=[float(t_test[0]), float(t_test[1])],
                            valid=[float(t_valid[0]), float(t_valid[1])]))
        for mode, fm in (("test", fm_t), ("valid", fm_v)):
            keep = fm.calls[:64]            # first 64 predict() calls are stored verbatim
            ev_arrays["%s/%s/n_calls" % (key, mode)] = np.int64(len(fm.calls))
            for nm, j in (("user", 0), ("seq", 1), ("timeseq", 2), ("hours", 3), ("days", 4), ("item_idx", 5)):
                ev_arrays["%s/%s/%s" % (key, mode, nm)] = np.stack([np.asarray(cl[j]) for cl in keep])
            # checksum of ALL candidate lists (pins the RNG stream position end-to-end)
            allc = np.stack([cl[5] for cl in fm.calls]).astype(np.int64)
            ev_arrays["%s/%s/cand_checksum" % (key, mode)] = np.int64((allc * (np.arange(101) + 1)).sum())
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **ev_arrays)
    with open(os.path.join(HERE, "eval_golden.json"), "w") as f:
        json.dump(ev_meta, f, indent=1)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()

