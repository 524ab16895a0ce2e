"""Native sampler (C ABI cr_sampler_*) vs the reference-generated golden batches: bit-exact."""
import time
import types

import numpy as np

import castrec_amd  # noqa: F401
from castrec_amd.sampler import WarpSampler, get_delta_range
from castrec_amd import synth
from helpers import BATCH_FIELDS, load_sampler_golden, train_split
from oracle import intpath as ip


def _args(case):
    return types.SimpleNamespace(seed=case["seed"], bin_in_hours=case["bin_in_hours"], max_bins=case["max_bins"],
                                 log_scale=case["log_scale"])


def test_native_sampler_matches_reference_goldens():
    z, meta, corpora = load_sampler_golden()
    trains = {n: synth.from_dict(train_split(c.to_dict()), c.usernum, c.itemnum) for n, c in corpora.items()}
    for case in meta["cases"]:
        tr = trains[case["corpus"]]
        s = WarpSampler(_args(case), tr, tr.usernum, tr.itemnum, batch_size=case["B"], maxlen=case["T"], n_workers=1)
        try:
            assert float(s.min_timedelta) == case["min_td"] and float(s.max_timedelta) == case["max_td"]
            for bi in range(case["nb"]):
                got = s.next_batch()
                assert got[8] is None
                for name, arr in zip(BATCH_FIELDS, got[:8]):
                    want = z["%s/b%d/%s" % (case["key"], bi, name)]
                    np.testing.assert_array_equal(arr, want, err_msg="%s b%d %s" % (case["key"], bi, name))
        finally:
            s.close()


def test_native_sampler_accepts_reference_style_objects_and_long_streams():
    """dict-of-objects input (what main.py hands over) and 40 batches against the oracle stream."""
    c = synth.make_corpus(120, 300, 3.5, 0.8, 150, 0.9, 11)
    d = train_split(c.to_dict())

    class UI:                                     # duck-types util.UserItems (util.py:32-43)
        def __init__(self, i, r, t):
            self.item, self.rating, self.timestamp_raw = i, r, t

    objs = {u: [UI(*e) for e in ev] for u, ev in d.items()}
    args = types.SimpleNamespace(seed=123, bin_in_hours=12, max_bins=50, log_scale=False)
    s = WarpSampler(args, objs, c.usernum, c.itemnum, batch_size=32, maxlen=40)
    o = ip.SamplerOracle(d, c.usernum, c.itemnum, 32, 40, 12, 50, False, 123)
    try:
        for _ in range(40):
            got, want = s.next_batch(), o.next_batch()
            for a, b in zip(got[:8], want):
                np.testing.assert_array_equal(a, b)
    finally:
        s.close()


def test_delta_range_vectorised_equals_oracle():
    c = synth.make_corpus(50, 100, 3.0, 0.5, 60, 1.0, 5)
    assert tuple(map(float, get_delta_range(c))) == tuple(map(float, ip.delta_range(c.to_dict())))


def test_sampler_throughput_smoke():
    c = synth.make_corpus(600, 3416, 4.6, 0.9, 2300, 0.8, 42)
    args = types.SimpleNamespace(seed=42, bin_in_hours=48, max_bins=200, log_scale=False)
    s = WarpSampler(args, c, c.usernum, c.itemnum, batch_size=128, maxlen=200)
    try:
        s.next_batch()
        t = time.time()
        for _ in range(20):
            s.next_batch()
        dt = time.time() - t
        assert 20 * 128 / dt > 3000        # reference: ~3.0k seq/s/core at maxlen=200
    finally:
        s.close()
