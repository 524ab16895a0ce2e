"""Statistical sanity of the counter-based dropout generator (restated in tests/dropout_ref.py from
csrc/cr_common.hpp): keep rate, no serial correlation along the element index, no column structure."""
import numpy as np
import pytest

import dropout_ref as dr


@pytest.mark.parametrize("rate", [0.1, 0.2, 0.5])
def test_keep_rate_and_independence(rate):
    rs = np.random.RandomState(int(rate * 100))
    n = 400_000
    worst_rate = worst_ac = worst_col = 0.0
    for trial in range(6):
        seed, step, site = int(rs.randint(1, 2 ** 31)), int(rs.randint(1, 10 ** 6)), int(rs.randint(0, 64))
        idx = np.arange(n, dtype=np.uint64) + np.uint64(rs.randint(0, 2 ** 31))
        k = dr.keep_mask(seed, step, site, rate, idx & dr.M32).astype(np.float64)
        p = 1.0 - float(np.float32(rate))
        worst_rate = max(worst_rate, abs(k.mean() - p) / np.sqrt(p * (1 - p) / n))
        kc = k - k.mean()
        for lag in (1, 2, 3, 4, 16, 50, 64, 200, 201):
            worst_ac = max(worst_ac, abs((kc[:-lag] * kc[lag:]).mean() / kc.var()) * np.sqrt(n))
        cols = k[:200 * 2000].reshape(2000, 200).mean(0)                     # a [rows, T=200] attention-like layout
        worst_col = max(worst_col, np.abs((cols - p) / np.sqrt(p * (1 - p) / 2000)).max())
    assert worst_rate < 4.5, worst_rate          # z-scores: 6 trials
    assert worst_ac < 4.8, worst_ac              # 54 tests
    assert worst_col < 5.2, worst_col            # 1200 tests


def test_steps_and_sites_are_independent():
    idx = np.arange(200_000, dtype=np.uint64)
    a = dr.keep_mask(7, 10, 3, 0.5, idx).astype(np.float64)
    for other in (dr.keep_mask(7, 11, 3, 0.5, idx), dr.keep_mask(7, 10, 4, 0.5, idx), dr.keep_mask(8, 10, 3, 0.5, idx)):
        b = other.astype(np.float64)
        c = ((a - a.mean()) * (b - b.mean())).mean() / np.sqrt(a.var() * b.var())
        assert abs(c) * np.sqrt(len(idx)) < 4.5
