"""Host-side mirror of util.py (castrec_amd.util) against the reference-generated goldens (CPU only)."""
import random
import types

import numpy as np
import pytest

import castrec_amd  # noqa: F401
from castrec_amd import util as U
from helpers import load_eval_golden, load_sampler_golden, load_util_golden


def test_hour_day_bins_and_partition_golden(tmp_path):
    g = load_util_golden()
    for ts, hour, day in g["hour_day"]:
        assert (U.hour_of(ts), U.day_of(ts)) == (hour, day)
    for tab in g["linear"]:
        assert [U.get_timedelta_bin(float(d), tab["bin_in_hours"], tab["max_bins"], False) for d in g["deltas"]] == tab["bins"]
    for tab in g["log"]:
        assert [U.get_timedelta_bin(float(d), 48, tab["max_bins"], True, tab["min_ts"], tab["max_ts"]) for d in g["deltas"]] == tab["bins"]
    p = tmp_path / "d.txt"
    p.write_text(g["partition"]["text"])
    tr, va, te, usernum, itemnum, ratingnum = U.data_partition(str(p))
    assert (usernum, itemnum, ratingnum) == (g["partition"]["usernum"], g["partition"]["itemnum"], g["partition"]["ratingnum"])
    for got, want in ((tr, g["partition"]["train"]), (va, g["partition"]["valid"]), (te, g["partition"]["test"])):
        assert {str(u): [list(e) for e in ev] for u, ev in got.items()} == want


class FakeModel:
    """Batched stand-in with the pseudo-logits tests/golden/make_golden.py used; records its inputs."""

    def __init__(self):
        self.calls = []

    def predict(self, sess, u, seq, item_idx, timeseq=None, hours_seq=None, days_seq=None):
        out = []
        for i, uu in enumerate(u):
            self.calls.append((int(uu), seq[i].copy(), timeseq[i].copy(), hours_seq[i].copy(), days_seq[i].copy(), item_idx[i].copy()))
            h = (item_idx[i].astype(np.int64) * 2654435761 + int(uu) * 40503) % 1000003
            out.append(h.astype(np.float64) / 1000003.0)
        return np.stack(out), None


def test_batched_evaluate_reproduces_reference_inputs_and_metrics():
    z, meta = load_eval_golden()
    _, _, corpora = load_sampler_golden()
    for case in meta:
        c = corpora[case["corpus"]]
        dataset = U.partition(c.to_dict(), c.usernum, c.itemnum)
        args = types.SimpleNamespace(maxlen=case["T"], bin_in_hours=case["bin_in_hours"], max_bins=case["max_bins"],
                                     log_scale=case["log_scale"], test_model=("/tmp" if case["test_seq_len"] else None),
                                     test_seq_len=case["test_seq_len"])
        random.seed(42); np.random.seed(42)                     # main.py:104-105
        fm_t, fm_v = FakeModel(), FakeModel()
        t_test = U.evaluate(fm_t, dataset, args)                # main.py:232 (test first, then valid: one RNG stream)
        t_valid = U.evaluate_valid(fm_v, dataset, args)
        assert list(t_test) == pytest.approx(case["test"], abs=1e-12)
        assert list(t_valid) == pytest.approx(case["valid"], abs=1e-12)
        for mode, fmk in (("test", fm_t), ("valid", fm_v)):
            key = "%s/%s" % (case["key"], mode)
            assert len(fmk.calls) == int(z[key + "/n_calls"])
            for j, nm in enumerate(("user", "seq", "timeseq", "hours", "days", "item_idx")):
                want = z["%s/%s" % (key, nm)]
                got = np.stack([np.asarray(cl[j]) for cl in fmk.calls[:len(want)]])
                np.testing.assert_array_equal(got, want, err_msg=key + "/" + nm)
            allc = np.stack([cl[5] for cl in fmk.calls]).astype(np.int64)
            assert int((allc * (np.arange(101) + 1)).sum()) == int(z[key + "/cand_checksum"])


def test_reads_the_two_column_sasrec_format(tmp_path):
    """`data/Video.txt` of the reference (baselines/SASRec/util.py:16-27): ``user item`` per line."""
    from castrec_amd import util as U
    f = tmp_path / "video.txt"
    f.write_text("1 5\n1 7\n1 9\n1 2\n2 4\n2 5\n3 1\n3 2\n3 3\n")
    train, valid, test, usernum, itemnum, ratingnum = U.data_partition(str(f))
    assert (usernum, itemnum) == (3, 9)
    assert [e[0] for e in train[1]] == [5, 7] and valid[1][0][0] == 9 and test[1][0][0] == 2
    assert train[2] and valid[2] == [] and test[2] == []                   # fewer than 3 events: all train
    ts = [e[2] for e in train[1] + valid[1] + test[1]]
    assert ts == sorted(ts) and ts[1] - ts[0] == 86400                      # synthetic cadence: one day per event
    g = tmp_path / "three.txt"
    g.write_text("1 5 100\n1 6 200\n1 7 300\n")
    tr, va, te, *_ = U.data_partition(str(g))
    assert tr[1][0][2] == 100 and te[1][0][2] == 300


def test_attention_heat_map_svg_round_trips(tmp_path):
    """plot_attention_weights (util.py:46-54): file name, and the embedded PNG decodes to the 'hot' colours of the matrix."""
    import base64, re, struct, zlib
    a = np.tril(np.arange(36, dtype=np.float64).reshape(6, 6))
    out = U.plot_attention_weights(a, str(tmp_path))
    assert out.endswith("attention_weights.svg")
    svg = open(out).read()
    assert "Attention weights" in svg
    png = base64.b64decode(re.search(r"base64,([A-Za-z0-9+/=]+)", svg).group(1))
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    w, h = struct.unpack(">II", png[16:24])
    assert (w, h) == (6, 6)
    i = png.index(b"IDAT")
    n = struct.unpack(">I", png[i - 4:i])[0]
    rows = zlib.decompress(png[i + 4:i + 4 + n])
    px = np.frombuffer(rows, np.uint8).reshape(6, 1 + 18)[:, 1:].reshape(6, 6, 3)
    assert tuple(px[0, 5]) == (0, 0, 0) and tuple(px[5, 5]) == (255, 255, 255)      # min -> black, max -> white
    assert px[3, 0, 0] == 255 and 0 < px[3, 0, 1] < 255 and px[3, 0, 2] == 0         # 18/35 = .51: red full, green partial
    with pytest.raises(ValueError):
        U.plot_attention_weights(np.zeros((2, 3, 3)), str(tmp_path))


def test_log_scale_needs_200_bins():
    """sampler.py:66 hard-codes max_bins=200 for log-scale bins; a smaller time table would be indexed out of range."""
    import types
    import pytest as _pt
    from castrec_amd.sampler import WarpSampler
    from castrec_amd import synth
    corpus = synth.preset("tiny") if hasattr(synth, "preset") else None
    args = types.SimpleNamespace(seed=1, bin_in_hours=48, max_bins=100, log_scale=True)
    with _pt.raises(ValueError, match="max_bins must be >= 200"):
        WarpSampler(args, corpus, 5, 5, batch_size=2, maxlen=4)
