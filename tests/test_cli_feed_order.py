"""main.py's training loop hands every batch over ONE step ahead (model.feed / model.train_fed) and must still run the sampler's
batches in the sampler's order, once each -- checked without a GPU on a recording stand-in for the model."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Recorder:
    def __init__(self):
        self.fed, self.ran, self.waiting = [], [], []

    def feed(self, u, seq, pos, neg, time_seq=None, hours=None, days=None):
        assert len(self.waiting) <= 1, "at most one batch ahead of the running step"
        self.waiting.append(np.asarray(seq).copy())
        self.fed.append(np.asarray(seq).copy())

    def train_fed(self, fetch=True):
        self.ran.append(self.waiting.pop(0))
        return (0.5, 1.0) if fetch else None

    def data_parallel(self, *a):
        raise AssertionError("single process")

    def save(self, path):
        open(path, "wb").close()
        return path


class RecorderMany(Recorder):
    """A model whose graph launches run up to four steps (Model.steps_per_launch = 4: feed_ahead = 5)."""
    feed_ahead = 5

    def __init__(self):
        super().__init__()
        self.launches = []

    def feed(self, u, seq, pos, neg, time_seq=None, hours=None, days=None):
        assert len(self.waiting) < self.feed_ahead, "at most feed_ahead batches wait"
        self.waiting.append(np.asarray(seq).copy())
        self.fed.append(np.asarray(seq).copy())

    def train_fed(self, fetch=True):
        raise AssertionError("the loop of a multi-step model calls train_fed_many")

    def train_fed_many(self, max_steps=None):
        n = 4 if (len(self.waiting) >= 4 and (max_steps is None or max_steps >= 4)) else 1
        for _ in range(n):
            self.ran.append(self.waiting.pop(0))
        self.launches.append((n, max_steps))
        return n

    def loss_auc(self):
        return (0.5, 1.0)


import pytest


@pytest.mark.parametrize("many", [False, True])
def test_main_feeds_one_batch_ahead_in_sampler_order(tmp_path, monkeypatch, many):
    import main as cli
    from castrec_amd.sampler import WarpSampler
    rec = RecorderMany() if many else Recorder()
    monkeypatch.setattr(cli, "build_model", lambda *a, **k: rec)
    monkeypatch.setattr(cli, "evaluate", lambda *a, **k: (0.1, 0.2))
    monkeypatch.setattr(cli, "evaluate_valid", lambda *a, **k: (0.3, 0.4))
    monkeypatch.chdir(tmp_path)
    argv = ["--dataset", "synthetic:tiny", "--train_dir", "t", "--model", "sasrec", "--maxlen", "12", "--batch_size", "4",
            "--num_epochs", "3", "--eval_every", "3", "--max_bins", "20"]
    assert cli.main(argv) == 0
    args = cli.parse_args(argv)
    # the same sampler stream, drawn directly
    from castrec_amd import synth
    from castrec_amd.util import partition, train_corpus
    c = synth.preset("tiny")
    train, valid, test, usernum, itemnum, ratingnum = partition(c.to_dict(), c.usernum, c.itemnum)
    num_batch = round(len(train) / args.batch_size)
    import random
    random.seed(args.seed); np.random.seed(args.seed)
    smp = WarpSampler(args, train_corpus(train, usernum, itemnum), usernum, itemnum, batch_size=args.batch_size, maxlen=args.maxlen, n_workers=1)
    want = [smp.next_batch()[1].copy() for _ in range(3 * num_batch)]
    smp.close()
    assert len(rec.ran) == 3 * num_batch == len(rec.fed) and not rec.waiting       # every fed batch ran, nothing drawn in vain
    for a, b in zip(rec.ran, want):
        np.testing.assert_array_equal(a, b)
    if many:
        # several steps per launch where enough batches wait, never across an epoch's end (max_steps = what the epoch has left)
        assert sum(n for n, _ in rec.launches) == 3 * num_batch and all(n <= m for n, m in rec.launches)
        assert any(n == 4 for n, _ in rec.launches) or num_batch < 4
    # the artefacts of the run directory: log.txt line and the TensorBoard scalars of the stand-in's numbers
    runs = os.listdir(tmp_path / "saved_models" / "synthetic_tiny")
    d = tmp_path / "saved_models" / "synthetic_tiny" / runs[0]
    assert (d / "log.txt").read_text().strip() == "(0.3, 0.4) (0.1, 0.2)"
    from castrec_amd.tb_events import read_events
    ev = read_events(str(d / [x for x in os.listdir(d) if x.startswith("events.out.tfevents.")][0]))
    assert [s for s, _ in ev] == [1, 2, 3, 3] and ev[0][1]["TRAIN/loss"] == 1.0 and abs(ev[3][1]["TEST/HR@10"] - 0.2) < 1e-7


def test_steps_per_launch_is_one_for_data_parallel_and_row_sparse_adam(monkeypatch):
    """Model.steps_per_launch / feed_ahead before the first feed() (main.py reads feed_ahead there): 1 under data parallelism (from
    the configuration data_parallel() stored, the wrapper itself exists only after the first step) and with row-sparse Adam (the
    engine's setting, or what it will read from the environment); 4 / 5 otherwise."""
    import castrec_amd  # noqa: F401
    from castrec_amd import models

    def bare(**kw):
        m = object.__new__(models.SASRec)
        m._graph, m._train, m._batch_global = True, None, None
        m.__dict__.update(kw)
        return m

    monkeypatch.delenv("CASTREC_LAZY_ADAM", raising=False)
    monkeypatch.delenv("CASTREC_STEPS_PER_GRAPH", raising=False)
    assert bare().steps_per_launch == 4 and bare().feed_ahead == 5
    assert bare(_dp_cfg=(0, 2, None, None)).steps_per_launch == 1 and bare(_dp_cfg=(0, 2, None, None)).feed_ahead == 1
    assert bare(_dp_cfg=(0, 1, None, None)).steps_per_launch == 4              # one rank: the plain step
    assert bare(_graph=False).steps_per_launch == 1

    class Eng:
        lazy_adam = True
    assert bare(_train=Eng()).steps_per_launch == 1
    monkeypatch.setenv("CASTREC_LAZY_ADAM", "1")
    assert bare().steps_per_launch == 1 and bare().feed_ahead == 1
    monkeypatch.setenv("CASTREC_LAZY_ADAM", "0")
    monkeypatch.setenv("CASTREC_STEPS_PER_GRAPH", "8")
    assert bare().steps_per_launch == 8 and bare().feed_ahead == 9
