"""Shared test helpers: golden fixture loading."""
import json
import os

import numpy as np

import castrec_amd  # noqa: F401  (registers the package)
from castrec_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_sampler_golden():
    z = np.load(os.path.join(GOLDEN, "sampler_golden.npz"))
    with open(os.path.join(GOLDEN, "sampler_golden.json")) as f:
        meta = json.load(f)
    corpora = {}
    for name in ("tiny", "ml", "tail"):
        m = meta["corpus/" + name]
        corpora[name] = synth.Corpus(m["usernum"], m["itemnum"],
                                     z["corpus/%s/offsets" % name], z["corpus/%s/items" % name],
                                     z["corpus/%s/ratings" % name], z["corpus/%s/ts" % name])
    return z, meta, corpora


def load_eval_golden():
    z = np.load(os.path.join(GOLDEN, "eval_golden.npz"))
    with open(os.path.join(GOLDEN, "eval_golden.json")) as f:
        meta = json.load(f)
    return z, meta


def load_util_golden():
    with open(os.path.join(GOLDEN, "util_golden.json")) as f:
        return json.load(f)


def train_split(corpus_dict):
    """main.py:94 -> util.data_partition train part."""
    return {u: (ev if len(ev) < 3 else ev[:-2]) for u, ev in corpus_dict.items()}


BATCH_FIELDS = ("user", "seq", "pos", "neg", "timeseq", "ratings", "hours", "days")
