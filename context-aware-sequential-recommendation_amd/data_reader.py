"""Raw dump -> the 4-column ``user item rating ts`` training file (reference: data_reader.py:10-281, driven by
preprocess.py:6-9).  Same class name, constructor and ``preprocess()`` entry as the reference so a caller switches by
import; the three dataset types share ONE filter/remap/sort routine here instead of three copies.

What the reference does, per type (all keep a review only if its user AND its item each have >= 5 reviews in the raw
dump -- a single pass over raw counts, not an iterated k-core; ids are then assigned densely from 1 in order of first
appearance; each user's reviews are stably sorted by time; users are written in id order):

  movielens       ``user,item,rating,ts`` csv lines (data_reader.py:59-142); rating stored as int(rating*10)
                  (half stars), ids compared as ints; also writes ``<out>_metadata.tsv`` / ``<out>_metadata_genres.tsv``
                  (``index title`` / ``index genre``) from ``movies.csv`` next to the raw file, 3 comma-separated fields
                  per line (:125-141).  Here: written when movies.csv exists, skipped otherwise (the reference raises
                  FileNotFoundError after the dataset itself has been written).
  amazon          gzip of one Python-literal dict per line with reviewerID / asin / overall / unixReviewTime
                  (:43-48, 216-281); rating written as the float's str ("5.0"); ``<out>_product_map.txt`` = ``asin index``.
  amazon_ratings  ``user,item,rating,ts`` csv with string ids (:143-214).  The time column stays a STRING, so the
                  per-user sort is lexicographic (:190-191) -- kept, it only differs from numeric order for timestamps
                  of different digit counts.  The reference then reads ``self.input_context`` (:196), which its
                  constructor never sets (AttributeError); here it is a class attribute, default True = 4 columns
                  (what util.get_users reads), False = ``user item ts``.  Product map goes to ``<out minus its last four
                  characters>_product_map.txt`` (:210).

``limit`` keeps the reference's off-by-one: records 0..limit inclusive are read (:45-46, 53-54)."""
import ast
import gzip
import logging
import os
from collections import Counter

log = logging.getLogger("ir2")


class DataReader:
    input_context = True

    def __init__(self, path, dataset_fp, type, limit=None, maxlen=None):
        self.path, self.dataset_fp, self.type, self.limit, self.maxlen = path, dataset_fp, type, limit, maxlen
        self.logger = log

    # ---- raw records: (user key, item key, rating as written, time as sorted) ---------------------
    def _limited(self, it):
        for n, rec in enumerate(it):
            if self.limit and n > self.limit:
                return
            yield rec

    def _csv_lines(self):
        with open(self.path, "r") as f:
            yield from self._limited(line.rstrip() for line in f)

    def _records(self):
        if self.type == "movielens":
            for line in self._csv_lines():
                user, item, rating, ts = line.split(",")
                yield int(user), int(item), int(float(rating) * 10), int(ts)
        elif self.type == "amazon_ratings":
            for line in self._csv_lines():
                user, item, rating, ts = line.split(",")
                yield user, item, rating, ts
        else:
            with gzip.open(self.path, "rb") as g:
                for raw in self._limited(g):
                    d = ast.literal_eval(raw.decode("utf-8").strip())
                    yield d["reviewerID"], d["asin"], d["overall"], d["unixReviewTime"]

    # ---- the shared pipeline ------------------------------------------------------------------
    def _filter_remap_sort(self, min_count=5):
        n_user, n_item = Counter(), Counter()
        for user, item, _, _ in self._records():
            n_user[user] += 1
            n_item[item] += 1
        user_id, item_id, events = {}, {}, []
        for user, item, rating, ts in self._records():
            if n_user[user] < min_count or n_item[item] < min_count:
                continue
            u = user_id.setdefault(user, len(user_id) + 1)
            if u > len(events):
                events.append([])
            events[u - 1].append((item_id.setdefault(item, len(item_id) + 1), rating, ts))
        for ev in events:
            ev.sort(key=lambda e: e[2])                        # stable: equal times keep file order
        return events, item_id

    def preprocess(self):
        assert isinstance(self.type, str)
        if self.type not in ("amazon", "movielens", "amazon_ratings"):
            raise ValueError("unknown dataset type %r (amazon, movielens, amazon_ratings)" % self.type)
        log.info("Reading and processing %s", self.path)
        events, item_id = self._filter_remap_sort()
        four = self.type != "amazon_ratings" or self.input_context
        with open(self.dataset_fp, "w") as f:
            for u, ev in enumerate(events, 1):
                for item, rating, ts in ev:
                    f.write("%s %s %s %s\n" % (u, item, rating, ts) if four else "%s %s %s\n" % (u, item, ts))
        d, bn = os.path.dirname(self.dataset_fp), os.path.basename(self.dataset_fp)
        if self.type == "movielens":
            self._write_movie_labels(d, bn, item_id)
        else:
            stem = bn[:-4] if self.type == "amazon_ratings" else bn
            with open(os.path.join(d, stem + "_product_map.txt"), "w") as f:
                for key, idx in item_id.items():
                    f.write("%s %s\n" % (key, idx))
        return len(events), len(item_id)

    def _write_movie_labels(self, d, bn, item_id):
        src = os.path.join(os.path.dirname(self.path), "movies.csv")
        if not os.path.exists(src):
            log.warning("%s not found: label files not written", src)
            return
        labels = {}
        with open(src, "r", encoding="ISO-8859-1") as f:
            for line in f:
                key, title, genre = line.rstrip().split(",")
                labels[int(key)] = (title, genre)
        with open(os.path.join(d, bn + "_metadata.tsv"), "w") as ft, open(os.path.join(d, bn + "_metadata_genres.tsv"), "w") as fg:
            for key, idx in item_id.items():
                ft.write("%s %s\n" % (idx, labels[key][0]))
                fg.write("%s %s\n" % (idx, labels[key][1]))


def main(raw_dataset, out_dataset, dataset_type, limit=None):
    """preprocess.py:6-9."""
    return DataReader(raw_dataset, out_dataset, dataset_type, limit=limit).preprocess()
